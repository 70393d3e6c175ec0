#!/bin/bash
# HBM traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only) for the
# gather kernels; prints per-kernel per-launch averages.  Usage: tools/pmc_traffic.sh <microbench args>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python tools/microbench.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python tools/microbench.py "$@" > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
out = collections.defaultdict(dict)
for d, name in (("gpurun_out/pmc_fetch", "FETCH_SIZE"), ("gpurun_out/pmc_write", "WRITE_SIZE")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        print("no csv in", d); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name:
            agg[(r["Kernel_Name"][:70] + ("|onesweep" if "onesweep" in r["Kernel_Name"] else ""), r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][name] = (sum(v) / len(v), len(v))
for k, v in sorted(out.items()):
    if any(s in k[0] for s in ("gather", "piece", "carry_apply", "build_keys", "chunk_flags", "onesweep", "interaction", "linear_fwd", "scorer")):
        print(k, {a: f"{b[0]:.4g} (n={b[1]})" for a, b in v.items()})
PY
