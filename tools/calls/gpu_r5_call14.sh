#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c14; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for b in 65536 65000 61440; do echo "B=$b"; PROBE_B=$b timeout 300 python tools/dbg/fused_f_probe.py 24 27 28 2>&1 | grep -v "$F"; done
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -c 600 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"],4), d["config"]["launch"], "roofline", round(d["roofline"]["frac"],3))
for k,v in d["secondary"].items():
    if isinstance(v,dict):
        print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_step","ms","value","error","frac_of_peak","GBps","bit_identical_to_f32_pipeline","fit_over_resident_step","examples_per_sec_callback_formula")} )
print("cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
exit 0
