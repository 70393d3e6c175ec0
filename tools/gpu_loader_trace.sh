#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ldtrace; mkdir -p gpurun_out/ldtrace
CHUNKS=8388608 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ldtrace -o t -- python tools/gpu_loader_probe.py 16777216 > gpurun_out/ldtrace/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
kt = glob.glob("gpurun_out/ldtrace/*kernel_trace.csv")[0]
mc = glob.glob("gpurun_out/ldtrace/*memory_copy_trace.csv")
starts = []
for r in csv.DictReader(open(kt)):
    if "dlrm_fused_fwd" in r["Kernel_Name"]:
        starts.append(int(r["Start_Timestamp"]))
starts.sort()
per = [(b - a) / 1e6 for a, b in zip(starts, starts[1:])]
copies = []
if mc:
    for r in csv.DictReader(open(mc[0])):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")))
# last 600 steps: classify each step period by whether an H2D copy was in flight
copies = [c for c in copies if (c[1] - c[0]) > 100000]  # > 0.1 ms: the chunk columns
tail = list(zip(starts[-600:], per[-599:]))
inc, outc = [], []
for st, p in tail:
    busy = any(c[0] < st + p * 1e6 and c[1] > st for c in copies)
    (inc if busy else outc).append(p)
import statistics as S
print("big copies:", len(copies), "median ms", S.median([(c[1]-c[0])/1e6 for c in copies]) if copies else None, "dirs", collections.Counter(c[2] for c in copies))
print("step period ms: with a chunk copy in flight n=%d median %.3f mean %.3f | without n=%d median %.3f mean %.3f" % (len(inc), S.median(inc) if inc else 0, S.mean(inc) if inc else 0, len(outc), S.median(outc) if outc else 0, S.mean(outc) if outc else 0))
print("largest periods:", sorted(p for _, p in tail)[-8:])
PY
