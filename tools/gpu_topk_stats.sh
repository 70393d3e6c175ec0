#!/bin/bash
# per-kernel times of the top-k call (rocprofv3 kernel stats, parsed with python: kernel names contain commas)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/topk_stats; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python bench.py --workload topk --no-cpu-baseline --steps 6 --warmup 3 --sustain 0 2>/dev/null | tail -1 | cut -c1-300
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/topk_stats/k/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:10]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
