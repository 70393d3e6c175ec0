#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c7; rm -rf $O; mkdir -p $O
MERLIN_HIP_TOPK_FILTER=tiled timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python bench.py --workload topk --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:10.1f} us", r['Percentage'])
PY
find $O -name "*kernel_trace.csv" -delete
