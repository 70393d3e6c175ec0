#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c11; rm -rf $O; mkdir -p $O
cat > /tmp/sc.py <<'PY'
import os, torch, sys
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
os.environ["MERLIN_HIP_SCORER_FWD"] = "tiled"
print(bench.run_scorer_fwd(dev))
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python /tmp/sc.py > /dev/null 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:10.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
