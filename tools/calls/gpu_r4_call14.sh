#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c14; rm -rf $O; mkdir -p $O
for m in classic lookback classic lookback; do
  echo "== $m" | tee -a $O/ab.txt
  if [ $m = classic ]; then export MERLIN_HIP_SORT=classic; else unset MERLIN_HIP_SORT; fi
  python tools/microbench.py embada emb1m 2>/dev/null | grep "embedding bwd" | tee -a $O/ab.txt
done
unset MERLIN_HIP_SORT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python tools/microbench.py embada > /dev/null 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:10.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
timeout 600 python -m pytest tests -m gpu -q -x -k "embedding or backward or fullsize" > $O/pytest_sel.log 2>&1; grep -E "passed|failed" $O/pytest_sel.log | tail -1
