#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c24; rm -rf $O; mkdir -p $O
for sel in embada emb1m; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$sel -o t -- python tools/microbench.py $sel > $O/$sel.log 2>&1
  python tools/trace_summary.py $(ls $O/$sel/*kernel_trace.csv | head -1) 2>/dev/null | head -14 | cut -c1-110 | tee $O/${sel}_summary.txt
  rm -rf $O/$sel
done
