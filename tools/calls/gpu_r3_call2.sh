#!/bin/bash
# isolated single-stream kernel traces of the dominant launch and the fused kernels (no side streams: microbench issues on one stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c2; mkdir -p $O
for sel in embada emb1m fused; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$sel -o t -- python tools/microbench.py $sel > $O/$sel.log 2>&1
  grep -v "Warning\|amdgpu.ids" $O/$sel.log | grep " us" 
  f=$(ls $O/$sel/*kernel_trace.csv | head -1); cp $f $O/${sel}_kernel_trace.csv
  cp $(ls $O/$sel/*kernel_stats.csv | head -1) $O/${sel}_kernel_stats.csv
  python tools/trace_summary.py $O/${sel}_kernel_trace.csv | head -14
  rm -rf $O/$sel
done
