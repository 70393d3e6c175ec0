#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c3; rm -rf $O; mkdir -p $O
for sp in 64 128 256; do for gr in 2 3 4; do
  MERLIN_HIP_TOPK_SPLITS=$sp MERLIN_HIP_TOPK_GROWTH=$gr timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('splits $sp growth $gr', round(d['ms_per_step'],3), d.get('bit_identical_to_f32_pipeline'))"
done; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o topk -- python bench.py --workload topk --steps 8 --warmup 4 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/topk_kernel_stats.csv && head -16 "$f" | cut -c1-200
exit 0
