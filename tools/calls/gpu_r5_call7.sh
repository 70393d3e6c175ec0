#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c7; rm -rf $O; mkdir -p $O
run() { timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d.get('bit_identical_to_f32_pipeline'))"; }
run "plain growth 4"
MERLIN_HIP_TOPK_GROWTH=3 run "plain growth 3"
MERLIN_HIP_TOPK_GROWTH=3 MERLIN_HIP_TOPK_SPLITS=32 run "plain growth 3 splits 32"
VAR=MERLIN_HIP_DW_LATE VALUES="0 1" REPS="1 2" bash tools/dbg/ab_env.sh
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | cut -c1-300
exit 0
