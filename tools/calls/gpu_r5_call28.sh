#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
mkdir -p gpurun_out/r5c28
timeout 900 python -m pytest tests/test_gpu_bag_backward.py tests/test_gpu_gemm_split.py tests/test_gpu_backward.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -40 | cut -c1-300
MERLIN_HIP_DETERMINISTIC=1 timeout 600 python -m pytest tests/test_gpu_bag_backward.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c28/prof -- python tools/dbg/bag_bwd_probe.py 4 2>&1 | grep -v "$F" | tail -1
f=$(find gpurun_out/r5c28/prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5c28/bag_bwd_multi_kernel_stats.csv
rm -rf gpurun_out/r5c28/prof
for a in f32 bf16x3; do MERLIN_HIP_GEMM_ARITH=$a timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -2 | cut -c1-400; done
exit 0
