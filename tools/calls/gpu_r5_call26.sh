#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
mkdir -p gpurun_out/r5c26
timeout 600 python -m pytest tests/test_gpu_bag_backward.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -30 | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c26/prof -- python tools/dbg/bag_bwd_probe.py 4 2>&1 | grep -v "$F" | tail -3
f=$(find gpurun_out/r5c26/prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5c26/bag_bwd_multi_kernel_stats.csv
head -25 "$f" | cut -c1-200
exit 0
