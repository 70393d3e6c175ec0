#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_world2.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest.txt
cut -c1-400 $O/pytest.txt | tail -25
