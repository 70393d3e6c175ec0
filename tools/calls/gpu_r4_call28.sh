#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c28; rm -rf $O; mkdir -p $O
f() { grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -60; }
MERLIN_HIP_MLP_CHAIN=0 MERLIN_HIP_FUSED_DLRM=0 timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x 2>&1 | f > $O/no_chain.txt
