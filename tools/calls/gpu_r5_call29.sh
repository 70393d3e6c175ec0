#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
mkdir -p gpurun_out/r5c29
for pad in 0 16 64 128 256; do echo "dx pad $pad floats"; MERLIN_HIP_FUSED_DX_PAD=$pad timeout 300 python tools/dbg/fused_f_probe.py 24 27 28 32 2>&1 | grep -v "$F" | grep "^F ="; done | tee gpurun_out/r5c29/fused_dx_pad.txt
timeout 900 python -m pytest tests/test_gpu_gemm_split.py tests/test_gpu_dense.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -20 | cut -c1-300
exit 0
