#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
mkdir -p gpurun_out/r5c30
for geo in 256x256k16 256x256k16s4; do
echo "== geo $geo"
MERLIN_HIP_GEMM_SPLIT_GEO=$geo timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -q -x 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
done
for geo in 256x256 256x256k16 256x256k16s4 256x256; do
echo "== geo $geo"
MERLIN_HIP_GEMM_SPLIT_GEO=$geo MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'cross' in k})"
done
exit 0
