#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for pack in 1 0; do
echo "== tests pack $pack"
MERLIN_HIP_GEMM_SPLIT_PACK=$pack MERLIN_HIP_GEMM_SPLIT_PIPE=1 timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -q -x 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
done
MERLIN_HIP_GEMM_SPLIT_GEO=256x128 timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -q -x 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
for cfg in "0 1" "1 1" "0 1" "1 1" "1 0"; do
set -- $cfg
echo "== pack $1 pipe $2"
MERLIN_HIP_GEMM_SPLIT_PACK=$1 MERLIN_HIP_GEMM_SPLIT_PIPE=$2 MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'cross' in k or 'linear_3341' in k})"
done
exit 0
