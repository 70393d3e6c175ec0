#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
MERLIN_HIP_GEMM_SPLIT_PIPE=1 timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -q -x 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
for pipe in 0 1 0 1; do
echo "== pipe $pipe"
MERLIN_HIP_GEMM_SPLIT_PIPE=$pipe MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'cross' in k or 'linear' in k})"
done
exit 0
