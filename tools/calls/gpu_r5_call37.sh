#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for pipe in 1 0; do
echo "== tests scorer pipe $pipe"
MERLIN_HIP_SCORER_PIPE=$pipe timeout 900 python -m pytest tests/test_gpu_scorer_split.py tests/test_gpu_gemm_split.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
done
for pipe in 0 1 0 1; do
echo "== twotower b64k bf16x3 scorer pipe $pipe"
MERLIN_HIP_SCORER_PIPE=$pipe MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py twotower batch=65536 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'softmax' in k or 'scorer' in k or 'inbatch' in k})"
done
MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('dcn', d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'cross' in k or 'linear_3341' in k})"
exit 0
