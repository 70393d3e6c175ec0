#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for ab in 2 0 2; do
echo "== ablate $ab (pipe 0)"
MERLIN_HIP_GEMM_SPLIT_ABLATE=$ab MERLIN_HIP_GEMM_SPLIT_PIPE=0 MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'cross' in k or 'linear_3341' in k})"
done
exit 0
