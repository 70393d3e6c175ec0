#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for ab in 0 1 2 4 3 6 7; do
echo "== ablate $ab"
MERLIN_HIP_SCORER_ABLATE=$ab MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py twotower batch=65536 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], {k:v for k,v in d.get('kernels_ms',{}).items() if 'softmax' in k})"
done
exit 0
