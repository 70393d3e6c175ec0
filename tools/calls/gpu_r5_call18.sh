#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c18; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error" | head -30 | cut -c1-300
for m in f32 bf16x3; do
MERLIN_HIP_GEMM_ARITH=$m timeout 600 python tools/dbg/run_secondary.py dcn_train 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$m dcn_train', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms'].items() if 'cross' in k or 'linear' in k})"
done
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 900 python -m pytest tests/test_gpu_fullsize_bwd.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed" | head -8 | cut -c1-300
exit 0
