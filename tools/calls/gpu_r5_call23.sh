#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for pipe in 1 0; do for geo in 256x128 256x256; do
MERLIN_HIP_GEMM_SPLIT_PIPE=$pipe MERLIN_HIP_GEMM_SPLIT_GEO=$geo timeout 900 python -m pytest tests/test_gpu_gemm_split.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error" | head -6 | cut -c1-300
MERLIN_HIP_GEMM_SPLIT_PIPE=$pipe MERLIN_HIP_GEMM_SPLIT_GEO=$geo MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 600 python tools/dbg/run_secondary.py dcn_train 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipe=$pipe geo=$geo bf16x3 dcn_train', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms'].items() if 'cross' in k}, d.get('roofline',{}).get('frac'))"
done; done
exit 0
