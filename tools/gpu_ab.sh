#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_models.py tests/test_gpu_backward.py tests/test_gpu_mlp_chain.py -x -q 2>&1 | tail -3
tools/exp/gemm_lab top | grep -v "BK32\|128x64\|2-stage\|4-stage"; tools/exp/gemm_lab dX | grep -v "BK32\|128x64\|2-stage\|4-stage"
python bench.py --no-cpu-baseline --no-secondary --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dlrm', round(d['ms_per_step'],4), d['kernels_ms'])"
