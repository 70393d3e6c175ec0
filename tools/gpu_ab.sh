#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_models.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3
python bench.py --workload dcn --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcn', d['ms_per_step'], d['kernels_ms'])"
