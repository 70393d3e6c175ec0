#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-secondary --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dlrm', round(d['ms_per_step'],4), d['kernels_ms'])"
