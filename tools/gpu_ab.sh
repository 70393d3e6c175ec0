#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['value']/1e6,2), d['step_ms'])"; }
python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | show graph
python bench.py --no-cpu-baseline --no-secondary --steps 200 --eager 2>/dev/null | show eager_streams
MERLIN_HIP_SIDE_STREAMS=0 python bench.py --no-cpu-baseline --no-secondary --steps 200 --eager 2>/dev/null | show eager_nostreams
