#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4))"; }
for i in 1 2 3; do
MERLIN_HIP_SORT_AFTER_GATHER=1 python bench.py --no-cpu-baseline --no-secondary --steps 300 --eager 2>/dev/null | show sort_after_gather
MERLIN_HIP_SORT_AFTER_GATHER=0 python bench.py --no-cpu-baseline --no-secondary --steps 300 --eager 2>/dev/null | show sort_at_start
done
python bench.py --no-cpu-baseline --no-secondary --steps 300 --launch graph 2>/dev/null | show graph
