#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 20 --sustain 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['avg_launch_ms'], json.dumps(r.get('apply_phase'))[:700])"
