#!/bin/bash
# smoke() + a short default-path bench line (graph capture, launch probe, roofline) on the current tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 40 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 45 python bench.py --no-secondary --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'], 4), round(d['roofline']['frac'], 3), d['kernels_ms'].get('embedding_bwd'), d['config']['launch_probe'])"
