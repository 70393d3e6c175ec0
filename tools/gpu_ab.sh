#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_route.py tests/test_gpu_hygiene.py tests/test_gpu_backward.py -x -q 2>&1 | tail -2
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), d['config']['launch'][:12])"; }
MH_FORCE_DISTRIBUTED=1 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --eager 2>/dev/null | show forced_dist_eager_streams
MH_FORCE_DISTRIBUTED=1 MERLIN_HIP_SIDE_STREAMS=0 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --eager 2>/dev/null | show forced_dist_eager_1stream
MH_FORCE_DISTRIBUTED=1 python bench.py --steps 100 --warmup 8 --no-cpu-baseline 2>/dev/null | show forced_dist_graph
