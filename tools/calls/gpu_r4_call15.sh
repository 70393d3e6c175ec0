#!/bin/bash
# the merged prepared branch: full suite twice (synchronisation change), then A/B of the tail work on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c15; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu_1.log 2>&1; grep -E "passed|failed" $O/pytest_gpu_1.log | tail -1
timeout 900 python -m pytest tests -m gpu -q -k "models or chain or backward or world2 or hygiene or recorder" > $O/pytest_gpu_2.log 2>&1; grep -E "passed|failed" $O/pytest_gpu_2.log | tail -1
run() { n=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
}
for r in 1 2 3; do
run tail_on MERLIN_HIP_TAIL=1
run tail_off MERLIN_HIP_TAIL=0
done
