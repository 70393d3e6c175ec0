#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c22; rm -rf $O; mkdir -p $O
run() { n=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
}
for r in 1 2; do
run base X=1
run side_high MERLIN_HIP_SIDE_PRIORITY=-1
done
