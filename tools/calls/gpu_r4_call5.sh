#!/bin/bash
# A/B of the side-stream schedule of the DLRM step (same box, alternating, 2 rounds)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5; rm -rf $O; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
}
for r in 1 2; do
run base X=1
run dw_early MERLIN_HIP_DW_EARLY=1
run dw_own_stream MERLIN_HIP_SIDE_ALIAS=sparse=sort
run dw_early_own MERLIN_HIP_DW_EARLY=1 MERLIN_HIP_SIDE_ALIAS=sparse=sort
run three_streams MERLIN_HIP_SIDE_ALIAS=none
done
