#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c19; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_edges.py tests/test_gpu_dense.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
run() { n=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
}
for r in 1 2; do
run tail_auto X=1
run tail_off MERLIN_HIP_TAIL=0
run tail_on MERLIN_HIP_TAIL=1
done
cat $O/pytest.txt
