#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c32; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp_chain.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_world2.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest.txt
cat $O/pytest.txt | cut -c1-220
run() { n=$1; shift
  env "$@" python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', round(d['ms_per_step'],4), d['config']['launch'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
}
for r in 1 2 3; do
run lazy X=1
run eagerloss MERLIN_HIP_LAZY_BCE=0
done
