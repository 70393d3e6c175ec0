#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c23; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bag_backward.py tests/test_gpu_models.py tests/test_gpu_edges.py tests/test_golden_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_route.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
for r in 1 2; do
python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 30 --sustain 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('new', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'] if 'avg_launch_ms' in d['roofline'] else d['roofline'], {k:round(v,4) for k,v in d['config']['launch_probe'].items() if isinstance(v,float)})
" | tee -a $O/ab.txt
done
timeout 200 python tools/microbench.py embada 2>&1 | tail -3 | tee -a $O/ab.txt
timeout 200 python tools/microbench.py emb1m 2>&1 | tail -3 | tee -a $O/ab.txt
