#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c12; rm -rf $O; mkdir -p $O
for f in stream tiled stream tiled; do
MERLIN_HIP_TOPK_FILTER=$f python bench.py --workload topk --no-cpu-baseline --steps 8 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" | tee -a $O/topk_ab.txt
done
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
