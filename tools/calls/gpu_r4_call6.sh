#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "topk or mean or dcn_variants" > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
for f in stream tiled stream tiled; do
MERLIN_HIP_TOPK_FILTER=$f python bench.py --workload topk --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))" | tee -a $O/topk_ab.txt
done
