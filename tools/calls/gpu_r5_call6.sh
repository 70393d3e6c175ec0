#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c6; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_topk_split.py tests/test_gpu_loader.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -30 > $O/pytest.txt
cut -c1-300 $O/pytest.txt | tail -12
run() { timeout 300 python bench.py --workload topk --steps 8 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d.get('bit_identical_to_f32_pipeline'))"; }
run default
MERLIN_HIP_TOPK_XCD_MAP=0 run "plain map"
MERLIN_HIP_TOPK_GROWTH=8 run "growth 8"
MERLIN_HIP_TOPK_GROWTH=3 run "growth 3"
timeout 600 python tools/dbg/run_secondary.py embedding_bag > $O/bag.json 2> $O/bag.err; tail -c 3000 $O/bag.json; tail -3 $O/bag.err
timeout 600 python tools/dbg/run_secondary.py fit_from_parquet > $O/fit.json 2> $O/fit.err; tail -c 2000 $O/fit.json; tail -3 $O/fit.err
exit 0
