#!/bin/bash
# split top-k: the bit-exactness suite, the bench line and a kernel trace ($1 = tag)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/topk_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_topk_split.py tests/test_gpu_retrieval.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --workload topk --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('topk ms', d['ms_per_step'], d.get('bit_identical_to_f32_pipeline'))"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python bench.py --workload topk --no-cpu-baseline --steps 3 --warmup 2 --sustain 0 > /dev/null 2>&1
