#!/bin/bash
# multi-hot update: correctness subset, then the kernel trace of tools/microbench.py bagbwd ($1 = tag)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/bagbwd_$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bag_backward.py tests/test_gpu_embedding.py -x -q 2>&1 | tail -5
timeout 300 python tools/microbench.py bagbwd 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python tools/microbench.py bagbwd > $O/log.txt 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-150'
