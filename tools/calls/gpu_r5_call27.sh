#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
mkdir -p gpurun_out/r5c27
timeout 600 python -m pytest tests/test_gpu_bag_backward.py tests/test_gpu_embedding.py tests/test_gpu_fullsize_bwd.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -30 | cut -c1-300
for t in 0 2 3 5; do echo "tile_log2=$t"; MERLIN_HIP_APPLY_TILE_LOG2=$t timeout 300 python tools/dbg/bag_bwd_probe.py 6 2>&1 | grep -v "$F" | tail -1; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c27/prof -- python tools/dbg/bag_bwd_probe.py 4 2>&1 | grep -v "$F" | tail -1
f=$(find gpurun_out/r5c27/prof -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r5c27/bag_bwd_multi_kernel_stats.csv
rm -rf gpurun_out/r5c27/prof
timeout 600 python bench.py --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernels_ms'].get('embedding_bwd'), d['roofline'].get('apply_phase',{}).get('ms'))"
exit 0
