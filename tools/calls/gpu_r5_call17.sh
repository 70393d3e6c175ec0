#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c17; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 900 python -m pytest tests/test_gpu_fullsize_bwd.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed" | head -20 | cut -c1-300
ab() { timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); lp=d['config']['launch_probe']; k=d['kernels_ms']
print('$1', 'step', round(d['ms_per_step'],4), d['config']['launch'][:10], 'graph', round(lp['hipGraph_replay_ms'],4), 'seg', round(lp.get('segmented_replay_ms',0),4), 'rec', round(lp.get('recorded_replay_ms',0),4), 'eager', round(lp['eager_side_streams_ms'],4), 'linbwd', k.get('linear_bwd_415x128'))"; }
for rep in 1 2; do
ab "base"
MERLIN_HIP_ASTAT=1 ab "astat"
MERLIN_HIP_DW_LATE=1 ab "late"
MERLIN_HIP_ASTAT=1 MERLIN_HIP_DW_LATE=1 ab "astat+late"
done
exit 0
