#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c12; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_scorer_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -25 | cut -c1-400
timeout 300 python tools/dbg/fused_f_probe.py 2>&1 | grep -v "$F" | tee $O/fused_f_probe.txt
timeout 300 python tools/dbg/loader_probe.py 2>&1 | grep -v "$F" | tee $O/loader_probe.txt
for m in f32 bf16x3; do
MERLIN_HIP_SCORER_ARITH=$m timeout 300 python tools/dbg/run_secondary.py twotower batch=65536 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],3), {k:v for k,v in d['kernels_ms'].items() if 'softmax' in k})"
done
exit 0
