#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c13; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for xt in 1 2; do
echo "== XT=$xt"
MERLIN_HIP_SCORER_XT=$xt timeout 900 python -m pytest tests/test_gpu_scorer_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -25 | cut -c1-300
MERLIN_HIP_SCORER_XT=$xt MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 300 python tools/dbg/run_secondary.py twotower batch=65536 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bf16x3 XT=$xt', round(d['ms_per_step'],3), {k:v for k,v in d['kernels_ms'].items() if 'softmax' in k})"
done
timeout 300 python tools/dbg/loader_probe.py 2>&1 | grep -v "$F" | head -45 | cut -c1-200 | tee $O/loader_probe.txt
for f in 27 28; do
WORK="python tools/dbg/fused_f_probe.py $f" PASSES="sq2 tcc" bash tools/dbg/pmc_kernels.sh 2>&1 | grep "fused_bwd\|fused_fwd" | cut -c1-400 | sed "s/^/F=$f /"
done
exit 0
