#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
P='import json,sys
d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["kernels_ms"].get("embedding_bwd"), d["kernels_ms"].get("dlrm_fused_bwd"), d["roofline"]["apply_phase"].get("ms"))'
for i in 1 2 3; do
echo "== new"; timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
echo "== old (d3df3ff)"; (cd _ab_old && timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P")
done
echo "== bf16x3-pinned tests"
MERLIN_HIP_SCORER_ARITH=bf16x3 MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_world2.py > gpurun_out/r5c40_pytest_bf16x3.log 2>&1; grep "passed\|failed\|FAILED" gpurun_out/r5c40_pytest_bf16x3.log | cut -c1-200
exit 0
