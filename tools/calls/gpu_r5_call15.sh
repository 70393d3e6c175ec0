#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c15; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
( time timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" > $O/pytest_gpu.log ) 2>&1 | grep real; tail -15 $O/pytest_gpu.log | cut -c1-300
echo "== retrieval parity files under MERLIN_HIP_SCORER_ARITH=bf16x3"
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 1500 python -m pytest tests/test_gpu_retrieval.py tests/test_golden_vectors.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_bwd.py tests/test_gpu_models.py tests/test_gpu_scorer_split.py -m gpu -q 2>&1 | grep -v "$F" > $O/pytest_bf16x3.log; tail -25 $O/pytest_bf16x3.log | cut -c1-400
timeout 300 python tools/dbg/scorer_arith_table.py 2>&1 | grep -v "$F" | tee $O/scorer_bf16x3_error_table.txt
exit 0
