#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c16; rm -rf $O; mkdir -p $O
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_topk_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | cut -c1-300
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 900 python -m pytest tests/test_gpu_fullsize_bwd.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -4 | cut -c1-300
PMC_CMD="python bench.py --workload topk --steps 3 --warmup 2" PMC_FILTER="topk_,gemm_nt" bash tools/pmc_busy.sh 2>&1 | grep -v "$F" | tee $O/pmc_busy_topk.txt
MERLIN_HIP_SCORER_ARITH=bf16x3 PMC_CMD="python bench.py --workload twotower --steps 3 --warmup 2 --sustain 0 --no-cpu-baseline" PMC_FILTER="stream_split,split_prepare,stream_kernel,linear_fwd,gemm" bash tools/pmc_busy.sh 2>&1 | grep -v "$F" | tee $O/pmc_busy_twotower_bf16x3.txt
PMC_CMD="python tools/dbg/run_secondary.py embedding_bag iters=2" PMC_FILTER="bag_fwd,bag_expand,piece,radix" bash tools/pmc_busy.sh 2>&1 | grep -v "$F" | tee $O/pmc_busy_bag.txt
exit 0
