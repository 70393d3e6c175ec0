#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c32
export MERLIN_HIP_GEMM_ARITH=bf16x3
PMC_CMD="python tools/dbg/run_secondary.py dcn_train" PMC_FILTER="gemm_split,gs_split,gs_reduce,gs_colsum,cross_bwd_pre" timeout 900 bash tools/pmc_busy.sh 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r5c32/pmc_busy_dcn_bf16x3.txt | cut -c1-400
rm -rf gpurun_out/pmc_busy1 gpurun_out/pmc_busy2
exit 0
