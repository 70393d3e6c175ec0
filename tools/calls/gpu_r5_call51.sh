#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c51
MERLIN_HIP_GEMM_ARITH=bf16x3 PMC_CMD="python tools/dbg/run_secondary.py dcn_train" PMC_FILTER="gemm_split,gs_split" timeout 900 bash tools/pmc_busy.sh 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r5c51/pmc_busy_dcn_bf16x3_final.txt
rm -rf gpurun_out/pmc_busy1 gpurun_out/pmc_busy2
MERLIN_HIP_SCORER_ARITH=bf16x3 PMC_CMD="python tools/dbg/run_secondary.py twotower batch=65536" PMC_FILTER="stream_split,split_prepare,stream_kernel" timeout 900 bash tools/pmc_busy.sh 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r5c51/pmc_busy_twotower_bf16x3_final.txt
rm -rf gpurun_out/pmc_busy1 gpurun_out/pmc_busy2
cut -c1-200 gpurun_out/r5c51/pmc_busy_dcn_bf16x3_final.txt | head -8; cut -c1-200 gpurun_out/r5c51/pmc_busy_twotower_bf16x3_final.txt | head -8
exit 0
