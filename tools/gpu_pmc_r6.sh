#!/bin/bash
# SQ / MFMA-busy / LDS counters of round 6's new kernels (two PMC passes each, kernel-trace only): outputs gpurun_out/pmc_r6/*.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r6; mkdir -p $O
PMC_CMD="python tools/gpu_scorer_arith.py B=32768 modes=bf16x6" PMC_FILTER="split_kernel,split_prepare" bash tools/pmc_busy.sh > $O/pmc_busy_scorer_bf16x6.txt 2>&1
PMC_CMD="python tools/gpu_scorer_arith.py B=32768 E=64 modes=bf16x6" PMC_FILTER="stream_split,split_prepare" bash tools/pmc_busy.sh > $O/pmc_busy_scorer_bf16x6_e64.txt 2>&1
PMC_CMD="python tools/dbg/run_secondary.py cross_gemm" PMC_FILTER="gemm_split,gs_split" bash tools/pmc_busy.sh > $O/pmc_busy_dcn_bf16x6.txt 2>&1
PMC_CMD="python tools/dbg/run_secondary.py topk" PMC_FILTER="topk_filter_bf16x3,topk_sort_merge,topk_finalize" bash tools/pmc_busy.sh > $O/pmc_busy_topk_two_level.txt 2>&1
PMC_CMD="python tools/microbench.py tower" PMC_FILTER="tower_" bash tools/pmc_busy.sh > $O/pmc_busy_tower.txt 2>&1
tail -n +1 $O/*.txt | cut -c1-400
