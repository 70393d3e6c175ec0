cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_r6; mkdir -p $O
PMC_CMD="python tools/gpu_scorer_arith.py B=32768 modes=bf16x6" PMC_FILTER="split_kernel,split_prepare" bash tools/pmc_busy.sh > $O/pmc_busy_scorer_bf16x6.txt 2>&1
PMC_CMD="python tools/gpu_scorer_arith.py B=32768 modes=bf16x3" PMC_FILTER="split_kernel,split_prepare" bash tools/pmc_busy.sh > $O/pmc_busy_scorer_bf16x3.txt 2>&1
cat $O/pmc_busy_scorer_bf16x6.txt $O/pmc_busy_scorer_bf16x3.txt | cut -c1-330
