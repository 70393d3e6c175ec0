#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for the current kernels (outputs under gpurun_out/refresh/):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
{
timeout 400 python bench.py
timeout 200 python bench.py --mode fwd --no-cpu-baseline --no-secondary
timeout 200 python bench.py --ids lognormal --no-cpu-baseline --no-secondary
MH_FORCE_DISTRIBUTED=1 timeout 200 python bench.py --steps 50 --warmup 8 --no-cpu-baseline
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --no-secondary
} 2>$O/bench_err.log | grep "^{" > $O/bench_lines.jsonl; echo "bench lines: $(wc -l < $O/bench_lines.jsonl) (expect 5)"
{
timeout 200 python bench.py --workload twotower --no-cpu-baseline
timeout 200 python bench.py --workload topk --no-cpu-baseline
timeout 200 python bench.py --workload dcn --mode fwd --no-cpu-baseline --steps 20 --warmup 3
timeout 300 python bench.py --workload dcn --no-cpu-baseline --steps 10 --warmup 3
} 2>/dev/null | grep '^{' > $O/secondary.jsonl; echo "secondary lines: $(wc -l < $O/secondary.jsonl) (expect 4)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --no-secondary > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/twotower -o w -- python bench.py --workload twotower --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dcn -o d -- python bench.py --workload dcn --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 600 python tools/pmc_traffic.py > $O/pmc_traffic.txt 2>&1; cp gpurun_out/pmc_traffic.json $O/ 2>/dev/null
tools/exp/gemm_lab 2>&1 | grep -v "NO \|128x64" > $O/gemm_lab.txt
ls -la $O
