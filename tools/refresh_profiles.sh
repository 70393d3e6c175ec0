#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for the current kernels:
#   tools/refresh_profiles.sh   (run on the GPU box from the repo root; outputs under gpurun_out/refresh/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
{
python bench.py
python bench.py --mode fwd --no-cpu-baseline
python bench.py --ids lognormal --no-cpu-baseline
MH_FORCE_DISTRIBUTED=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline
} 2>$O/bench_err.log | grep "^{" > $O/bench_lines.jsonl; test $(wc -l < $O/bench_lines.jsonl) -eq 4 || echo "!! a bench.py invocation printed no JSON line (see $O/bench_err.log)"
{
python bench.py --workload twotower
python bench.py --workload topk
python bench.py --workload dcn --mode fwd
python bench.py --workload dcn
} 2>/dev/null | grep '^{' > $O/secondary.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -o f -- python bench.py --mode fwd --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/twotower -o w -- python bench.py --workload twotower --steps 5 --warmup 2 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tools/pmc_traffic.sh embbwd inter > $O/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_fetch/*counter_collection.csv $O/pmc_fetch_counter_collection.csv 2>/dev/null
cp gpurun_out/pmc_write/*counter_collection.csv $O/pmc_write_counter_collection.csv 2>/dev/null
ls -la $O
