#!/bin/bash
# One gpurun call: gpu tests, smoke, the bench lines and a rocprof kernel summary of the default bench.  Outputs under gpurun_out/round/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/round; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
{
timeout 200 python bench.py --workload twotower --no-cpu-baseline
timeout 200 python bench.py --workload topk --no-cpu-baseline
timeout 200 python bench.py --workload dcn --mode fwd --no-cpu-baseline
} 2>$O/secondary.err | grep '^{' > $O/secondary.jsonl; cat $O/secondary.jsonl | cut -c1-1500
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --steps 40 --warmup 5 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
f=$(ls $O/train/*kernel_stats.csv | head -1); column -s, -t $f | cut -c1-60,150-260 | head -30
