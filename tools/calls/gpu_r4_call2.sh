#!/bin/bash
# round 4, call 2: fused cross backward + slab reductions + vector eltwise: tests, DCN step (kernel stats), default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "cross or slab or eltwise or c5 or dcn or DCN or golden or wide" > $O/pytest_sel.log 2>&1; tail -8 $O/pytest_sel.log
timeout 600 python bench.py --workload dcn --no-cpu-baseline --steps 6 --warmup 2 --sustain 0 > $O/dcn.json 2> $O/dcn.err; tail -c 1500 $O/dcn.json; tail -3 $O/dcn.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dcnprof -o t -- python bench.py --workload dcn --no-cpu-baseline --steps 6 --warmup 2 --sustain 0 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
f=$(ls $O/dcnprof/*kernel_stats.csv | head -1); column -s, -t $f | cut -c1-70,150-230 | head -24
