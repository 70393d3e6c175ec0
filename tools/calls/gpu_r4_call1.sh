#!/bin/bash
# round 4, call 1: the whole GPU suite after the DLRM layout fix + RCCL one-rank communicator, smoke, default bench line, --gpus 2 refusal
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json; tail -5 $O/bench_default.err
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/gpus2.out 2> $O/gpus2.err; echo "gpus2 rc=$?"; tail -2 $O/gpus2.err
