#!/bin/bash
# Last GPU call of a round on a small remaining budget: sparse-update tests, A/B of the pass-0 load path, the default bench
# line, its rocprofv3 kernel stats, then the whole GPU suite if time remains.  Outputs under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 70 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bag_backward.py tests/test_gpu_fullsize.py tests/test_gpu_hygiene.py -x -q > $O/pytest_sparse.log 2>&1
rc=$?; tail -1 $O/pytest_sparse.log
if [ $rc -ne 0 ]; then echo "FASTLOAD tests failed: falling back to MERLIN_HIP_SORT_FASTLOAD=0"; export MERLIN_HIP_SORT_FASTLOAD=0; fi
for m in 0 1; do echo "fastload $m"; MERLIN_HIP_SORT_FASTLOAD=$m timeout 30 python tools/microbench.py embbwd 2>&1 | grep "embedding bwd"; done | tee $O/ab.txt
timeout 120 python bench.py 2>$O/bench_err.log | grep "^{" > $O/bench_line.jsonl; python -c "
import json; d=json.loads(open('$O/bench_line.jsonl').read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],4), d['roofline']['frac'], d['kernels_ms'].get('embedding_bwd'))"
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --no-secondary --steps 100 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
timeout 100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
