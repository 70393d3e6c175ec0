#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for the current kernels (outputs under gpurun_out/refresh/):
#   gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# HBM traffic of the HBM-bound launches FIRST, and into profiles/ on this box: the bench lines below then carry roofline.traffic
# measured with exactly the kernels they time (bench.py refuses a file stamped with other kernel sources)
timeout 600 python tools/pmc_traffic.py > $O/pmc_traffic.txt 2>&1; cp gpurun_out/pmc_traffic.json $O/ 2>/dev/null && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; rm -rf gpurun_out/pmc_traffic
{
timeout 500 python bench.py
timeout 200 python bench.py --mode fwd --no-cpu-baseline --no-secondary --sustain 1
timeout 200 python bench.py --ids lognormal --no-cpu-baseline --no-secondary --sustain 1
MH_FORCE_DISTRIBUTED=1 timeout 200 python bench.py --steps 50 --warmup 8 --no-cpu-baseline --sustain 1
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --sustain 1
} 2>$O/bench_err.log | grep "^{" > $O/bench_lines.jsonl; echo "bench lines: $(wc -l < $O/bench_lines.jsonl) (expect 5)"
{
timeout 300 python bench.py --workload twotower --negatives queue,popularity --no-cpu-baseline --sustain 1
timeout 200 python bench.py --workload topk --no-cpu-baseline
timeout 200 python bench.py --workload dcn --mode fwd --no-cpu-baseline --steps 20 --warmup 3 --sustain 0
timeout 300 python bench.py --workload dcn --no-cpu-baseline --steps 10 --warmup 3 --sustain 0
} 2>/dev/null | grep '^{' > $O/secondary.jsonl; echo "secondary lines: $(wc -l < $O/secondary.jsonl) (expect 4)"
# isolated, SINGLE-STREAM kernel traces of the dominant launch and the fused kernels (tools/microbench.py issues on one stream)
for sel in embada emb1m fused; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$sel -o t -- python tools/microbench.py $sel > $O/$sel.log 2>&1
  cp $(ls $O/$sel/*kernel_trace.csv | head -1) $O/${sel}_kernel_trace.csv; cp $(ls $O/$sel/*kernel_stats.csv | head -1) $O/${sel}_kernel_stats.csv
  rm -rf $O/$sel
done
# the bench step itself (eager launches WITH side streams: kernel durations include overlap) and the secondary workloads
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/twotower -o w -- python bench.py --workload twotower --no-cpu-baseline --steps 5 --warmup 2 --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dcn -o d -- python bench.py --workload dcn --no-cpu-baseline --steps 3 --warmup 1 --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/topk -o k -- python bench.py --workload topk --no-cpu-baseline --steps 3 --warmup 2 --sustain 0 > /dev/null 2>&1
for w in train twotower dcn topk; do cp $(ls $O/$w/*kernel_stats.csv | head -1) $O/bench_${w}_kernel_stats.csv; rm -rf $O/$w; done
# warmed-up forward scorer: stream vs tiled kernel on the same inputs, and the launch-mode A/B of the headline step
timeout 300 python tools/dbg/scorer_probe.py > $O/scorer_probe.txt 2>&1
timeout 300 python tools/dbg/route_probe.py > $O/route_probe.txt 2>&1
ls -la $O
