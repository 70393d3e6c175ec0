#!/bin/bash
# One gpurun call that regenerates the round-6 evidence under profiles/ for the current kernels (outputs under gpurun_out/refresh6/):
#   gpurun --timeout 2700 -- 'bash tools/refresh_profiles_r6.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh6; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
MERLIN_HIP_SCORER_ARITH=bf16x3 MERLIN_HIP_GEMM_ARITH=bf16x3 timeout 1200 python -m pytest tests -m gpu -q \
  --deselect tests/test_gpu_bench_world2.py > $O/pytest_gpu_bf16x3.log 2>&1; tail -2 $O/pytest_gpu_bf16x3.log
MERLIN_HIP_SCORER_ARITH=f32 MERLIN_HIP_GEMM_ARITH=f32 timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_world2.py > $O/pytest_gpu_f32_chain.log 2>&1; tail -1 $O/pytest_gpu_f32_chain.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# HBM traffic of the HBM-bound launches FIRST, and into profiles/ on this box: the bench line below then carries roofline.traffic
# measured with exactly the kernels it times (bench.py refuses a file stamped with other kernel sources)
timeout 600 python tools/pmc_traffic.py > $O/pmc_traffic.txt 2>&1; cp gpurun_out/pmc_traffic.json $O/ 2>/dev/null && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; rm -rf gpurun_out/pmc_traffic
# the driver's command: stdout as the driver sees it (secondary / detail lines, the <= 4 KB headline LAST), then the other lines (headline only)
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_stdout.txt 2> $O/bench_default_err.log; tail -1 $O/bench_default_stdout.txt | wc -c
cp gpurun_out/bench_full_n1.json $O/bench_full_n1.json 2>/dev/null
{
MERLIN_HIP_GEMM_ARITH=f32 timeout 200 python bench.py --no-cpu-baseline --no-secondary --sustain 1 | tail -1
timeout 200 python bench.py --mode fwd --no-cpu-baseline --no-secondary --sustain 1 | tail -1
timeout 200 python bench.py --ids lognormal --no-cpu-baseline --no-secondary --sustain 1 | tail -1
MH_FORCE_DISTRIBUTED=1 timeout 200 python bench.py --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --sustain 1 | tail -1
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --sustain 1 | tail -1
} 2>$O/bench_err.log | grep "^{" > $O/bench_headlines.jsonl; echo "headline lines: $(wc -l < $O/bench_headlines.jsonl) (expect 5: f32-chain A/B, fwd, lognormal, forced sharding, torchrun)"
MH_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 \
  --master-port 29544 tests/bench_world2_harness.py --gpus 2 --steps 4 --warmup 2 --batch 2048 --sustain 0 --no-cpu-baseline \
  --shard-threshold 100000 --c4-rows 1000001 --tt-batches 2048,4096 2>$O/world2_err.log | grep "^{" > $O/bench_world2_shared_gpu_stdout.jsonl
echo "world-2 lines: $(wc -l < $O/bench_world2_shared_gpu_stdout.jsonl); last line bytes: $(tail -1 $O/bench_world2_shared_gpu_stdout.jsonl | wc -c)"
# kernel stats: the bench step (eager launches WITH side streams: durations include overlap), top-k, the multi-hot update, the tower layers
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python bench.py --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/topk -o k -- python bench.py --workload topk --no-cpu-baseline --steps 3 --warmup 2 --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bag -o b -- python tools/microbench.py bagbwd > $O/bag_microbench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tower -o w -- python tools/microbench.py tower > $O/tower_microbench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/twotower -o w -- python bench.py --workload twotower --tt-batch 65536 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dcn -o w -- python bench.py --workload dcn --steps 4 --warmup 2 --batches 2 --no-cpu-baseline --no-secondary --sustain 0 > /dev/null 2>&1
timeout 200 python tools/gpu_scorer_arith.py > $O/scorer_arith.log 2>&1; cat $O/scorer_arith.log
for w in train topk bag tower twotower dcn; do cp $(find $O/$w -name '*kernel_stats.csv' | head -1) $O/bench_${w}_kernel_stats.csv; rm -rf $O/$w; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/embada -o t -- python tools/microbench.py embada > $O/embada.log 2>&1
cp $(find $O/embada -name '*kernel_stats.csv' | head -1) $O/embada_kernel_stats.csv; rm -rf $O/embada
ls $O
