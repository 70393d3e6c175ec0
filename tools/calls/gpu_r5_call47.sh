#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c47; rm -rf $O; mkdir -p $O
{
timeout 600 python bench.py
timeout 200 python bench.py --mode fwd --no-cpu-baseline --no-secondary --sustain 1
timeout 200 python bench.py --ids lognormal --no-cpu-baseline --no-secondary --sustain 1
MH_FORCE_DISTRIBUTED=1 timeout 200 python bench.py --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --sustain 1
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 8 --no-cpu-baseline --no-secondary --sustain 1
} 2>$O/bench_err.log | grep "^{" > $O/bench_lines.jsonl; echo "bench lines: $(wc -l < $O/bench_lines.jsonl) (expect 5)"
MH_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 \
  --master-port 29544 tests/bench_world2_harness.py --gpus 2 --steps 4 --warmup 2 --batch 2048 --sustain 0 --no-cpu-baseline \
  --shard-threshold 100000 --c4-rows 1000001 --tt-batches 2048,4096 2>$O/world2_err.log | grep "^{" > $O/bench_world2_shared_gpu.jsonl
echo "world-2 lines: $(wc -l < $O/bench_world2_shared_gpu.jsonl) (expect 1)"
head -1 $O/bench_lines.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); rl=d['roofline']; print(d['ms_per_step'], rl['kernel'][:60], rl['frac'], rl['traffic'], rl['avg_launch_ms'], rl['whole_update']['frac'])"
for f in $O/bench_lines.jsonl; do python - <<PY
import json
for l in open("$f"):
    d=json.loads(l); print(d['config'].get('mode'), d['ms_per_step'], (d.get('roofline') or {}).get('frac'))
PY
done
exit 0
