#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c34; rm -rf $O; mkdir -p $O
export MH_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tests/bench_world2_harness.py --gpus 2 --steps 20 --warmup 5 --batch 16384 --sustain 0 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/line.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29612 tests/bench_world2_harness.py --gpus 2 --steps 20 --warmup 5 --batch 16384 --sustain 0 --no-cpu-baseline --ids lognormal 2>/dev/null | grep "^{" >> $O/line.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r4c34/line.jsonl'):
    d=json.loads(l); r=d['sharded']
    print(d['n_gpus'], round(d['ms_per_step'],3), d['config']['parallelism'], r.get('bytes_sent_per_rank_per_step'), r.get('max_abs_err_vs_w1_oracle'), r.get('time_split_ms_serialised'))
PY
