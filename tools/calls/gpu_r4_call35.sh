#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c35; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_world2.py tests/test_gpu_bench_world2.py tests/test_gpu_route.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/pytest.txt
cut -c1-300 $O/pytest.txt | tail -20
export MH_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 tests/bench_world2_harness.py --gpus 2 --steps 20 --warmup 5 --batch 16384 --sustain 0 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/line.jsonl
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29612 tests/bench_world2_harness.py --gpus 2 --steps 20 --warmup 5 --batch 16384 --sustain 0 --no-cpu-baseline --ids lognormal 2>/dev/null | grep "^{" >> $O/line.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r4c35/line.jsonl'):
    d=json.loads(l); r=d['sharded']
    print(d['n_gpus'], round(d['ms_per_step'],3), r.get('bytes_sent_per_rank_per_step'), r.get('max_abs_err_vs_w1_oracle'))
PY
