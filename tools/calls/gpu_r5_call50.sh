#!/bin/bash
# the driver's N = 2 command at its DEFAULT sizes (configs[3] with the 100 M-row table, TwoTower 32 K / 64 K, DCN-v2 at 64 K per rank), two
# ranks sharing this one GPU over gloo: does every branch of the line survive the real shapes?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c50; rm -rf $O; mkdir -p $O
( time MH_BENCH_SHARED_GPU=1 MASTER_ADDR=127.0.0.1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 \
  --master-port 29546 tests/bench_world2_harness.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 2>$O/err.log | grep "^{" > $O/line.jsonl ) 2>&1 | tail -3
wc -l $O/line.jsonl; grep -v "amdgpu.ids\|socket.cpp" $O/err.log | tail -8 | cut -c1-300
python - <<PY
import json
l=open("$O/line.jsonl").readline()
if l:
    d=json.loads(l); print(d['n_gpus'], d['ms_per_step'], d['config'].get('launch'))
    for k,v in d['secondary'].items():
        if isinstance(v,dict): print(k, {kk:v[kk] for kk in ('ms_per_step','value','n_gpus','error','skipped','deadline') if kk in v})
PY
exit 0
