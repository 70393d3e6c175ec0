#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "pipelined or graph_replayed or fit" 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -30 | cut -c1-300
P='import json,sys
d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), d["config"]["launch"], {k:round(v,4) for k,v in d["config"]["launch_probe"].items() if isinstance(v,float)})'
for i in 1 2; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>gpurun_out/r5c41_err.log | tail -1 | python -c "$P"
done
MERLIN_HIP_DW_DEFER=0 timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
tail -5 gpurun_out/r5c41_err.log | grep -v "$F"
exit 0
