#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c38; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_world2.py  tests/test_gpu_models.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $O/pytest.txt
cut -c1-300 $O/pytest.txt | tail -8
for r in 1 2; do
MH_FORCE_DISTRIBUTED=1 python bench.py --steps 100 --warmup 15 --no-cpu-baseline --sustain 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced-shard W=1', round(d['ms_per_step'],4), d['config']['launch'])"
done
