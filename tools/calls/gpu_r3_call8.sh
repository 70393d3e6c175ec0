#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_hygiene.py tests/test_gpu_fullsize.py tests/test_gpu_world2.py -m gpu -q -x 2>&1 | tail -12 | tee $O/pytest.txt
for rep in 1 2; do
for v in 0 1; do
  MERLIN_HIP_ROW_PIPELINE=$v timeout 300 python bench.py --no-cpu-baseline --no-secondary --sustain 1 2>/dev/null | tail -1 > $O/bench_$v.json
  python -c "
import json; d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('row pipeline $v:', round(d['ms_per_step'], 4), d['config']['launch'], d['config']['launch_probe'])"
done; done
