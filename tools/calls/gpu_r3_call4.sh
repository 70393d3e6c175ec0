#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_hygiene.py tests/test_gpu_backward.py tests/test_gpu_world2.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary --sustain 2 2>$O/bench_err.txt | tail -1 > $O/bench.json
python -c "
import json; d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'], 4), d['config']['launch'], d['config']['launch_probe'], d['sustained'], d['roofline']['frac'], d['roofline']['traffic_source'])"
tail -5 $O/bench_err.txt
