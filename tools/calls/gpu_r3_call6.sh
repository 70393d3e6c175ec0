#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_sampling.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
timeout 300 python bench.py --workload twotower --negatives queue,popularity --no-cpu-baseline --steps 20 --warmup 3 --sustain 1 2>$O/err.txt | tail -1 > $O/tt.json
python -c "
import json; d = json.loads(open('$O/tt.json').read().strip().splitlines()[-1]); print('tt', round(d['ms_per_step'], 3), d['config']['launch']); print(json.dumps(d.get('negatives'), indent=1))"
tail -5 $O/err.txt
