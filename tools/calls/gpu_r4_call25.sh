#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c25; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_models.py tests/test_golden_vectors.py -m gpu -x -q 2>&1 | tail -12 > $O/pytest.txt
cat $O/pytest.txt
python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 20 --sustain 1 --launch recorded 2>&1 | tail -3 | cut -c1-600
