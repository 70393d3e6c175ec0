#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_world2.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt
