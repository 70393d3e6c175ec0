#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c21; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_world2.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
timeout 600 python tools/dbg/route_probe.py > $O/route_probe.txt 2>&1
cat $O/pytest.txt $O/route_probe.txt
