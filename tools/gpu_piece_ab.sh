#!/bin/bash
# sparse-update timing (tools/microbench.py embbwd), its tests, and an isolated kernel trace of the all-big / all-tiny table cases
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 python tools/microbench.py embbwd 2>&1 | grep "embedding bwd"
timeout 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bag_backward.py tests/test_gpu_fullsize.py tests/test_gpu_hygiene.py -x -q 2>&1 | tail -3
MB_ARGS="embbwd embbig" bash tools/gpu_embbwd_trace.sh
