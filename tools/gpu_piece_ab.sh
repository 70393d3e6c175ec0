#!/bin/bash
# A/B of the piece kernel modes (MERLIN_HIP_PIECE_MODE=0 per-piece index loads, 1 = lane-held indices) + the sparse-update tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do echo "piece mode $m"; MERLIN_HIP_PIECE_MODE=$m timeout 120 python tools/microbench.py embbwd 2>&1 | grep "embedding bwd"; done
timeout 300 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bag_backward.py tests/test_gpu_fullsize.py tests/test_gpu_hygiene.py -x -q 2>&1 | tail -3
