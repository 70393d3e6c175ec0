#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5c44_pytest_gpu.log 2>&1; grep "passed\|failed\|FAILED\|^E  " gpurun_out/r5c44_pytest_gpu.log | head -20 | cut -c1-250
exit 0
