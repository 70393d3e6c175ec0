#!/bin/bash
# isolated per-kernel durations of the sparse update (rocprofv3 kernel trace of tools/microbench.py embbwd)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/embbwd
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/embbwd -o t -- python tools/microbench.py ${MB_ARGS:-embbwd} > gpurun_out/embbwd/log.txt 2>&1
grep "embedding" gpurun_out/embbwd/log.txt
