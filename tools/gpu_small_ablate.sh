#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MERLIN_HIP_LIB=$GRAFT_REPO_ROOT/models_amd/csrc/lab/libmerlin_hip_lab.so
for ab in 0 1 2 4 3 7; do
  MERLIN_HIP_BAG_SMALL_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sabl/$ab -o t -- python tools/microbench.py bagbwd > /dev/null 2>&1
done
