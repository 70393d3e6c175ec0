#!/bin/bash
# sweep of the multi-hot walk kernel's lab knobs (LAB library): rows in flight x stretch length
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MERLIN_HIP_LIB=$GRAFT_REPO_ROOT/models_amd/csrc/lab/libmerlin_hip_lab.so
for cfg in ${SWEEP:-"4 64" "8 64" "16 64" "4 16" "8 16" "8 32" "8 128" "8 256"}; do
  set -- $cfg
  MERLIN_HIP_APPLY_WALK_ROWS=$1 MERLIN_HIP_APPLY_WALK_T=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep/r$1_t$2 -o t -- python tools/microbench.py bagbwd > /dev/null 2>&1
done
