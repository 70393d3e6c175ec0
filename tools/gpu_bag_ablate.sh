#!/bin/bash
# ablation of the multi-hot walk kernel (LAB library; results are wrong by construction): which resource bounds it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export MERLIN_HIP_LIB=$GRAFT_REPO_ROOT/models_amd/csrc/lab/libmerlin_hip_lab.so
for ab in ${ABL:-0 1 2 4 8 16 3 7 15 31 12 27}; do
  echo "== ablate $ab  T=${MERLIN_HIP_APPLY_WALK_T:-64}"
  MERLIN_HIP_APPLY_WALK_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/$ab -o t -- python tools/microbench.py bagbwd > /dev/null 2>&1
  f=$(find gpurun_out/abl/$ab -name "*kernel_stats.csv" | head -1); grep -h "piece_walk\|carry_apply" $f | awk -F'","' '{printf "%s avg %.1f us\n", substr($1,2,40), $4/1000}'
done
