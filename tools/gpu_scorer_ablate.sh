#!/bin/bash
# lab runs of the six-term scorer (LAB library; timing only for ablation bits >= 4)
export MERLIN_HIP_LIB=models_amd/csrc/lab/libmerlin_hip_lab.so
for cfg in "${@:-0:1}"; do
  lab=${cfg%%:*}; xt=${cfg##*:}
  echo "== MERLIN_HIP_SCORER_LAB=$lab MERLIN_HIP_SCORER_XT=$xt"
  MERLIN_HIP_SCORER_LAB=$lab MERLIN_HIP_SCORER_XT=$xt timeout 200 python tools/gpu_scorer_arith.py modes=bf16x6 2>&1 | grep bf16x6 | cut -c1-200
done
