#!/bin/bash
# lab geometries of the six-term split GEMM on the DCN-v2 cross layer (LAB library)
export MERLIN_HIP_LIB=models_amd/csrc/lab/libmerlin_hip_lab.so
for geo in default 256x128 256x256k16 256x256k16s4; do
  echo "== MERLIN_HIP_GEMM_SPLIT_GEO=$geo"
  if [ $geo = default ]; then unset MERLIN_HIP_GEMM_SPLIT_GEO; else export MERLIN_HIP_GEMM_SPLIT_GEO=$geo; fi
  timeout 200 python tools/dbg/run_secondary.py cross_gemm 2>&1 | tail -1 | cut -c1-400
done
