#!/bin/bash
# A/B of the two GEMM cores on the DLRM train step (per-kernel times)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  unset MERLIN_HIP_GEMM_V1; if [ "$v" = "1" ]; then export MERLIN_HIP_GEMM_V1=1; fi
  python bench.py --no-cpu-baseline --no-secondary --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v1=$v dlrm', round(d['ms_per_step'],4), d['kernels_ms']['linear_bwd_415x128'], d['kernels_ms']['linear_415x128'])"
done
