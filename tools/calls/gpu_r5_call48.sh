#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
for g in 1 512 1024 1280 2048 4096; do echo "lds granule $g"; MERLIN_HIP_FUSED_LDS_GRANULE=$g timeout 300 python tools/dbg/fused_f_probe.py 23 24 25 26 27 28 29 2>&1 | grep -v "$F" | grep "^F ="; done | tee gpurun_out/r5c48_fused_lds_granule.txt
exit 0
