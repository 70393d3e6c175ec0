#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c10; rm -rf $O; mkdir -p $O
gpurun_in/lds_occ_lab | tee $O/lds_occ_lab.txt | awk 'NR==1 || /: [0-9]$/ || /per CU/' | uniq -f 5 | head -40
timeout 600 python tools/dbg/run_secondary.py fit_from_parquet > $O/fit.json 2> $O/fit.err; tail -c 1800 $O/fit.json; tail -3 $O/fit.err
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_compat.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 | cut -c1-300
exit 0
