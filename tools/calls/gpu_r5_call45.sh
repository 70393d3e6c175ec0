#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
echo skip tests
for i in 1 2 3; do timeout 300 python tools/dbg/run_secondary.py topk 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d.get('ms_per_step', d.get('ms', 0)),4), d.get('bit_identical_to_f32_pipeline'), d.get('stages_ms'))"; done
exit 0
