#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests -m gpu -q -k "fused or interaction or dlrm or c4 or graph" 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -10 | cut -c1-300
timeout 600 python tools/dbg/run_secondary.py c4_one_gpu 2>&1 | grep -v "$F" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('roofline_dlrm_fused_bwd'), {k:v for k,v in d.get('kernels_ms',{}).items() if 'fused' in k})"
timeout 300 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernels_ms'].get('dlrm_fused_bwd'), d['kernels_ms'].get('dlrm_fused_fwd'))"
exit 0
