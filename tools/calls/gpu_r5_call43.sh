#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_route.py tests/test_gpu_models.py tests/test_gpu_distributed.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -20 | cut -c1-300
timeout 600 python tools/dbg/route_probe.py 2>&1 | grep -v "$F" | cut -c1-400
MERLIN_HIP_DETERMINISTIC=1 timeout 600 python tools/dbg/route_probe.py 2>&1 | grep -v "$F" | cut -c1-400 | head -2
exit 0
