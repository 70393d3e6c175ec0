#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python -m pytest tests/test_gpu_scorer_split.py tests/test_gpu_retrieval.py -m gpu -x -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error" | head -10 | cut -c1-300
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 900 python -m pytest tests/test_gpu_scorer_split.py tests/test_gpu_retrieval.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -q 2>&1 | grep -v "$F" | tail -5 | cut -c1-300
python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
print("scorer_fwd f32", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bench.run_scorer_fwd(dev).items()})
os.environ["MERLIN_HIP_SCORER_ARITH"] = "bf16x3"
print("scorer_fwd bf16x3", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bench.run_scorer_fwd(dev).items()})
PY
exit 0
