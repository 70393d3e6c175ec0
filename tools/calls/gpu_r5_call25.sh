#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F='amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl'
timeout 600 python -m pytest tests/test_gpu_bag_backward.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E \|passed\|failed\|Error\|FAILED" | head -30 | cut -c1-300
MERLIN_HIP_SCORER_ARITH=bf16x3 timeout 900 python -m pytest tests/test_gpu_scorer_split.py tests/test_gpu_retrieval.py tests/test_gpu_fullsize.py tests/test_gpu_models.py -m gpu -q 2>&1 | grep -v "$F" | grep "^E  \|passed\|failed\|FAILED" | head -30 | cut -c1-300
timeout 600 python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
r = bench.run_embedding_bag(dev)
print(json.dumps({k: r[k] for k in ("bwd_adagrad", "bwd_adagrad_one_update", "fwd")}, indent=1))
PY
exit 0
