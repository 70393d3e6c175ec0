"""One-update multi-hot backward (mh_embedding_bag_bwd_multi) on the bench's embedding_bag shape, for rocprofv3 --kernel-trace:
   python tools/dbg/bag_bwd_probe.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from models_amd import ops  # noqa: E402
from models_amd.synthetic import CRITEO_CARDINALITIES  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
B, D, mean_nnz = 65536, 64, 20
rng = np.random.default_rng(99)
g = torch.Generator(device=dev).manual_seed(9)
tabs = [torch.rand((int(v), D), device=dev, generator=g) - 0.5 for v in CRITEO_CARDINALITIES]
accs = [torch.full_like(t, 0.1) for t in tabs]
vs, os_ = [], []
for v in CRITEO_CARDINALITIES:
    lens = np.maximum(rng.poisson(mean_nnz, size=B), 1)
    offs = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    vs.append(torch.from_numpy(rng.integers(0, int(v), size=int(offs[-1])).astype(np.int32)).to(dev))
    os_.append(torch.from_numpy(offs).to(dev))
F = len(tabs)
grad = torch.rand((B, F * D), device=dev, generator=g) - 0.5
slot = [f * D for f in range(F)]
for _ in range(2):
    ops.embedding_bag_backward_multi(tabs, accs, vs, os_, grad, slot, "mean", optimizer="adagrad", lr=0.0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    ops.embedding_bag_backward_multi(tabs, accs, vs, os_, grad, slot, "mean", optimizer="adagrad", lr=0.0)
b.record()
torch.cuda.synchronize()
print("bag_bwd_multi ms", a.elapsed_time(b) / iters, "values", sum(int(v.numel()) for v in vs))
