#!/usr/bin/env python
"""cold-table multi-hot forward: one 50 M-row x 64 table (12.8 GB), 65 536 bags of ~20 uniform ids"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from models_amd import ops
dev = torch.device("cuda"); B, D, rows = 65536, 64, 50_000_000
rng = np.random.default_rng(99); g = torch.Generator(device=dev).manual_seed(9)
big = torch.rand((rows, D), device=dev, generator=g)
lens = np.maximum(rng.poisson(20, size=B), 1); offs = np.zeros(B + 1, dtype=np.int32); np.cumsum(lens, out=offs[1:])
sets = [(torch.randint(0, rows, (int(offs[-1]),), dtype=torch.int32, device=dev, generator=g), torch.from_numpy(offs).to(dev)) for _ in range(3)]
o1 = torch.empty((B, D), device=dev)
for i in range(6): ops.embedding_bag(big, *sets[i % 3], "mean", out=o1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(18): ops.embedding_bag(big, *sets[i % 3], "mean", out=o1)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 18
cb = int(offs[-1]) * (D * 4 + 4) + B * (D * 4 + 8)
print(f"cold bag fwd: {ms * 1e3:.1f} us, {cb / ms / 1e6:.0f} GB/s, frac {cb / ms / 1e6 / 8000:.3f}")
