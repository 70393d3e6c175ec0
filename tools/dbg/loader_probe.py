"""Where an epoch of models_amd.Loader (device-chunk mode) spends its time: device time of the chunk shuffle, host time of the
per-batch view slicing (GPU box)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import models_amd as mm  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
rows, B = 4_194_304, 65536
rng = np.random.default_rng(0)
from models_amd.synthetic import CRITEO_CONT_NAMES  # noqa: E402

cols = {n: rng.integers(0, v, size=rows).astype(np.int32) for n, v in bench._cat_columns()}
for n in CRITEO_CONT_NAMES:
    cols[n] = rng.random(rows, dtype=np.float32)
cols["label"] = rng.integers(0, 2, size=rows).astype(np.float32)
_, schema = bench.build_model(dev)
ld = mm.Loader(cols, schema, batch_size=B, shuffle=True, seed=1, device=dev, drop_last=True)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = iter(ld)
    first = next(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = 1
    for _ in it:
        n += 1
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"epoch {rep}: first batch host {1e3 * (t1 - t0):.1f} ms (+ device {1e3 * (t2 - t1):.1f}), remaining {n - 1} batches host {1e3 * (t3 - t2):.1f} ms "
          f"(+ device {1e3 * (t4 - t3):.1f}) = {1e3 * (t3 - t2) / max(n - 1, 1):.3f} ms per batch")
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for _ in ld:
    pass
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
x = torch.randint(0, 1000, (rows,), dtype=torch.int32, device=dev)
perm = torch.randperm(rows, device=dev)
for name, fn in (("index_select int32 4M", lambda: x.index_select(0, perm)), ("x[perm]", lambda: x[perm]), ("randperm 4M", lambda: torch.randperm(rows, device=dev))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, f"{1e2 * (time.perf_counter() - t0):.3f} ms")
