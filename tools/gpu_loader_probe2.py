#!/usr/bin/env python
"""host time of the loader's next() calls (no model): where a chunk is staged the call takes as long as the staging costs the HOST"""
import os, shutil, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import pyarrow as pa, pyarrow.parquet as pq
import bench
import models_amd as mm
from models_amd.synthetic import CRITEO_CONT_NAMES
rows = 16777216; dev = torch.device("cuda", 0); B = 65536
rng = np.random.default_rng(4321)
tmp = tempfile.mkdtemp(prefix="mh_fit_")
try:
    cols = {n: rng.integers(0, v, size=rows).astype(np.int32) for n, v in bench._cat_columns()}
    for n in CRITEO_CONT_NAMES: cols[n] = rng.random(rows, dtype=np.float32)
    cols["label"] = rng.integers(0, 2, size=rows).astype(np.float32)
    path = os.path.join(tmp, "p.parquet")
    pq.write_table(pa.table(cols), path, row_group_size=1 << 20, compression="none", use_dictionary=False); del cols
    _, schema = bench.build_model(dev)
    ld = mm.Loader(path, schema, batch_size=B, shuffle=True, seed=1, device=dev, drop_last=True, device_resident_bytes=0)
    for ep in range(3):
        it = iter(ld); ts = []
        while True:
            t0 = time.perf_counter()
            try: next(it)
            except StopIteration: break
            ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        big = sorted(ts)[-4:]
        print(f"epoch {ep}: {len(ts)} batches, host ms per next(): median {sorted(ts)[len(ts)//2]:.3f}, largest {['%.1f' % b for b in big]}, sum {sum(ts):.1f}")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
