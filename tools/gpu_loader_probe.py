#!/usr/bin/env python
"""Streaming fit probe: Criteo-shaped Parquet of ROWS rows, device-resident cache OFF: loader alone, fit, resident step rate."""
import math, os, shutil, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import pyarrow as pa, pyarrow.parquet as pq
import bench
import models_amd as mm
from models_amd.synthetic import CRITEO_CONT_NAMES

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_194_304
dev = torch.device("cuda", 0)
B = 65536
rng = np.random.default_rng(4321)
tmp = tempfile.mkdtemp(prefix="mh_fit_")
try:
    cols = {n: rng.integers(0, v, size=rows).astype(np.int32) for n, v in bench._cat_columns()}
    for n in CRITEO_CONT_NAMES:
        cols[n] = rng.random(rows, dtype=np.float32)
    cols["label"] = rng.integers(0, 2, size=rows).astype(np.float32)
    path = os.path.join(tmp, "part0.parquet")
    pq.write_table(pa.table(cols), path, row_group_size=1 << 20, compression="none", use_dictionary=False)
    del cols
    model, schema = bench.build_model(dev)
    model.compile(optimizer="adagrad", learning_rate=0.01)
    for chunk_rows in [int(x) for x in os.environ.get("CHUNKS", "8388608,2097152").split(",")]:
        ld = mm.Loader(path, schema, batch_size=B, shuffle=os.environ.get("SHUF", "1") == "1", seed=1, device=dev, drop_last=True, device_resident_bytes=0,
                       device_chunk_rows=chunk_rows)
        if os.environ.get("COPY_PRIO"):
            ld._copy_stream = torch.cuda.Stream(device=dev, priority=int(os.environ["COPY_PRIO"]))
        torch.cuda.synchronize(); t0 = time.perf_counter(); k = 0
        for _ in range(2):
            for x, y in ld:
                k += 1
        torch.cuda.synchronize()
        alone = k * B / (time.perf_counter() - t0)
        model.fit(ld, epochs=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ep = max(2, int(2.5 / (len(ld) * 1e-3)))
        model.fit(ld, epochs=ep)
        torch.cuda.synchronize()
        rate = ep * len(ld) * B / (time.perf_counter() - t0)
        print(f"rows {rows} chunk_rows {chunk_rows}: loader alone {alone / 1e6:.1f} M/s, streaming fit {rate / 1e6:.1f} M/s ({ep} epochs)", flush=True)
        del ld
    from models_amd.graph import PackedBatch, SegmentedStep
    batches = [PackedBatch(bench.make_batch(dev, B, 900 + i)) for i in range(4)]
    split = lambda t: ({k: v for k, v in t.items() if k != "__label__"}, t["__label__"])
    seg = SegmentedStep(lambda t: model.train_step(*split(t)), batches[0])
    for i in range(10): seg.replay(batches[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): seg.replay(batches[i % 4])
    torch.cuda.synchronize()
    print(f"resident step rate {200 * B / (time.perf_counter() - t0) / 1e6:.1f} M/s")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
