#!/usr/bin/env python
"""Where does the HOST time of an eagerly launched DLRM train step go?  (The eager step is within ~0.1 ms of being host-bound:
0.87 ms of Python dispatch for 0.96 ms of GPU work.)  Runs the real Python path on CPU tensors with the C library replaced by a stub
whose entry points return at once (results are garbage; only the host cost of issuing a step is of interest), and prints a cProfile
of 200 steps.  No GPU needed:  python tools/dbg/host_profile.py [top_n]"""
import cProfile
import ctypes
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import models_amd as mm
from models_amd import _lib, ops, schema as S


class _Stub:
    """Every mh_* symbol: a Python callable taking anything, returning 0 (workspace-size queries: 1 MiB)."""

    def __getattr__(self, name):
        if name.endswith("_workspace_bytes"):
            return lambda *a: 1 << 20
        if name == "mh_mlp_chain_supported":
            return lambda *a: 1
        if name == "mh_last_error":
            return lambda: b""
        return lambda *a: 0


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 35
    stub = _Stub()
    _lib.load = lambda: stub
    ops._dev = lambda t, name, dtype=None: t          # accept CPU tensors
    ops._stream = lambda: ctypes.c_void_p(0)
    ops._sync_deterministic = lambda *a, **k: None
    ops.SIDE.enabled = False                           # torch.cuda streams do not exist here; their cost is not Python's anyway
    from models_amd.synthetic import CRITEO_CARDINALITIES

    cards = [min(c, 2000) for c in CRITEO_CARDINALITIES]  # small tables: the host path does not depend on their size
    cols = [S.categorical(f"C{i + 1}", c) for i, c in enumerate(cards)] + [S.continuous(f"I{i + 1}") for i in range(13)]
    cols.append(S.binary_target("label"))
    schema = mm.Schema(cols)
    dev = torch.device("cpu")
    model = mm.DLRMModel(schema, embedding_dim=64, bottom_block=mm.MLPBlock([128, 64], device=dev), top_block=mm.MLPBlock([128, 64, 32], device=dev),
                         device=dev)
    model.compile(optimizer="adagrad", learning_rate=0.01)
    B = 4096
    g = torch.Generator().manual_seed(0)
    x = {f"C{i + 1}": torch.randint(0, c, (B,), generator=g, dtype=torch.int32) for i, c in enumerate(cards)}
    x.update({f"I{i + 1}": torch.rand(B, generator=g) for i in range(13)})
    y = torch.randint(0, 2, (B, 1), generator=g).float()
    for _ in range(5):
        model.train_step(x, y)
    import time

    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        model.train_step(x, y)
    dt = (time.perf_counter() - t0) / n
    print(f"host time per issued step with a stubbed library (CPU tensors, B = {B}): {dt * 1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        model.train_step(x, y)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(top)


if __name__ == "__main__":
    main()
