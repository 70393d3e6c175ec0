#!/usr/bin/env python
"""Why does the tiled forward scorer run 12 % slower in the product than the same kernel in tools/exp/scorer_lab?  Same call,
varied inputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from models_amd import ops

dev = torch.device("cuda", 0)
B, E = 32768, 128
g = torch.Generator(device="cpu").manual_seed(5)


def timeit(tag, fn, iters=6):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{tag:50s} {a.elapsed_time(b) / iters * 1e3:9.1f} us")


q = (torch.randn(B, E, generator=g) * 0.1).to(dev)
it = (torch.randn(B, E, generator=g) * 0.1).to(dev)
ids = torch.randperm(1_000_000, generator=g)[:B].to(torch.int32).to(dev)
ids2 = torch.randint(0, 1_000_000, (B,), generator=g).to(torch.int32).to(dev)
neg = it.clone()
uq = ((torch.rand(B, E, generator=g) - 0.5) * 0.18).to(dev)
un = ((torch.rand(B, E, generator=g) - 0.5) * 0.18).to(dev)
for mode in ("tiled", "stream"):
    os.environ["MERLIN_HIP_SCORER_FWD"] = mode
    timeit(f"{mode}: bench inputs (neg is item, T=1)", lambda: ops.inbatch_softmax(q, it, it, ids, ids, materialize=False))
    timeit(f"{mode}: T=0.05", lambda: ops.inbatch_softmax(q, it, it, ids, ids, 0.05, materialize=False))
    timeit(f"{mode}: separate neg buffer", lambda: ops.inbatch_softmax(q, it, neg, ids, ids, materialize=False))
    timeit(f"{mode}: separate neg buffer + other neg ids", lambda: ops.inbatch_softmax(q, it, neg, ids, ids2, materialize=False))
    timeit(f"{mode}: no ids", lambda: ops.inbatch_softmax(q, it, neg, None, None, materialize=False))
    timeit(f"{mode}: uniform data, T=0.05, separate", lambda: ops.inbatch_softmax(uq, un, neg, ids, ids2, 0.05, materialize=False))
