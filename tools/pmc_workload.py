#!/usr/bin/env python
"""Workload segments for the HBM-traffic PMC passes (tools/pmc_traffic.py runs each under rocprofv3 --pmc).
  calib   : a 1 GiB fill (write-only, 16 B / lane) and a 1 GiB sum (read-only, 16 B / lane): known byte counts
  embbwd  : mh_embedding_gather_bwd, Adagrad, BASELINE configs[1] shapes, uniform ids, 5 launches
  gather  : mh_embedding_gather_fwd at the same shapes, 5 launches
  cold    : gather of 512 K uniform rows of one 12.8 GB table (nothing cache-resident), 5 launches"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from models_amd import ops

dev = torch.device("cuda")
B, F, D = 65536, 27, 64
seg = sys.argv[1]
N = 5
if seg == "calib":
    x = torch.empty(1 << 28, device=dev)  # 1 GiB of fp32
    for _ in range(N):
        x.fill_(1.0)
    torch.cuda.synchronize()
    for _ in range(N):
        x.sum()
    torch.cuda.synchronize()
elif seg in ("embbwd", "gather"):
    from models_amd.synthetic import CRITEO_CARDINALITIES

    g = torch.Generator(device="cpu").manual_seed(0)
    tabs = [torch.rand(v, D, device=dev) for v in CRITEO_CARDINALITIES]
    ids = [torch.randint(0, v, (B,), dtype=torch.int32, generator=g).to(dev) for v in CRITEO_CARDINALITIES]
    if seg == "embbwd":
        st = [torch.full_like(t, 0.1) for t in tabs]
        grad = torch.randn(B, F, D, device=dev)
        offs = [i * D for i in range(26)]
        for _ in range(N):
            ops.embedding_gather_backward(tabs, st, ids, grad, offs, "adagrad", 0.01, 1e-7)
    else:
        out = torch.empty(B, F, D, device=dev)
        for _ in range(N):
            ops.embedding_gather(tabs, ids, out=out)
elif seg == "cold":
    big = torch.rand(50_000_000, D, device=dev)
    idb = [torch.randint(0, 50_000_000, (B * 8,), dtype=torch.int32, device=dev)]
    out = torch.empty(B * 8, 1, D, device=dev)
    for _ in range(N):
        ops.embedding_gather([big], idb, out=out)
torch.cuda.synchronize()
