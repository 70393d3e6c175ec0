#!/usr/bin/env python
"""Workload segments for the HBM-traffic PMC passes (tools/pmc_traffic.py runs each under rocprofv3 --pmc).
  calib   : a 1 GiB fill (write-only, 16 B / lane) and a 1 GiB sum (read-only, 16 B / lane): known byte counts
  embbwd  : mh_embedding_gather_bwd, Adagrad, BASELINE configs[1] shapes, uniform ids, 5 launches
  gather  : mh_embedding_gather_fwd at the same shapes, 5 launches
  cold    : gather of 512 K uniform rows of one 12.8 GB table (nothing cache-resident), 5 launches
  fused   : dlrm_fused_fwd / dlrm_fused_bwd (gather -> interaction in one kernel) at the same shapes, 5 launches each"""
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
elif seg == "fused":
    from models_amd.synthetic import CRITEO_CARDINALITIES

    g = torch.Generator(device="cpu").manual_seed(0)
    names = sorted([f"C{i}" for i in range(1, 27)] + ["bottom_block"])  # the stack order of the DLRM (sorted keys)
    card = {f"C{i}": v for i, v in enumerate(CRITEO_CARDINALITIES, 1)}
    tabs = [None if n == "bottom_block" else torch.rand(card[n], D, device=dev) for n in names]
    ids = [None if n == "bottom_block" else torch.randint(0, card[n], (B,), dtype=torch.int32, generator=g).to(dev) for n in names]
    dense = torch.rand(B, D, device=dev)
    P = F * (F - 1) // 2
    buf = torch.zeros(B, 416, device=dev)
    out = buf[:, :P + D]
    dout = torch.randn(B, 416, device=dev)[:, :P + D]
    for _ in range(N):
        ops.dlrm_interaction_fused(tabs, ids, dense, append_dense=True, out=out)
    for _ in range(N):
        ops.dlrm_interaction_fused_backward(tabs, ids, dense, dout, tail_to_dense=True)
elif seg == "cold":
    big = torch.rand(50_000_000, D, device=dev)
    idb = [torch.randint(0, 50_000_000, (B * 8,), dtype=torch.int32, device=dev)]
    out = torch.empty(B * 8, 1, D, device=dev)
    for _ in range(N):
        ops.embedding_gather([big], idb, out=out)
torch.cuda.synchronize()
