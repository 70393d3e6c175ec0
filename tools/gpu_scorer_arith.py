"""Scorer passes at B x B x 128 under the three arithmetics (f32 chains, bf16x6, bf16x3): time per pass and distance to float64.
Usage (GPU box): python tools/gpu_scorer_arith.py [B=65536] [E=128|64] [modes=f32,bf16x6,bf16x3]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from models_amd import ops  # noqa: E402

B, E_ARG = 65536, 128
for a in sys.argv[1:]:
    if a.startswith("B="):
        B = int(a[2:])
    if a.startswith("E="):
        E_ARG = int(a[2:])
dev = torch.device("cuda:0")
E, T = E_ARG, 0.05
g = torch.Generator().manual_seed(5)
unit = lambda x: x / x.norm(dim=1, keepdim=True)
q, it = unit(torch.randn(B, E, generator=g)).to(dev), unit(torch.randn(B, E, generator=g)).to(dev)
ids = torch.randint(0, B // 2, (B,), generator=g, dtype=torch.int32).to(dev)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ref64(n):
    q64, i64 = q[:n].double().cpu().requires_grad_(True), it[:n].double().cpu().requires_grad_(True)
    idc = ids[:n].cpu()
    s = q64 @ i64.T
    pos = (q64 * i64).sum(1, keepdim=True)
    s = torch.where(idc.view(-1, 1) == idc.view(1, -1), torch.full_like(s, -655.04), s)
    lg = torch.cat([pos, s], 1) / T
    lse = torch.logsumexp(lg, 1)
    (lse - lg[:, 0]).mean().backward()
    return lse.detach(), q64.grad, i64.grad


n = 2048
lse64, dq64, di64 = ref64(n)
MODES = [a[6:].split(",") for a in sys.argv[1:] if a.startswith("modes=")]
for mode in (MODES[0] if MODES else ("f32", "bf16x6", "bf16x3")):
    os.environ["MERLIN_HIP_SCORER_ARITH"] = mode
    res = [None]

    def fwd_dq():
        res[0] = ops.inbatch_softmax_train(q, it, it, ids, ids, T)

    def col():
        ops.inbatch_softmax_backward(q, it, it, res[0][0].lse, ids, ids, T, need_dq=False)

    def fwd():
        ops.inbatch_softmax(q, it, it, ids, ids, T, materialize=False)

    t1, t2, t3 = timed(fwd_dq), timed(col), timed(fwd)
    r, dq, ditem = ops.inbatch_softmax_train(q[:n].contiguous(), it[:n].contiguous(), it[:n].contiguous(), ids[:n].contiguous(), ids[:n].contiguous(), T)
    _, _, dneg = ops.inbatch_softmax_backward(q[:n].contiguous(), it[:n].contiguous(), it[:n].contiguous(), r.lse, ids[:n].contiguous(), ids[:n].contiguous(), T, need_dq=False)
    e_lse = float((r.lse.double().cpu() - lse64).abs().max())
    e_dq = float((dq.double().cpu() - dq64).abs().max() * n)
    e_di = float(((ditem + dneg).double().cpu() - di64).abs().max() * n)
    fl = 2.0 * B * B * E
    print(f"{mode:7s} B={B} E={E}: fwd+dq {t1:7.3f} ms ({2 * fl / t1 / 1e9:7.1f} TF fp32-eq)  column pass {t2:7.3f} ms  fwd only {t3:7.3f} ms   "
          f"n={n}: |lse-f64| {e_lse:.2e}  |dq-f64|*n {e_dq:.2e}  |ditem-f64|*n {e_di:.2e}", flush=True)
