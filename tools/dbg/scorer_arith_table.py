"""Error table of the in-batch scorer's two arithmetics against float64 (host): exact fp32 MFMA kernels vs the opt-in bf16x3 split
(mh_scorer_split.hip).  L2-normalised rows and un-normalised rows, 1 / T = 20, duplicate ids (false negatives rescored).
Prints max |error| of loss, lse and of the gradients of the MEAN loss (scaled back by B)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from models_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
E, T, fns = 128, 0.05, -655.04


def ref64(q, it, neg, pid, nid):
    q64, i64, n64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (q, it, neg))
    pos = (q64 * i64).sum(1, keepdim=True)
    s = q64 @ n64.T
    s = torch.where(torch.tensor(pid.reshape(-1, 1) == nid.reshape(1, -1)), torch.full_like(s, fns), s)
    logits = torch.cat([pos, s], dim=1) / T
    lse = torch.logsumexp(logits, dim=1)
    loss = lse - logits[:, 0]
    loss.mean().backward()
    return [loss.detach().numpy(), lse.detach().numpy(), q64.grad.numpy(), i64.grad.numpy() + 0 * n64.grad.numpy()[: len(q)] if False else i64.grad.numpy(), n64.grad.numpy()]


for B, kind in ((2048, "unit"), (4096, "unit"), (2048, "randn x 0.1")):
    rng = np.random.default_rng(B)
    mk = lambda: rng.normal(size=(B, E))
    if kind == "unit":
        norm = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    else:
        norm = lambda a: (a * 0.1).astype(np.float32)
    q, it = norm(mk()), norm(mk())
    pid = rng.integers(0, B // 2, size=B).astype(np.int32)
    want = ref64(q, it, it, pid, pid)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    print(f"B = Nn = {B}, E = {E}, rows: {kind}, 1/T = {1 / T:.0f}  (|logit| <= {float(np.abs(q @ it.T).max()) / T:.1f})")
    for mode in ("f32", "bf16x3"):
        os.environ["MERLIN_HIP_SCORER_ARITH"] = mode
        res, dq, ditem = ops.inbatch_softmax_train(t(q), t(it), t(it), t(pid), t(pid), T, fns)
        _, _, dneg = ops.inbatch_softmax_backward(t(q), t(it), t(it), res.lse, t(pid), t(pid), T, fns, need_dq=False)
        got = [x.cpu().numpy() for x in (res.loss, res.lse, dq, ditem, dneg)]
        err = [float(np.abs(g.astype(np.float64) - w).max()) for g, w in zip(got, want)]
        print(f"  {mode:7s}: loss {err[0]:.3e}  lse {err[1]:.3e}  dq*B {err[2] * B:.3e}  ditem*B {err[3] * B:.3e}  dneg*B {err[4] * B:.3e}"
              f"   (max |dq|*B = {float(np.abs(want[2]).max()) * B:.3f})")
os.environ["MERLIN_HIP_SCORER_ARITH"] = "f32"
