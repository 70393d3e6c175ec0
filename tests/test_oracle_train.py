"""The numpy DLRM train-step restatement vs PyTorch-CPU autograd (no GPU)."""
import numpy as np
import torch

from oracle import oracle as O
from tests import torch_ref as R


def test_oracle_dlrm_train_step_matches_torch_autograd():
    rng = np.random.default_rng(0)
    cards = {"C1": 30, "C10": 5, "C2": 200}
    D, B, lr = 8, 97, 0.05
    tables = {n: O.embedding_uniform(rng, v, D) for n, v in cards.items()}
    mk = lambda dims: [(O.glorot_uniform(rng, a, b), (rng.normal(size=b) * 0.01).astype(np.float32), "relu")
                       for a, b in zip(dims[:-1], dims[1:])]
    bottom, top = mk([3, 16, D]), mk([4 * 3 // 2 + D, 16, 8])
    head = (O.glorot_uniform(rng, 8, 1), np.zeros(1, np.float32))
    t = lambda a: torch.from_numpy(a.copy()).requires_grad_()
    tt = {n: t(v) for n, v in tables.items()}
    tb = [(t(W), t(b), a) for W, b, a in bottom]
    tp = [(t(W), t(b), a) for W, b, a in top]
    th = (t(head[0]), t(head[1]))
    params = list(tt.values()) + [x for l in tb + tp for x in l[:2]] + list(th)
    accs = [torch.full_like(p, 0.1) for p in params]
    states = None
    for step in range(3):
        cat = {n: rng.integers(0, v, size=(B, 1)) for n, v in cards.items()}
        cont = {f"I{i}": rng.random(size=(B, 1)).astype(np.float32) for i in range(3)}
        y = rng.integers(0, 2, size=(B, 1)).astype(np.float32)
        loss, states = O.dlrm_train_step(cat, cont, y, tables, bottom, top, head, states, "adagrad", lr)
        ref = R.keras_bce(R.dlrm_forward({k: torch.from_numpy(v) for k, v in cat.items()},
                                         {k: torch.from_numpy(v) for k, v in cont.items()}, tt, tb, tp, th), torch.from_numpy(y))
        assert abs(loss - ref.item()) < 1e-5
        grads = torch.autograd.grad(ref, params)
        with torch.no_grad():
            for k, (p, g) in enumerate(zip(params, grads)):
                touched = (g != 0).any(dim=-1, keepdim=True) if k < len(tt) else torch.ones_like(p, dtype=torch.bool)
                w2, a2 = R.adagrad_update(p, g, accs[k], lr)
                p.copy_(torch.where(touched, w2, p))
                accs[k] = torch.where(touched, a2, accs[k])
    for n in cards:
        np.testing.assert_allclose(tables[n], tt[n].detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(top[0][0], tp[0][0].detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(head[0], th[0].detach().numpy(), atol=1e-5)


# ---- list lookups: oracle gradient vs torch autograd (CPU) -------------------------------------------
import pytest as _pytest


@_pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
def test_embedding_bag_grad_matches_autograd(combiner):
    import torch
    from oracle import oracle as O

    rng = np.random.default_rng(3)
    V, D, B = 50, 8, 17
    lens = rng.integers(0, 6, size=B)
    lens[3] = 0
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
    grad = rng.standard_normal((B, D)).astype(np.float32)
    W = torch.randn(V, D, requires_grad=True)
    rows = []
    for b in range(B):
        seg = W[torch.from_numpy(values[offsets[b]:offsets[b + 1]])]
        n = seg.shape[0]
        r = seg.sum(0)
        if n and combiner == "mean":
            r = r / n
        if n and combiner == "sqrtn":
            r = r / np.sqrt(np.float32(n))
        rows.append(r)
    (torch.stack(rows) * torch.from_numpy(grad)).sum().backward()
    got = O.embedding_bag_grad(V, values, offsets, grad, combiner)
    np.testing.assert_allclose(got, W.grad.numpy(), rtol=1e-5, atol=1e-6)
    # forward of the same oracle agrees with the graph it differentiates
    np.testing.assert_allclose(O.embedding_bag(W.detach().numpy(), values, offsets, combiner),
                               torch.stack(rows).detach().numpy(), rtol=1e-5, atol=1e-6)


def test_embedding_bag_grad_prunes_and_dense_list():
    from oracle import oracle as O

    grad = np.ones((2, 4), np.float32)
    # bag 0 = [-1, 2, 99(out of range)], bag 1 = [2, 2]; mean divides by kept (non-negative) ids
    dW = O.embedding_bag_grad(5, np.array([-1, 2, 99, 2, 2]), np.array([0, 3, 5]), grad, "mean")
    np.testing.assert_allclose(dW[2], np.full(4, 0.5 + 0.5 + 0.5, np.float32))
    assert np.count_nonzero(dW) == 4
    dL = O.embedding_bag_grad(5, np.array([[0, 1, 1], [4, 4, 4]]), None, grad, "mean")
    np.testing.assert_allclose(dL[1], np.full(4, 2 / 3, np.float32), rtol=1e-6)
    np.testing.assert_allclose(dL[4], np.full(4, 1.0, np.float32), rtol=1e-6)
