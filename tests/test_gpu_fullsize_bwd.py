"""Backward passes at the BASELINE.json sizes that the small-shape tests never reach (round-2 review, item 1):

  C3  flash-style scorer (forward + dq pass, ditem / dneg column pass) at B = 32 768, E = 128: 256 column tiles, duplicate
      ids, the lazy-rescale branch -- against an fp64 statement of tf/outputs/contrastive.py:276-344 + tf/losses/listwise.py:38-52
      evaluated in row blocks on the device (plain torch fp64 matmuls: an independent implementation), the numpy oracle on a
      row subset, and the checksum identities of the softmax gradient;
  C5  second-generation GEMM (NT dX, split-M TN dW) at K = N = 3 344 and the cross layer backward with the saved
      pre-activation at d = 3 341 (tf/blocks/cross.py:188-202) -- dX bit for bit against the C oracle's fmaf chain on a row
      subset, dW / db against fp64;
  C4  one > 4 GB table (20 M rows x 64): gather forward bit-exact, fused Adagrad backward against the numpy statement on
      the touched rows, every other row untouched (tf/inputs/embedding.py:424-471, IndexedSlices apply of models/base.py:1121-1174).
"""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import blocks, ops
from oracle import cbind, oracle as O

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------------------------
# C3
# ------------------------------------------------------------------------------------------------------------------
def _scorer_fp64_blocked(q, it, ids, T, fns, block=2048):
    """loss[B], lse[B], dq, ditem (positive role), dneg (negative role) of mean_b(lse_b - logit_b0) in fp64, row blocks."""
    B, E = q.shape
    qd, itd = q.double(), it.double()
    loss = torch.empty(B, dtype=torch.float64, device=q.device)
    lse = torch.empty_like(loss)
    dq = torch.empty_like(qd)
    ditem = torch.empty_like(qd)
    dneg = torch.zeros_like(itd)
    for s in range(0, B, block):
        e = min(B, s + block)
        qb, ib = qd[s:e], itd[s:e]
        pos = (qb * ib).sum(-1, keepdim=True)
        neg = qb @ itd.T
        mask = ids[s:e, None] == ids[None, :]
        neg = torch.where(mask, torch.full_like(neg, fns), neg)
        z = torch.cat([pos, neg], 1) / T
        l = torch.logsumexp(z, 1)
        lse[s:e] = l
        loss[s:e] = l - z[:, 0]
        P = torch.exp(z - l[:, None])
        P[:, 0] -= 1.0
        dz = P / (B * T)                                   # d mean-loss / d score (before the temperature)
        dn = torch.where(mask, torch.zeros_like(neg), dz[:, 1:])  # rescored entries are constants
        dq[s:e] = dz[:, :1] * ib + dn @ itd
        ditem[s:e] = dz[:, :1] * qb
        dneg += dn.T @ qb
    return loss, lse, dq, ditem, dneg


def _c3_inputs(case, device):
    g = torch.Generator().manual_seed(303)
    B, E = 32768, 128
    if case == "unique":
        # the headline workload: item ids without replacement (only the diagonal is masked), tower-like magnitudes
        q = torch.randn(B, E, generator=g) * 0.1
        it = torch.randn(B, E, generator=g) * 0.1
        ids = torch.randperm(1_000_000, generator=g)[:B].to(torch.int32)
        T = 1.0
    else:
        # duplicate ids (reference-skew lognormal(3, 1) recipe, datasets/synthetic.py:218-222): thousands of masked
        # off-diagonal entries; low temperature, positives far BELOW the negatives, negatives growing along the stream
        # and late spikes: the fused pass has to rescale its accumulators (lazy-rescale branch) in the last tiles
        q = torch.randn(B, E, generator=g) * 0.3
        it = torch.randn(B, E, generator=g) * 0.3
        it *= torch.linspace(0.3, 2.5, B)[:, None]
        it[B - 5] = q[17] * 4.0
        it[B // 2 + 3] = q[20000] * 5.0
        ids = torch.clamp(torch.exp(torch.randn(B, generator=g) + 3.0).long(), 1, 999_999).to(torch.int32)
        T = 0.1
    return q.to(device), it.to(device), ids.to(device), T


@pytest.mark.parametrize("case", ["unique", "duplicates_rescale"])
def test_c3_scorer_backward_full_batch(device, case):
    q, it, ids, T = _c3_inputs(case, device)
    B, E = q.shape
    fns = float(np.float32(O.MIN_FLOAT))
    want_loss, want_lse, want_dq, want_ditem, want_dneg = _scorer_fp64_blocked(q, it, ids, T, fns)

    (r, dq, ditem) = ops.inbatch_softmax_train(q, it, it, ids, ids, T)
    _, _, dneg = ops.inbatch_softmax_backward(q, it, it, r.lse, ids, ids, T, need_dq=False)
    # standalone passes (the path E > 128 or a non-fused caller takes)
    r0 = ops.inbatch_softmax(q, it, it, ids, ids, T, materialize=False)
    dq0, ditem0, dneg0 = ops.inbatch_softmax_backward(q, it, it, r0.lse, ids, ids, T)

    # the numpy oracle on a row subset (all 32 768 columns): logits -> loss / lse
    n = 256
    rows = torch.linspace(0, B - 1, n).long().to(device)
    lg, _ = O.contrastive_outputs(q[rows].cpu().numpy(), it[rows].cpu().numpy(), it.cpu().numpy(), ids[rows].cpu().numpy(),
                                  ids.cpu().numpy(), temperature=T)
    o_loss, o_lse = O.softmax_ce_first_column(lg)
    np.testing.assert_allclose(r.loss[rows].cpu().numpy(), o_loss, atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(r.lse[rows].cpu().numpy(), o_lse, atol=1e-4, rtol=1e-5)

    # Under the opt-in bf16x3 arithmetic (the suite is also run with MERLIN_HIP_SCORER_ARITH=bf16x3) a dot product carries
    # ~2.3e-6 |q| |item| of error: at the tower-like magnitudes of "unique" and of L2-normalised rows that is inside 1e-4 on a logit;
    # this case scales items up to 2.5 x, plants two 4-5 x spikes and divides by T = 0.1 (|q| |item| / T in the hundreds): one row in
    # 32 768 lands at 1.7e-4 absolute = 3.8e-5 relative.  The fp32 arithmetic keeps the 1e-4 bound here too.
    loose = case == "duplicates_rescale" and ops.scorer_arith() == "bf16x3"
    for got in (r, r0):
        torch.testing.assert_close(got.loss.double(), want_loss, atol=3e-4 if loose else 1e-4, rtol=5e-5 if loose else 1e-5)
        torch.testing.assert_close(got.lse.double(), want_lse, atol=3e-4 if loose else 1e-4, rtol=5e-5 if loose else 1e-6)
    # gradients of the MEAN loss are O(1 / (B T)): the absolute floor is relative to the largest entry (the same
    # 1e-4-of-scale the small-shape tests use; a wrong or missing tile is off by O(scale))
    scale = float(want_dq.abs().max())
    # (bf16x3, adversarial case: the standalone passes exponentiate APPROXIMATE logits against the lse of the EXACT forward-only
    # kernel -- a 1.7e-4 logit error is a 1.7e-4 relative error of a probability that the fused pass, whose lse comes from the same
    # approximate logits, normalises away)
    tol = dict(atol=(1e-3 if loose else 1e-4) * scale, rtol=3e-4)
    for name, got, want in (("dq", dq, want_dq), ("ditem", ditem, want_ditem), ("dneg", dneg, want_dneg),
                            ("dq0", dq0, want_dq), ("ditem0", ditem0, want_ditem), ("dneg0", dneg0, want_dneg)):
        torch.testing.assert_close(got.double(), want, msg=lambda m, n=name: f"{n}: {m}", **tol)
    # checksum identities (size-independent): the column sums of every gradient against fp64 (random-sign rounding
    # errors average out: far tighter than the element-wise bound times B) ...
    for got, want in ((dq, want_dq), (ditem, want_ditem), (dneg, want_dneg)):
        torch.testing.assert_close(got.double().sum(0), want.sum(0), atol=2e-2 * scale, rtol=1e-4)
    # ... and the scalar identity <dq, q> == <ditem, item> + <dneg, item> (scores are bilinear in q and the items)
    lhs = (dq.double() * q.double()).sum()
    rhs = (ditem.double() * it.double()).sum() + (dneg.double() * it.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-4 * max(1.0, abs(float(lhs)))
    if case == "duplicates_rescale":
        m = ids[:, None] == ids[None, :4096]
        assert int(m.sum()) > 4096 * 4  # the duplicate-id mask really has off-diagonal entries


# ------------------------------------------------------------------------------------------------------------------
# C5
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,act,x_act", [(3344, 3344, None, None), (3344, 512, "relu", None), (512, 256, "relu", "relu")])
def test_c5_linear_backward_wide(device, K, N, act, x_act):
    """dX = dz W^T (NT) and dW = x^T dz (split-M TN), db = colsum(dz) at the DCN-v2 widths, M = 4 096."""
    M = 4096
    if ops._linear_split_ok(M, K, N):
        pytest.skip("pins the fp32 GEMMs bit for bit; the bf16x3 Dense layer has its own test (test_gpu_gemm_split.py)")
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(M, K, generator=g)
    if x_act == "relu":
        x = torch.relu(x)
    W = (torch.rand(K, N, generator=g) - 0.5) * (2.0 / K ** 0.5)
    y = torch.randn(M, N, generator=g)
    if act == "relu":
        y = torch.relu(y)  # the layer's stored output: zero where the unit was off
    dy = torch.randn(M, N, generator=g) / M
    xd, Wd, yd, dyd = (t.to(device) for t in (x, W, y, dy))
    dz_in = dyd.clone()
    dx, dW, db = ops.linear_backward(xd, Wd, yd, dz_in, act, need_dx=True, need_db=True, x_activation=x_act)
    dz = dy * (y > 0) if act == "relu" else dy
    assert torch.equal(dz_in.cpu(), dz)  # dy overwritten with dz
    # dX: one k-ascending fmaf chain per output -> bit for bit against the C oracle on a row subset
    rows = np.linspace(0, M - 1, 48).astype(np.int64)
    ref = cbind.gemm_nt_fmaf(dz.numpy()[rows], W.numpy())  # [48, K] = dz @ W^T
    if x_act == "relu":
        ref = ref * (x.numpy()[rows] > 0)
    np.testing.assert_array_equal(dx.cpu().numpy()[rows], ref)
    # all of dX, dW, db against fp64 on the device
    dz64, x64, W64 = dz.to(device).double(), x.to(device).double(), Wd.double()
    want_dx = dz64 @ W64.T
    if x_act == "relu":
        want_dx = want_dx * (x64 > 0)
    torch.testing.assert_close(dx.double(), want_dx, atol=2e-7, rtol=1e-4)
    torch.testing.assert_close(dW.double(), x64.T @ dz64, atol=2e-6, rtol=1e-4)
    torch.testing.assert_close(db.double(), dz64.sum(0), atol=1e-6, rtol=1e-4)


def test_c5_cross_block_backward_with_saved_preactivation(device):
    """CrossBlock(depth = 2) at d = 3 341 (26 x 128 + 13, zero-padded to 3 344), M = 4 096, under blocks.tape(): forward against
    the numpy oracle on a row subset, dx / dW / db against fp64 autograd of x0 * (x W + b) + x on the device."""
    M, d = 4096, 3341
    g = torch.Generator().manual_seed(5)
    x_h = torch.randn(M, d, generator=g) * 0.5
    blk = mm.CrossBlock(depth=2, device=device)
    x = x_h.to(device)
    with blocks.tape():
        out = blk(x)
    for l in blk.layers:  # non-zero biases so that db is exercised
        l.bias.data[:d] = (torch.rand(d, generator=g) * 0.1).to(device)
    with blocks.tape():
        out = blk(x)
    assert all(l._p is not None for l in blk.layers)  # the forward GEMM stored p = x W + b
    layers = [(l.kernel.data[:d, :d], l.bias.data[:d]) for l in blk.layers]
    rows = np.linspace(0, M - 1, 32).astype(np.int64)
    ref = O.cross_block(x_h.numpy()[rows], [(W.cpu().numpy(), b.cpu().numpy()) for W, b in layers])
    np.testing.assert_allclose(out.cpu().numpy()[rows], ref, atol=1e-4 * d / 512, rtol=1e-4)
    dout = (torch.randn(M, d, generator=g) / M).to(device)
    dx = blk.backward(dout.clone())
    # fp64 autograd of the same block
    x64 = x.double().requires_grad_()
    ps = [(W.double().requires_grad_(), b.double().requires_grad_()) for W, b in layers]
    h = x64
    for W, b in ps:
        h = x64 * (h @ W + b) + h
    torch.testing.assert_close(out.double(), h.detach(), atol=1e-4 * d / 512, rtol=1e-4)
    h.backward(dout.double())
    torch.testing.assert_close(dx.double(), x64.grad, atol=1e-6, rtol=2e-4)
    for l, (W, b) in zip(blk.layers, ps):
        torch.testing.assert_close(l.kernel.grad[:d, :d].double(), W.grad, atol=2e-6, rtol=2e-4)
        torch.testing.assert_close(l.bias.grad[:d].double(), b.grad, atol=2e-6, rtol=2e-4)
        assert not bool(l.kernel.grad[d:].any()) and not bool(l.kernel.grad[:, d:].any())  # pad stays exactly zero


# ------------------------------------------------------------------------------------------------------------------
# C4
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_c4_table_above_4gb_gather_and_adagrad(device, idt):
    """One 20 M-row x 64 table (5.12 GB: byte offsets beyond 2^32, 25 key bits -> a three-pass sort) beside a 1 M-row and a
    tiny hot table; ids spread over the whole range with the last row, row 0 and a block of duplicates included."""
    V, D, B, lr, eps = 20_000_000, 64, 65536, 0.05, 1e-7
    rows = [V, 1_000_000, 7]
    g = torch.Generator(device=device).manual_seed(44)
    tabs0 = [torch.rand(v, D, generator=g, device=device) for v in rows]
    acc0 = [torch.full((v, D), 0.1, device=device) for v in rows]
    rng = np.random.default_rng(44)
    ids_h = [rng.integers(0, v, size=B) for v in rows]
    ids_h[0][:4] = [V - 1, 0, V - 1, 2 ** 24 + 1]            # both ends, a duplicate, a row just beyond 2^32 bytes
    ids_h[0][1000:1600] = rng.integers(V - 300, V, size=600)  # a cluster of duplicates at the far end (row offset > 4 GB)
    ids = [torch.from_numpy(i).to(idt).to(device) for i in ids_h]
    grad_h = rng.normal(size=(B, len(rows), D)).astype(np.float32) * 0.01
    grad = torch.from_numpy(grad_h).to(device)

    out = ops.embedding_gather(tabs0, ids)
    for f in range(len(rows)):
        assert torch.equal(out[:, f], tabs0[f][ids[f].long()]), f  # bit-exact row copies

    tabs = [t.clone() for t in tabs0]
    acc = [a.clone() for a in acc0]
    ops.embedding_gather_backward(tabs, acc, ids, grad, [f * D for f in range(len(rows))], "adagrad", lr, eps)
    for f, v in enumerate(rows):
        uniq, inv = np.unique(ids_h[f], return_inverse=True)
        gsum = np.zeros((len(uniq), D), np.float32)
        np.add.at(gsum, inv, grad_h[:, f])
        ut = torch.from_numpy(uniq).to(device)
        a_want = acc0[f][ut].cpu().numpy() + gsum * gsum
        t_want = tabs0[f][ut].cpu().numpy() - np.float32(lr) * gsum / (np.sqrt(a_want) + np.float32(eps))
        np.testing.assert_allclose(tabs[f][ut].cpu().numpy(), t_want, rtol=2e-5, atol=2e-6, err_msg=f"table {f}")
        np.testing.assert_allclose(acc[f][ut].cpu().numpy(), a_want, rtol=1e-4, atol=1e-7, err_msg=f"accumulator {f}")
        touched = torch.zeros(v, dtype=torch.bool, device=device)
        touched[ut] = True
        assert torch.equal((tabs[f] != tabs0[f]).any(dim=1) | (acc[f] != acc0[f]).any(dim=1), touched), f
