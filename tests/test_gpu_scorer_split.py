"""The split-bf16 arithmetics of the in-batch scorer (mh_scorer_split.hip): the fp32-grade six-term "bf16x6" (mh_set_scorer_arith(2), the
default) and the opt-in three-term "bf16x3" (mh_set_scorer_arith(1)): loss, lse, dq, ditem, dneg against the exact-fp32 kernels on the
same inputs and against a float64 statement of ContrastiveOutput.outputs + CategoricalCrossEntropy (tf/outputs/contrastive.py:276-344,
tf/losses/listwise.py:38-52).  bf16x3: north_star's tolerance (logits / lse within 1e-4).  bf16x6: no further from float64 than a small
multiple of the exact fp32 kernels' own distance."""
import numpy as np
import pytest
import torch

from models_amd import ops

pytestmark = pytest.mark.gpu


def _ref64(q, it, neg, pid, nid, T, fns):
    """float64: logits = [pos, q neg^T with false negatives rescored] / T; loss = lse - pos; gradients of mean loss."""
    q64, i64, n64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (q, it, neg))
    pos = (q64 * i64).sum(1, keepdim=True)
    s = q64 @ n64.T
    if pid is not None:
        mask = torch.tensor(pid.reshape(-1, 1) == nid.reshape(1, -1))
        s = torch.where(mask, torch.full_like(s, fns), s)
    logits = torch.cat([pos, s], dim=1) / T
    lse = torch.logsumexp(logits, dim=1)
    loss = lse - logits[:, 0]
    loss.mean().backward()
    return loss.detach().numpy(), lse.detach().numpy(), q64.grad.numpy(), i64.grad.numpy(), n64.grad.numpy()


@pytest.mark.parametrize("arith", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("B,Nn,ids,idt,E", [(512, 512, True, np.int32, 128), (300, 300, True, np.int64, 128), (256, 1000, False, None, 128),
                                            (700, 97, True, np.int64, 128), (333, 65, True, np.int32, 128), (4096, 4096, True, np.int32, 128),
                                            (512, 512, True, np.int32, 64), (300, 1000, True, np.int64, 64), (700, 97, False, None, 64),
                                            (4096, 4096, True, np.int32, 64)])
def test_split_scorer_matches_fp32_kernels_and_float64(device, monkeypatch, B, Nn, ids, idt, E, arith):
    if E == 64 and arith == "bf16x3":
        pytest.skip("E = 64 exists in the six-term kernel only (bf16x3 calls run the exact chains there)")
    rng = np.random.default_rng(B + Nn + E)
    T, fns = 0.05, -655.04
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    q, it = unit(rng.normal(size=(B, E))), unit(rng.normal(size=(B, E)))
    neg = it if Nn == B else unit(rng.normal(size=(Nn, E)))
    pid = nid = None
    if ids:
        pid = rng.integers(0, max(B // 2, 8), size=B).astype(idt)  # duplicates: false negatives beside the diagonal
        nid = pid if Nn == B else rng.integers(0, max(B // 2, 8), size=Nn).astype(idt)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)
    args = (t(q), t(it), t(neg), t(pid), t(nid), T, fns)

    def run():
        res, dq, ditem = ops.inbatch_softmax_train(*args)
        _, _, dneg = ops.inbatch_softmax_backward(args[0], args[1], args[2], res.lse, args[3], args[4], T, fns, need_dq=False)
        dq2, ditem2, dneg2 = ops.inbatch_softmax_backward(args[0], args[1], args[2], res.lse, args[3], args[4], T, fns)
        fw = ops.inbatch_softmax(*args, materialize=False)  # forward-only pass (evaluation): loss / lse
        return [x.cpu().numpy() for x in (res.loss, res.lse, dq, ditem, dneg, dq2, ditem2, dneg2, fw.loss, fw.lse)]

    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    f32 = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", arith)
    sp = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    names = ("loss", "lse", "dq", "ditem", "dneg", "dq(bwd)", "ditem(bwd)", "dneg(bwd)", "loss(fwd)", "lse(fwd)")
    assert any(not np.array_equal(a, b) for a, b in zip(f32, sp)), "the switch did not change the arithmetic"
    x6 = arith == "bf16x6"
    for n, a, b in zip(names, f32, sp):
        # gradients of the MEAN loss carry 1 / B; bf16x6 against the fp32 kernels: a few ulps of the lse (~ 10: ulp 1e-6)
        tol = (1e-5 if x6 else 1e-4) if n.startswith(("loss", "lse")) else (2e-7 if x6 else 2e-6)
        np.testing.assert_allclose(b, a, atol=tol, rtol=2e-5 if x6 else 2e-4, err_msg=n)
    if B <= 1024:
        loss, lse, dq, ditem, dneg = _ref64(q, it, neg, pid, nid, T, fns)
        if x6:  # fp32-grade: never further from float64 than 4 x the exact fp32 kernels (+ an ulp of the quantity's scale)
            for k, want in ((0, loss), (1, lse), (2, dq), (3, ditem), (4, dneg), (8, loss), (9, lse)):
                e_sp, e_f32 = np.abs(sp[k] - want).max(), np.abs(f32[k] - want).max()
                floor = 2.0 ** -23 * max(np.abs(want).max(), 1e-30)
                assert e_sp <= 4 * e_f32 + floor, (names[k], e_sp, e_f32)
        np.testing.assert_allclose(sp[0], loss, atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(sp[1], lse, atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(sp[2], dq, atol=2e-6, rtol=2e-4)
        np.testing.assert_allclose(sp[3], ditem, atol=2e-6, rtol=2e-4)
        np.testing.assert_allclose(sp[4], dneg, atol=2e-6, rtol=2e-4)
        np.testing.assert_allclose(sp[8], loss, atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(sp[9], lse, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("after_mask", [False, True])
def test_six_term_scorer_applies_the_logq_corrections(device, monkeypatch, after_mask):
    """logQ sampling correction (outputs/contrastive.py:309-319) inside the six-term kernel, before or after the false-negative
    rescoring: against float64 never further than 4 x the exact-fp32 kernels (the corrected logits reach +-150: an ulp of a logit is
    1.5e-5 there, for either kernel) -- and not the same bits as those kernels (the split kernel really ran)."""
    rng = np.random.default_rng(7)
    B, Nn, E, T, fns = 640, 1000, 128, 0.05, -655.04
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    q, it, neg = unit(rng.normal(size=(B, E))), unit(rng.normal(size=(B, E))), unit(rng.normal(size=(Nn, E)))
    pid, nid = rng.integers(0, 300, size=B).astype(np.int32), rng.integers(0, 300, size=Nn).astype(np.int32)
    plq = np.log(rng.random(B) * 0.3 + 1e-3).astype(np.float32)
    nlq = np.log(rng.random(Nn) * 0.3 + 1e-3).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    args = (t(q), t(it), t(neg), t(pid), t(nid), T, fns)
    kw = dict(pos_logq=t(plq), neg_logq=t(nlq), logq_after_mask=after_mask)

    # float64 statement of the corrected logits and the gradients of the mean loss
    q64, i64, n64 = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (q, it, neg))
    lp, ln = torch.tensor(plq, dtype=torch.float64), torch.tensor(nlq, dtype=torch.float64)
    sc = q64 @ n64.T
    mask = torch.tensor(pid.reshape(-1, 1) == nid.reshape(1, -1))
    sc = (torch.where(mask, torch.full_like(sc, fns), sc) - ln[None, :]) if after_mask else torch.where(mask, torch.full_like(sc, fns), sc - ln[None, :])
    z = torch.cat([(q64 * i64).sum(1, keepdim=True) - lp[:, None], sc], 1) / T
    lse64 = torch.logsumexp(z, 1)
    (lse64 - z[:, 0]).mean().backward()
    want = [(lse64 - z[:, 0]).detach().numpy(), lse64.detach().numpy(), q64.grad.numpy(), i64.grad.numpy(), n64.grad.numpy(), lse64.detach().numpy()]

    def run():
        res, dq, ditem = ops.inbatch_softmax_train(*args, **kw)
        _, _, dneg = ops.inbatch_softmax_backward(args[0], args[1], args[2], res.lse, args[3], args[4], T, fns, need_dq=False, **kw)
        fw = ops.inbatch_softmax(*args, materialize=False, **kw)
        return [x.cpu().numpy() for x in (res.loss, res.lse, dq, ditem, dneg, fw.lse)]

    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    f32 = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "bf16x6")
    sp = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    assert any(not np.array_equal(a, b) for a, b in zip(f32, sp)), "the six-term kernel did not take the logQ-corrected passes"
    for n, a, b, w in zip(("loss", "lse", "dq", "ditem", "dneg", "lse(fwd)"), f32, sp, want):
        e_sp, e_f32 = np.abs(b - w).max(), np.abs(a - w).max()
        floor = 2.0 ** -23 * max(np.abs(w).max(), 1e-30)
        assert e_sp <= 4 * e_f32 + floor, (n, e_sp, e_f32)
        np.testing.assert_allclose(b, w, atol=1e-4 if n.startswith(("loss", "lse")) else 1e-6, rtol=1e-3, err_msg=n)


@pytest.mark.parametrize("E", [128, 64])
@pytest.mark.parametrize("B,Nn,ids,logq", [(10, 1000, True, False), (65, 64, True, True), (1, 64, False, False), (31, 97, True, True),
                                           (257, 2049, True, False), (1000, 65, True, True), (33, 4097, False, True)])
def test_six_term_scorer_odd_shapes(device, monkeypatch, E, B, Nn, ids, logq):
    """Fewer stationary rows than a wavefront block, one streamed tile, ragged everything, int64 ids, corrections in either order: every
    output of the six-term kernel within 2e-4 of its scale of the exact-chain kernels' (and finite)."""
    rng = np.random.default_rng(B * 7 + Nn + E)
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    q, it, neg = unit(rng.normal(size=(B, E))), unit(rng.normal(size=(B, E))), unit(rng.normal(size=(Nn, E)))
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)
    pid = rng.integers(0, 50, size=B).astype(np.int64) if ids else None
    nid = rng.integers(0, 50, size=Nn).astype(np.int64) if ids else None
    kw = {}
    if logq:
        kw = dict(pos_logq=t(np.log(rng.random(B) * 0.3 + 1e-3).astype(np.float32)),
                  neg_logq=t(np.log(rng.random(Nn) * 0.3 + 1e-3).astype(np.float32)), logq_after_mask=bool(B % 2))
    args = (t(q), t(it), t(neg), t(pid), t(nid), 0.05, -655.04)

    def run():
        res, dq, ditem = ops.inbatch_softmax_train(*args, **kw)
        _, _, dneg = ops.inbatch_softmax_backward(args[0], args[1], args[2], res.lse, args[3], args[4], 0.05, -655.04, need_dq=False, **kw)
        fw = ops.inbatch_softmax(*args, materialize=False, **kw)
        return [x.double().cpu() for x in (res.loss, res.lse, dq, ditem, dneg, fw.lse)]

    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    f32 = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "bf16x6")
    sp = run()
    monkeypatch.setenv("MERLIN_HIP_SCORER_ARITH", "f32")
    for n, a, b in zip(("loss", "lse", "dq", "ditem", "dneg", "lse(fwd)"), f32, sp):
        assert torch.isfinite(b).all(), n
        assert float((a - b).abs().max()) <= 2e-4 * max(float(a.abs().max()), 1e-30), n
