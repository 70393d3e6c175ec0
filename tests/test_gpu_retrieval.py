"""Scorer / top-k / cross HIP kernels vs the oracle and the reference's known answers."""
import os

import numpy as np
import pytest
import torch

from models_amd import ops
from oracle import cbind, oracle as O
from tests import torch_ref as R

pytestmark = pytest.mark.gpu
ATOL = 1e-4  # north-star fp32 logits tolerance


def _pins_f32_arithmetic():
    """Tests that pin the scorer to the oracle at ATOL in an fp32-grade arithmetic (the exact chains, or the default six-term bf16x6):
    under the opt-in MERLIN_HIP_SCORER_ARITH=bf16x3 the logits-free E=128 forward runs the 3-term bf16 split, whose own tolerance is
    tested in test_gpu_scorer_split.py."""
    if ops.scorer_arith() == "bf16x3":
        pytest.skip("pins the fp32-grade scorer; the bf16x3 mode has its own tests (test_gpu_scorer_split.py)")


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_scorer_reference_known_answers(device):
    # tests/unit/torch/outputs/test_constrastive.py:31-47 (no downscore)
    q = np.array([[0.1, 0.2, 0, 0], [0.3, 0.4, 0, 0]], np.float32)
    p = np.array([[0.5, 0.6, 0, 0], [0.7, 0.8, 0, 0]], np.float32)
    n = np.array([[0.9, 1.0, 0, 0], [1.1, 1.2, 0, 0], [1.3, 1.4, 0, 0]], np.float32)
    r = ops.inbatch_softmax(_t(q, device), _t(p, device), _t(n, device))
    exp = np.array([[0.17, 0.29, 0.35, 0.41], [0.53, 0.67, 0.81, 0.95]], np.float32)
    np.testing.assert_allclose(r.logits.cpu().numpy(), exp, atol=ATOL)
    # :49-73 (false negative -> -100)
    q1, p1 = q[:1], p[:1]
    n1 = np.array([[0.5, 0.6, 0, 0], [0.9, 1.0, 0, 0]], np.float32)
    r = ops.inbatch_softmax(_t(q1, device), _t(p1, device), _t(n1, device), _t(np.array([0]), device),
                            _t(np.array([0, 1]), device), false_neg_score=-100.0)
    np.testing.assert_allclose(r.logits.cpu().numpy(), [[0.17, -100.0, 0.29]], atol=ATOL)


@pytest.mark.parametrize("B,E", [(300, 32), (1000, 128), (129, 64), (64, 8)])
@pytest.mark.parametrize("temperature", [1.0, 0.25])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_inbatch_scorer_matches_oracle(device, B, E, temperature, idt):
    if E == 128:
        _pins_f32_arithmetic()
    rng = np.random.default_rng(B + E)
    q = rng.normal(size=(B, E)).astype(np.float32) * 0.3
    it = rng.normal(size=(B, E)).astype(np.float32) * 0.3
    ids = rng.integers(0, max(B // 3, 2), size=B).astype(idt)  # many duplicate ids -> off-diagonal false negatives
    logits, _ = O.contrastive_outputs(q, it, it, ids, ids, temperature=temperature)
    loss, lse = O.softmax_ce_first_column(logits)
    r = ops.inbatch_softmax(_t(q, device), _t(it, device), _t(it, device), _t(ids, device), _t(ids, device), temperature)
    got = r.logits.cpu().numpy()
    np.testing.assert_allclose(got, logits, atol=ATOL * max(1.0, 1.0 / temperature), rtol=1e-5)
    np.testing.assert_allclose(r.lse.cpu().numpy(), lse, atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(r.loss.cpu().numpy(), loss, atol=ATOL, rtol=1e-4)
    # diagonal always masked (tests/unit/tf/outputs/test_contrastive.py:173-206)
    assert np.all(np.diag(got[:, 1:]) == np.float32(np.float32(O.MIN_FLOAT) * np.float32(1.0 / temperature)))
    # fused mode: identical loss without materialising the logits
    r2 = ops.inbatch_softmax(_t(q, device), _t(it, device), _t(it, device), _t(ids, device), _t(ids, device), temperature,
                             materialize=False)
    assert r2.logits is None
    # one kernel family behind both modes: bit for bit -- unless the opt-in tiled forward kernel (another summation order) is on, or the
    # logits-free E = 128 / 64 forward runs the (fp32-grade) six-term split while the materialised logits come from the exact chains: a few
    # ulps of the lse's scale then
    tiled = os.environ.get("MERLIN_HIP_SCORER_FWD") == "tiled"
    if (E == 128 and ops.scorer_arith() != "f32") or (E == 64 and ops.scorer_arith() == "bf16x6"):
        torch.testing.assert_close(r2.loss, r.loss, atol=2e-5, rtol=4e-6)
    else:
        torch.testing.assert_close(r2.loss, r.loss, atol=2e-6 if tiled else 0, rtol=1e-6 if tiled else 0)


def test_scorer_scores_are_fmaf_chains(device):
    rng = np.random.default_rng(5)
    B, Nn, E = 257, 300, 96
    q, it, ng = (rng.normal(size=s).astype(np.float32) for s in ((B, E), (B, E), (Nn, E)))
    r = ops.inbatch_softmax(_t(q, device), _t(it, device), _t(ng, device))
    np.testing.assert_array_equal(r.logits[:, 1:].cpu().numpy(), cbind.gemm_nt_fmaf(q, ng))


def test_scorer_backward_matches_autograd(device):
    g = torch.Generator().manual_seed(3)
    B, E, T = 300, 64, 0.5
    q = (torch.randn(B, E, generator=g) * 0.3).requires_grad_()
    it = (torch.randn(B, E, generator=g) * 0.3).requires_grad_()
    ids = torch.randint(0, 100, (B,), generator=g)
    pos = (q * it).sum(-1, keepdim=True)
    neg = q @ it.T
    neg = torch.where(ids[:, None] == ids[None, :], torch.full_like(neg, O.MIN_FLOAT), neg)
    logits = torch.cat([pos, neg], 1) / T
    loss = (torch.logsumexp(logits, 1) - logits[:, 0]).mean()
    loss.backward()
    qd, itd, idd = q.detach().to(device), it.detach().to(device), ids.to(device)
    r = ops.inbatch_softmax(qd, itd, itd, idd, idd, T, materialize=False)
    assert abs(r.loss.mean().item() - loss.item()) < 1e-4
    dq, ditem, dneg = ops.inbatch_softmax_backward(qd, itd, itd, r.lse, idd, idd, T)
    torch.testing.assert_close(dq.cpu(), q.grad, atol=1e-5, rtol=1e-3)
    torch.testing.assert_close((ditem + dneg).cpu(), it.grad, atol=1e-5, rtol=1e-3)


def _autograd_scorer(q, it, ng, pid, nid, T, fns=O.MIN_FLOAT):
    """fp64 autograd statement of the sampled-softmax loss mean and its gradients (separate negatives)."""
    q, it, ng = (torch.from_numpy(a).double().requires_grad_() for a in (q, it, ng))
    pos = (q * it).sum(-1, keepdim=True)
    neg = q @ ng.T
    if pid is not None:
        m = torch.from_numpy(pid.astype(np.int64))[:, None] == torch.from_numpy(nid.astype(np.int64))[None, :]
        neg = torch.where(m, torch.full_like(neg, float(np.float32(fns))), neg)
    logits = torch.cat([pos, neg], 1) / T
    per_row = torch.logsumexp(logits, 1) - logits[:, 0]
    per_row.mean().backward()
    return per_row.detach().numpy(), q.grad.numpy(), it.grad.numpy(), ng.grad.numpy()


@pytest.mark.parametrize("B,Nn,E,idt", [(300, 300, 64, np.int64), (513, 700, 128, np.int32), (1000, 257, 32, np.int32),
                                        (130, 130, 96, np.int64), (70, 90, 8, None), (257, 64, 128, None)])
def test_scorer_flash_backward_matches_autograd(device, B, Nn, E, idt):
    """Streaming kernels (mh_scorer_stream.hip): standalone row + column passes, and the fused forward+dq pass followed
    by the column pass, against fp64 autograd -- ragged sizes, separate negatives, both id widths, padded E."""
    rng = np.random.default_rng(B * 7 + Nn + E)
    T = 0.5
    q, it, ng = (rng.normal(size=s).astype(np.float32) * 0.4 for s in ((B, E), (B, E), (Nn, E)))
    pid = nid = None
    if idt is not None:
        pid = rng.integers(0, 60, size=B).astype(idt)
        nid = rng.integers(0, 60, size=Nn).astype(idt)
    loss, gq, git, gng = _autograd_scorer(q, it, ng, pid, nid, T)
    qd, itd, ngd = _t(q, device), _t(it, device), _t(ng, device)
    pd = None if pid is None else _t(pid, device)
    nd = None if nid is None else _t(nid, device)
    r = ops.inbatch_softmax(qd, itd, ngd, pd, nd, T, materialize=False)
    np.testing.assert_allclose(r.loss.cpu().numpy(), loss, atol=1e-4, rtol=1e-4)
    dq, ditem, dneg = ops.inbatch_softmax_backward(qd, itd, ngd, r.lse, pd, nd, T)
    tol = dict(atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(dq.cpu().numpy(), gq, **tol)
    np.testing.assert_allclose(ditem.cpu().numpy(), git, **tol)
    np.testing.assert_allclose(dneg.cpu().numpy(), gng, **tol)
    fused = ops.inbatch_softmax_train(qd, itd, ngd, pd, nd, T)
    assert fused is not None
    r2, dq2, ditem2 = fused
    np.testing.assert_allclose(r2.loss.cpu().numpy(), loss, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(r2.lse.cpu().numpy(), r.lse.cpu().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(dq2.cpu().numpy(), gq, **tol)
    np.testing.assert_allclose(ditem2.cpu().numpy(), git, **tol)
    _, _, dneg2 = ops.inbatch_softmax_backward(qd, itd, ngd, r2.lse, pd, nd, T, need_dq=False)
    np.testing.assert_allclose(dneg2.cpu().numpy(), gng, **tol)


def test_scorer_fused_pass_rescale_branch(device):
    """The fused forward+dq pass rescales its accumulators lazily (only when a tile exceeds the reference max by 2^16).
    Force that branch: negatives far above the positive logit, growing along the stream, and one late spike."""
    rng = np.random.default_rng(11)
    B, Nn, E, T = 300, 900, 64, 0.1
    q = rng.normal(size=(B, E)).astype(np.float32) * 0.5
    it = (-q + rng.normal(size=(B, E)).astype(np.float32) * 0.1).astype(np.float32)  # strongly negative positives
    ng = rng.normal(size=(Nn, E)).astype(np.float32) * 0.5
    ng *= np.linspace(0.2, 3.0, Nn, dtype=np.float32)[:, None]   # maxima keep growing along the stream
    ng[700] = q[17] * 6.0                                         # late spike for one row
    loss, gq, git, gng = _autograd_scorer(q, it, ng, None, None, T)
    qd, itd, ngd = _t(q, device), _t(it, device), _t(ng, device)
    r, dq, ditem = ops.inbatch_softmax_train(qd, itd, ngd, None, None, T)
    np.testing.assert_allclose(r.loss.cpu().numpy(), loss, atol=1e-3, rtol=1e-5)
    np.testing.assert_allclose(dq.cpu().numpy(), gq, atol=5e-6, rtol=5e-4)
    np.testing.assert_allclose(ditem.cpu().numpy(), git, atol=5e-6, rtol=5e-4)
    ref = ops.inbatch_softmax(qd, itd, ngd, None, None, T, materialize=False)
    np.testing.assert_allclose(r.lse.cpu().numpy(), ref.lse.cpu().numpy(), atol=1e-4, rtol=1e-6)


def test_scorer_logq_golden_vectors(device):
    """logQ sampling correction through the HIP scorer against vectors produced by the reference's own code
    (tests/golden/make_golden.py): before the false-negative rescoring (contrastive.py:309-319) and as the
    PopularityLogitsCorrection post block (transforms/bias.py:238-254)."""
    from pathlib import Path

    G = np.load(Path(__file__).parent / "golden" / "reference_vectors.npz")
    t = lambda a: _t(a, device)
    q, pos, neg, pid, nid = t(G["sc_q"]), t(G["sc_pos"]), t(G["sc_neg"]), t(G["sc_pos_id"]), t(G["sc_neg_id"])
    r = ops.inbatch_softmax(q, pos, neg, pid, nid, pos_logq=torch.log(t(G["lq_ppos"]) + 1e-16),
                            neg_logq=torch.log(t(G["lq_pneg"]) + 1e-16))
    np.testing.assert_allclose(r.logits.cpu().numpy(), G["lq_logits"], atol=ATOL)
    probs, reg = t(G["pc_probs"]), float(G["pc_reg"])
    r = ops.inbatch_softmax(q, pos, neg, pid, nid, pos_logq=reg * torch.log(probs[pid.long()] + 1e-16),
                            neg_logq=reg * torch.log(probs[nid.long()] + 1e-16), logq_after_mask=True)
    np.testing.assert_allclose(r.logits.cpu().numpy(), G["pc_logits"], atol=ATOL)


@pytest.mark.parametrize("after_mask", [False, True])
@pytest.mark.parametrize("B,Nn,E", [(300, 450, 64), (129, 129, 128), (70, 33, 24), (64, 200, 160), (600, 1000, 128), (700, 97, 128)])
def test_scorer_logq_forward_backward(device, B, Nn, E, after_mask):
    """Forward (materialised and fused), the fused forward+dq pass and both backward passes with the logQ terms,
    against the oracle and fp64 autograd of the corrected logits (E = 160 takes the tiled E > 128 kernels; at E = 128 the passes
    without logits run the six-term split kernel by default, whose epilogue applies the corrections in either order)."""
    rng = np.random.default_rng(B + Nn + E)
    T = 0.7
    q, it, ng = (rng.normal(size=s).astype(np.float32) * 0.4 for s in ((B, E), (B, E), (Nn, E)))
    pid = rng.integers(0, 40, size=B).astype(np.int64)
    nid = rng.integers(0, 40, size=Nn).astype(np.int64)
    pp = (rng.random(B) * 0.3 + 1e-3).astype(np.float32)
    pn = (rng.random(Nn) * 0.3 + 1e-3).astype(np.float32)
    kw = (dict(post_positive_prob=pp, post_negative_prob=pn) if after_mask
          else dict(positive_sampling_prob=pp, negative_sampling_prob=pn))
    logits, _ = O.contrastive_outputs(q, it, ng, pid, nid, temperature=T, **kw)
    loss, lse = O.softmax_ce_first_column(logits)
    # fp64 autograd of the same corrected logits
    qt, itt, ngt = (torch.from_numpy(a).double().requires_grad_() for a in (q, it, ng))
    lp, ln = torch.log(torch.from_numpy(pp).double() + 1e-16), torch.log(torch.from_numpy(pn).double() + 1e-16)
    posd = (qt * itt).sum(-1, keepdim=True)
    negd = qt @ ngt.T
    m = torch.from_numpy(pid)[:, None] == torch.from_numpy(nid)[None, :]
    fns = float(np.float32(O.MIN_FLOAT))
    if after_mask:
        negd = torch.where(m, torch.full_like(negd, fns), negd) - ln[None, :]
    else:
        negd = torch.where(m, torch.full_like(negd, fns), negd - ln[None, :])
    z = torch.cat([posd - lp[:, None], negd], 1) / T
    (torch.logsumexp(z, 1) - z[:, 0]).mean().backward()
    dev = lambda a: _t(a, device)
    args = (dev(q), dev(it), dev(ng), dev(pid), dev(nid), T)
    lkw = dict(pos_logq=torch.log(dev(pp) + 1e-16), neg_logq=torch.log(dev(pn) + 1e-16), logq_after_mask=after_mask)
    r = ops.inbatch_softmax(*args, **lkw)
    np.testing.assert_allclose(r.logits.cpu().numpy(), logits, atol=ATOL * 2, rtol=1e-5)
    np.testing.assert_allclose(r.loss.cpu().numpy(), loss, atol=ATOL, rtol=1e-4)
    tol = dict(atol=2e-6, rtol=3e-4)
    dq, ditem, dneg = ops.inbatch_softmax_backward(*args[:3], r.lse, *args[3:], **lkw)
    np.testing.assert_allclose(dq.cpu().numpy(), qt.grad.numpy(), **tol)
    np.testing.assert_allclose(ditem.cpu().numpy(), itt.grad.numpy(), **tol)
    np.testing.assert_allclose(dneg.cpu().numpy(), ngt.grad.numpy(), **tol)
    rf = ops.inbatch_softmax(*args, materialize=False, **lkw)  # loss / lse without logits
    np.testing.assert_allclose(rf.loss.cpu().numpy(), loss, atol=ATOL, rtol=1e-4)
    np.testing.assert_allclose(rf.lse.cpu().numpy(), lse, atol=ATOL, rtol=1e-5)
    fused = ops.inbatch_softmax_train(*args, **lkw)
    if E <= 128:
        r2, dq2, ditem2 = fused
        np.testing.assert_allclose(r2.loss.cpu().numpy(), loss, atol=ATOL, rtol=1e-4)
        np.testing.assert_allclose(dq2.cpu().numpy(), qt.grad.numpy(), **tol)
        np.testing.assert_allclose(ditem2.cpu().numpy(), itt.grad.numpy(), **tol)
    else:
        assert fused is None


@pytest.mark.parametrize("Bq,N,E,k", [(5, 40, 8, 7), (130, 5000, 64, 100), (64, 70000, 128, 10), (3, 300, 32, 300), (257, 1025, 16, 1)])
def test_topk_bit_exact_vs_c_oracle(device, Bq, N, E, k):
    rng = np.random.default_rng(Bq + N)
    q = rng.normal(size=(Bq, E)).astype(np.float32)
    c = rng.normal(size=(N, E)).astype(np.float32)
    ids = rng.permutation(10 * N)[:N].astype(np.int32)
    vals, out_ids, idx = cbind.bruteforce_topk(q, c, ids, k)
    s, i, ix = ops.topk_dot(_t(q, device), _t(c, device), _t(ids, device), k)
    np.testing.assert_array_equal(ix.cpu().numpy(), idx)   # indices bit-exact
    np.testing.assert_array_equal(i.cpu().numpy(), out_ids)
    np.testing.assert_array_equal(s.cpu().numpy(), vals)    # same fmaf chains -> same bits
    assert i.dtype == torch.int32


def test_topk_ties_lowest_index_first(device):
    # tests/unit/tf/utils/test_tf_utils.py:42-75 row 3: all-equal scores -> indices 0..k-1
    q = np.ones((3, 4), np.float32)
    c = np.ones((500, 4), np.float32)
    c[100:] *= 0.5
    s, i, ix = ops.topk_dot(_t(q, device), _t(c, device), None, 20)
    np.testing.assert_array_equal(ix.cpu().numpy(), np.tile(np.arange(20, dtype=np.int32), (3, 1)))
    # quantised scores: massive ties across chunk boundaries
    rng = np.random.default_rng(0)
    q = rng.integers(-2, 3, size=(17, 8)).astype(np.float32)
    c = rng.integers(-2, 3, size=(3000, 8)).astype(np.float32)
    s, i, ix = ops.topk_dot(_t(q, device), _t(c, device), None, 50)
    v0, i0 = O.top_k(q @ c.T, 50)
    np.testing.assert_array_equal(ix.cpu().numpy(), i0)
    np.testing.assert_array_equal(s.cpu().numpy(), v0)


def test_topk_full_size_property(device):
    """BASELINE config-3 shape (1M x 128 catalogue, k=100) on a query block: the k-th score is a
    threshold -- exactly k candidates are >= it after tie handling, and results are sorted."""
    g = torch.Generator(device="cpu").manual_seed(0)
    N, E, Bq, k = 1_000_000, 128, 256, 100
    c = torch.randn(N, E, generator=g).to(device)
    q = torch.randn(Bq, E, generator=g).to(device)
    s, i, ix = ops.topk_dot(q, c, None, k)
    assert torch.all(s[:, :-1] >= s[:, 1:])
    full = q[:8] @ c.T  # torch GEMM as an independent (non-bit-exact) cross-check on 8 rows
    ref_s, ref_i = torch.topk(full, k, dim=1)
    torch.testing.assert_close(s[:8], ref_s, atol=1e-3, rtol=1e-5)
    assert (ix[:8].long() == ref_i).float().mean() > 0.99


@pytest.mark.parametrize("M,d", [(300, 96), (129, 200), (64, 3341)])
def test_cross_layer_matches_oracle(device, M, d):
    rng = np.random.default_rng(d)
    x0 = rng.normal(size=(M, d)).astype(np.float32)
    x = rng.normal(size=(M, d)).astype(np.float32)
    W = (rng.normal(size=(d, d)) * 0.05).astype(np.float32)
    b = (rng.normal(size=d) * 0.1).astype(np.float32)
    out = ops.cross_layer(_t(x0, device), _t(x, device), _t(W, device), _t(b, device)).cpu().numpy()
    np.testing.assert_allclose(out, O.cross_layer(x0, x, W, b), atol=ATOL * max(1.0, d / 512), rtol=1e-4)


def test_l2norm_and_rowwise_dot(device):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(300, 128)).astype(np.float32)
    y = rng.normal(size=(300, 128)).astype(np.float32)
    np.testing.assert_allclose(ops.l2norm(_t(x, device)).cpu().numpy(), O.l2norm(x), atol=1e-6)
    np.testing.assert_allclose(ops.rowwise_dot(_t(x, device), _t(y, device)).cpu().numpy()[:, 0], (x * y).sum(-1), atol=1e-4)


@pytest.mark.parametrize("filt,E", [("stream", 32), ("tiled", 32), ("tiled", 128), ("stream", 128), ("tiled", 40)])
@pytest.mark.parametrize("order", ["random", "ascending", "ties"])
def test_topk_fused_filter_path_bit_exact(device, order, filt, E, monkeypatch):
    """N large enough for the fused-filter stages; 'ascending' makes every later candidate a survivor
    (compact-list overflow -> dense fallback); 'ties' quantises scores so survivors tie across stages.  Both MFMA filters:
    the row-stationary stream kernel and the tiled kernel on the second-generation GEMM core (transposed product)."""
    monkeypatch.setenv("MERLIN_HIP_TOPK_FILTER", filt)
    rng = np.random.default_rng(11)
    Bq, N, k = 33, 300_000, 50
    if E == 128:
        Bq, N = 140, 120_000  # two query column tiles of the tiled kernel, one ragged
    q = np.abs(rng.normal(size=(Bq, E))).astype(np.float32)
    if order == "ties":
        q = rng.integers(0, 3, size=(Bq, E)).astype(np.float32)
        c = rng.integers(0, 3, size=(N, E)).astype(np.float32)
    else:
        c = np.abs(rng.normal(size=(N, E))).astype(np.float32)
        if order == "ascending":
            c = c[np.argsort(c.sum(1))]  # scores grow with the index for every (positive) query
    vals, out_ids, idx = cbind.bruteforce_topk(q, c, None, k)
    s, i, ix = ops.topk_dot(_t(q, device), _t(c, device), None, k)
    np.testing.assert_array_equal(ix.cpu().numpy(), idx)
    np.testing.assert_array_equal(s.cpu().numpy(), vals)


def test_topk_metrics_match_reference_literals_and_oracle(device):
    # tests/unit/tf/metrics/test_metrics_topk.py:49-140
    labels = np.array([[0, 1, 0, 1, 0], [1, 0, 0, 1, 0], [0, 0, 0, 0, 1]], np.float32)
    preds = np.array([[10, 9, 8, 7, 6], [1, 4, 3, 2, 5], [10, 9, 8, 7, 6]], np.float32)
    _, y, cnt = O.extract_topk(5, preds, labels)
    got = ops.topk_metrics(_t(y, device), 4, _t(cnt.astype(np.float32), device)).cpu().numpy()
    np.testing.assert_allclose(got[:, 0], [1.0, 0.5, 0.0], atol=1e-6)            # recall
    np.testing.assert_allclose(got[:, 1], [0.5, 0.25, 0.0], atol=1e-6)           # precision
    np.testing.assert_allclose(got[:, 2], [(1 / 2 + 2 / 4) / 2, (1 / 4) / 2, 0], atol=1e-6)  # MAP
    np.testing.assert_allclose(got[:, 5], [0.5, 0.25, 0.0], atol=1e-6)           # MRR
    np.testing.assert_allclose(got, O.topk_metrics(y, cnt, 4), atol=1e-6)
    rng = np.random.default_rng(0)
    y2 = (rng.random((500, 100)) < 0.05).astype(np.float32)
    c2 = y2.sum(1) + rng.integers(0, 3, 500)
    got = ops.topk_metrics(_t(y2, device), 20, _t(c2.astype(np.float32), device)).cpu().numpy()
    np.testing.assert_allclose(got, O.topk_metrics(y2, c2, 20), atol=1e-5, rtol=1e-5)
    # retrieval evaluation loop: BruteForce testing mode -> one-hot hit labels -> recall@k
    q = rng.normal(size=(64, 16)).astype(np.float32)
    c = rng.normal(size=(500, 16)).astype(np.float32)
    target = np.argsort(-(q @ c.T), axis=1)[:, 3]  # the true item is always ranked 4th
    import models_amd as mm
    pred = mm.BruteForce(10).index(_t(c, device)).forward(_t(q, device), targets=_t(target.astype(np.int64), device), testing=True)
    m = ops.topk_metrics(pred.targets, 10).cpu().numpy()
    np.testing.assert_allclose(m[:, 0], 1.0)
    np.testing.assert_allclose(m[:, 5], 0.25)


def test_l2norm_backward_matches_autograd_including_clamped_rows(device):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 48, generator=g)
    x[5] = 0.0        # below the clamp: y = x / eps, constant scale
    x[9] *= 1e-9
    dy = torch.randn(37, 48, generator=g)
    xr = x.clone().requires_grad_()
    y = xr * torch.rsqrt(torch.clamp((xr * xr).sum(-1, keepdim=True), min=1e-12))
    (y * dy).sum().backward()
    got = ops.l2norm_backward(x.to(device), dy.to(device)).cpu()
    torch.testing.assert_close(got, xr.grad, atol=1e-3, rtol=1e-4)  # rows 5 / 9 carry a 1e6 scale
    torch.testing.assert_close(got[[0, 1, 2, 20]], xr.grad[[0, 1, 2, 20]], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("B,Nn,E", [(300, 700, 24), (130, 1100, 128), (513, 257, 64)])
@pytest.mark.parametrize("ids_dtype", [None, np.int32, np.int64])
@pytest.mark.parametrize("logq", [None, "before", "after"])
def test_tiled_forward_scorer_matches_oracle_and_stream_kernel(device, B, Nn, E, ids_dtype, logq, monkeypatch):
    """The forward-only scorer on the second-generation GEMM core (mh_scorer_tiled.hip: transposed product, register softmax)
    against the numpy oracle (logits -> loss / lse) and against the row-stationary stream kernel on the same inputs: ragged
    candidate / query tiles, false-negative mask with both id widths, logQ before and after the mask."""
    if E == 128:
        _pins_f32_arithmetic()
    rng = np.random.default_rng(B + Nn + E)
    q = (rng.normal(size=(B, E)) * 0.3).astype(np.float32)
    it = (rng.normal(size=(B, E)) * 0.3).astype(np.float32)
    ng = (rng.normal(size=(Nn, E)) * 0.3).astype(np.float32)
    pid = nid = None
    if ids_dtype is not None:
        pid = rng.integers(0, 40, size=B).astype(ids_dtype)
        nid = rng.integers(0, 40, size=Nn).astype(ids_dtype)
    plq = nlq = None
    if logq is not None:
        plq = np.log(rng.uniform(0.01, 0.2, size=B)).astype(np.float32)
        nlq = np.log(rng.uniform(0.01, 0.2, size=Nn)).astype(np.float32)
    T = 0.25
    kw = dict(pos_logq=None if plq is None else _t(plq, device), neg_logq=None if nlq is None else _t(nlq, device),
              logq_after_mask=(logq == "after"))
    args = (_t(q, device), _t(it, device), _t(ng, device), None if pid is None else _t(pid, device),
            None if nid is None else _t(nid, device), T)
    monkeypatch.setenv("MERLIN_HIP_SCORER_FWD", "tiled")
    rt = ops.inbatch_softmax(*args, materialize=False, **kw)
    monkeypatch.setenv("MERLIN_HIP_SCORER_FWD", "stream")
    rs = ops.inbatch_softmax(*args, materialize=False, **kw)
    full = ops.inbatch_softmax(*args, materialize=True, **kw)  # materialised logits (stream kernel) -> fp64 log-sum-exp
    z = full.logits.double()
    lse = torch.logsumexp(z, dim=1)
    np.testing.assert_allclose(rt.lse.cpu().numpy(), lse.cpu().numpy(), atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(rt.loss.cpu().numpy(), (lse - z[:, 0]).cpu().numpy(), atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(rt.lse.cpu().numpy(), rs.lse.cpu().numpy(), atol=2e-5, rtol=1e-6)
    np.testing.assert_allclose(rt.loss.cpu().numpy(), rs.loss.cpu().numpy(), atol=2e-5, rtol=1e-6)
    if logq is None:  # the oracle's own statement of the logits
        ref, _ = O.contrastive_outputs(q, it, ng, pid, nid, temperature=T, downscore_false_negatives=pid is not None)
        r64 = torch.from_numpy(ref).double()
        np.testing.assert_allclose(rt.lse.cpu().numpy(), torch.logsumexp(r64, dim=1).numpy(), atol=ATOL, rtol=1e-5)
