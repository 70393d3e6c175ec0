"""Top-k with the threshold filter on the bf16 matrix pipe (mh_topk_split / mh_topk_dot_split, BruteForce.index + call,
tf/outputs/topk.py:124-237): the RESULT must be what the fp32 pipeline and oracle/oracle_c.c return, bit for bit -- scores are
the exact k-ascending fmaf chains, ties go to the lower index -- whatever the approximate filter saw."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import ops
from oracle import cbind

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _bf16_rne(x):
    u = x.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def test_split_kernel_bits_and_norm(device):
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(1000, 128)) * np.exp(rng.normal(size=(1000, 1)) * 3)).astype(np.float32)
    sp = ops.TopKSplit(_t(x, device))
    hi = _bf16_rne(x)
    hif = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = _bf16_rne((x - hif).astype(np.float32))
    np.testing.assert_array_equal(sp.hi.cpu().numpy().view(np.uint16), hi)
    np.testing.assert_array_equal(sp.lo.cpu().numpy().view(np.uint16), lo)
    want = (x.astype(np.float64) ** 2).sum(1).max()
    assert abs(float(sp.norm2_max.item()) - want) <= 1e-5 * want
    # the residual of the two-term split is below 2^-16 of the value (what the filter's error bound rests on)
    lof = (lo.astype(np.uint32) << 16).view(np.float32)
    assert np.all(np.abs(x - hif - lof) <= np.abs(x) * 2.0 ** -16)


@pytest.mark.parametrize("E", [128, 64])
@pytest.mark.parametrize("order", ["random", "unit", "ties", "ascending", "duplicates"])
@pytest.mark.parametrize("k", [10, 100])
def test_split_topk_equals_fp32_pipeline_and_oracle(device, order, k, E):
    """Bq large enough that the dense bootstrap is one 8 K chunk and N = 100 001 takes the staged filter path (two stages, ragged
    last tile, ragged query block).  'ties' / 'duplicates': piles of equal scores -- rows the error-band proof cannot decide are
    recomputed exactly; 'ascending': every later candidate survives (survivor-list overflow)."""
    rng = np.random.default_rng(7)
    Bq, N = 4099, 100_001
    if order in ("ties", "duplicates", "ascending"):
        Bq = 1100  # rows that fall back to the exact recomputation cost a full scan each (N > 4 x 30 464 / ... still staged)
        N = 130_001
    if order == "ties":
        q = rng.integers(0, 3, size=(Bq, E)).astype(np.float32)
        c = rng.integers(0, 3, size=(N, E)).astype(np.float32)
    else:
        q = rng.normal(size=(Bq, E)).astype(np.float32)
        c = rng.normal(size=(N, E)).astype(np.float32)
        if order == "unit":
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            c /= np.linalg.norm(c, axis=1, keepdims=True)
        if order == "duplicates":
            c = np.tile(c[:997], (N // 997 + 1, 1))[:N]
        if order == "ascending":
            q, c = np.abs(q), np.abs(c)
            c = c[np.argsort(c.sum(1))]
    qd, cd = _t(q, device), _t(c, device)
    ids = _t(rng.permutation(3 * N)[:N].astype(np.int32), device)
    sp = ops.TopKSplit(cd)
    s1, i1, x1 = ops.topk_dot(qd, cd, ids, k, split=sp)
    s0, i0, x0 = ops.topk_dot(qd, cd, ids, k)
    assert torch.equal(x1, x0) and torch.equal(i1, i0)
    assert torch.equal(s1.view(torch.int32), s0.view(torch.int32))
    rows = np.array([0, 1, 255, 256, 257, Bq - 1])
    vals, _, idx = cbind.bruteforce_topk(q[rows], c, None, k)
    np.testing.assert_array_equal(x1.cpu().numpy()[rows], idx)
    np.testing.assert_array_equal(s1.cpu().numpy()[rows], vals)


def test_split_topk_config3_shape_equals_fp32_pipeline(device):
    """BASELINE configs[2] retrieval shape: 4096 queries x 1 M x 128, k = 100 -- every score and index of the split path equals the
    fp32 pipeline's (which the smaller shapes pin to the C oracle)."""
    g = torch.Generator(device="cpu").manual_seed(0)
    N, E, Bq, k = 1_000_000, 128, 4096, 100
    c = torch.randn(N, E, generator=g).to(device)
    q = torch.randn(Bq, E, generator=g).to(device)
    sp = ops.TopKSplit(c)
    s1, _, x1 = ops.topk_dot(q, c, None, k, split=sp)
    s0, _, x0 = ops.topk_dot(q, c, None, k)
    assert torch.equal(x1, x0) and torch.equal(s1.view(torch.int32), s0.view(torch.int32))
    vals, _, idx = cbind.bruteforce_topk(q[:2].cpu().numpy(), c.cpu().numpy(), None, k)
    np.testing.assert_array_equal(x1[:2].cpu().numpy(), idx)
    np.testing.assert_array_equal(s1[:2].cpu().numpy(), vals)


def test_bruteforce_layer_builds_the_split_at_index_time(device, monkeypatch):
    g = torch.Generator(device="cpu").manual_seed(3)
    c = torch.randn(50_000, 128, generator=g).to(device)
    q = torch.randn(4096, 128, generator=g).to(device)
    layer = mm.BruteForce(20).index(c)
    assert layer._split is not None and layer._split.shape == (50_000, 128)
    out = layer(q)
    monkeypatch.setenv("MERLIN_HIP_TOPK", "f32")
    plain = mm.BruteForce(20).index(c)
    assert plain._split is None
    ref = plain(q)
    assert torch.equal(out.identifiers, ref.identifiers) and torch.equal(out.scores, ref.scores)
    small = mm.BruteForce(5).index(torch.randn(100, 32, generator=g).to(device))  # other widths: no split, fp32 pipeline
    assert small._split is None


@pytest.mark.parametrize("Bq,N,k", [(3, 300_000, 10), (257, 280_000, 500), (64, 270_000, 1)])
def test_split_topk_few_queries_and_large_k(device, Bq, N, k):
    """One (ragged) query block and up to 256 candidate splits per stage; k = 500 -> k' = 576 (nine list registers per lane)."""
    rng = np.random.default_rng(Bq)
    q = rng.normal(size=(Bq, 128)).astype(np.float32)
    c = rng.normal(size=(N, 128)).astype(np.float32)
    qd, cd = _t(q, device), _t(c, device)
    sp = ops.TopKSplit(cd)
    s1, _, x1 = ops.topk_dot(qd, cd, None, k, split=sp)
    s0, _, x0 = ops.topk_dot(qd, cd, None, k)
    assert torch.equal(x1, x0) and torch.equal(s1.view(torch.int32), s0.view(torch.int32))
    rows = np.arange(min(Bq, 3))
    vals, _, idx = cbind.bruteforce_topk(q[rows], c, None, k)
    np.testing.assert_array_equal(x1.cpu().numpy()[rows], idx)
    np.testing.assert_array_equal(s1.cpu().numpy()[rows], vals)
