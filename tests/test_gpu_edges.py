"""Edge cases: empty inputs, extreme sizes of the ABI limits, dtype / argument errors."""
import numpy as np
import pytest
import torch

from models_amd import _lib, ops
from oracle import cbind, oracle as O

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_empty_batches(device):
    W = torch.randn(8, 4, device=device)
    assert ops.linear(torch.zeros(0, 8, device=device), W).shape == (0, 4)
    assert ops.dot_interaction(torch.zeros(0, 3, 8, device=device)).shape == (0, 3)
    tab = torch.randn(10, 8, device=device)
    off = torch.zeros(1, dtype=torch.int64, device=device)
    assert ops.embedding_bag(tab, torch.zeros(0, dtype=torch.int64, device=device), off).shape == (0, 8)
    s, i, ix = ops.topk_dot(torch.zeros(0, 8, device=device), torch.randn(20, 8, device=device), None, 3)
    assert s.shape == (0, 3)


def test_all_bags_empty_and_single_giant_bag(device):
    rng = np.random.default_rng(0)
    W = rng.normal(size=(50, 16)).astype(np.float32)
    offsets = np.zeros(6, np.int64)
    out = ops.embedding_bag(_t(W, device), torch.zeros(0, dtype=torch.int64, device=device), _t(offsets, device), "mean")
    assert torch.all(out == 0)
    values = rng.integers(0, 50, size=5000).astype(np.int64)  # one bag of 5000 ids (wave-cooperative path)
    offsets = np.array([0, 5000], np.int64)
    for comb in ("sum", "mean", "sqrtn"):
        got = ops.embedding_bag(_t(W, device), _t(values, device), _t(offsets, device), comb).cpu().numpy()
        np.testing.assert_allclose(got, O.embedding_bag(W, values, offsets, comb), rtol=2e-4, atol=2e-3)


def test_abi_limit_sizes(device):
    rng = np.random.default_rng(1)
    # D = 1024 rows (the gather limit), int64 ids
    W = rng.normal(size=(33, 1024)).astype(np.float32)
    ids = rng.integers(0, 33, size=70).astype(np.int64)
    out = ops.embedding_gather([_t(W, device)], [_t(ids, device)])
    np.testing.assert_array_equal(out[:, 0].cpu().numpy(), W[ids])
    # k = 1024 (the top-k limit), k == N
    q = rng.normal(size=(3, 8)).astype(np.float32)
    c = rng.normal(size=(1024, 8)).astype(np.float32)
    s, i, ix = ops.topk_dot(_t(q, device), _t(c, device), None, 1024)
    vals, _, idx = cbind.bruteforce_topk(q, c, None, 1024)
    np.testing.assert_array_equal(ix.cpu().numpy(), idx)
    # E = 512 scorer, Nn not a multiple of the tile, B < one tile
    B, Nn, E = 37, 300, 512
    qq, it, ng = (rng.normal(size=s_).astype(np.float32) * 0.1 for s_ in ((B, E), (B, E), (Nn, E)))
    r = ops.inbatch_softmax(_t(qq, device), _t(it, device), _t(ng, device), temperature=0.3)
    ref, _ = O.contrastive_outputs(qq, it, ng, downscore_false_negatives=False, temperature=0.3)
    np.testing.assert_allclose(r.logits.cpu().numpy(), ref, atol=2e-4, rtol=1e-5)
    # F = 32 features x D = 64 (the interaction limit F*D = 2048)
    X = rng.normal(size=(65, 32, 64)).astype(np.float32)
    np.testing.assert_allclose(ops.dot_interaction(_t(X, device)).cpu().numpy(), O.dot_interaction(X), atol=2e-4, rtol=1e-5)


def test_argument_errors_are_loud(device):
    W = torch.randn(8, 4, device=device)
    with pytest.raises(ValueError):
        ops.linear(torch.randn(3, 7, device=device), W)
    with pytest.raises(ValueError):
        ops.linear(torch.randn(3, 8, device=device), W, activation="tanh")
    with pytest.raises(TypeError):
        ops.embedding_gather([torch.randn(4, 8, device=device)], [torch.zeros(3, device=device)])  # float ids
    with pytest.raises(_lib.MerlinHipError):
        ops.embedding_gather([torch.randn(4, 6, device=device)], [torch.zeros(3, dtype=torch.int64, device=device)])  # D % 4
    with pytest.raises(_lib.MerlinHipError):
        ops.topk_dot(torch.randn(2, 8, device=device), torch.randn(5, 8, device=device), None, 6)  # k > N
    with pytest.raises(_lib.MerlinHipError):
        ops.dot_interaction(torch.randn(2, 40, 8, device=device))  # F > 32
    lib = _lib.load()
    assert b"F=40" in lib.mh_last_error()


def test_int32_and_int64_ids_agree_in_backward(device):
    g = torch.Generator().manual_seed(0)
    t = torch.randn(100, 16, generator=g)
    ids = torch.randint(0, 100, (400,), generator=g)
    grad = torch.randn(400, 1, 16, generator=g)
    a, b = t.clone().to(device), t.clone().to(device)
    ops.embedding_gather_backward([a], None, [ids.to(torch.int32).to(device)], grad.to(device), [0], "sgd", 0.1)
    ops.embedding_gather_backward([b], None, [ids.to(torch.int64).to(device)], grad.to(device), [0], "sgd", 0.1)
    torch.testing.assert_close(a, b, atol=1e-6, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 255, 1237])
def test_concat_columns_equals_torch_cat_and_zero_pads(device, B):
    """mh_concat_columns (ConcatFeatures of the continuous columns, tf/core/aggregation.py:38-66): [B] / [B, w] / strided sources,
    bit-equal to torch.cat, padding columns zero, rows 16-byte aligned."""
    from models_amd import ops

    g = torch.Generator().manual_seed(B)
    wide = torch.randn(B, 7, generator=g).to(device)
    cols = [torch.randn(B, generator=g).to(device), torch.randn(B, 1, generator=g).to(device), wide[:, 2:5],
            torch.randn(B, 3, generator=g).to(device)] + [torch.randn(B, generator=g).to(device) for _ in range(9)]
    out = ops.concat_columns(cols, pad_to=4)
    ref = torch.cat([c.reshape(B, -1) for c in cols], dim=-1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    assert out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
    base = torch.as_strided(out, (B, out.stride(0)), (out.stride(0), 1))
    assert bool((base[:, ref.shape[1]:] == 0).all())
    # the block the models call
    from models_amd.core import ConcatFeatures

    agg = ConcatFeatures()
    named = {f"c{i:02d}": c for i, c in enumerate(cols)}
    assert torch.equal(agg(named), ref)


def test_mean_matches_fp64(device):
    g = torch.Generator().manual_seed(2)
    for n in (1, 255, 32768, 100003):
        x = torch.randn(n, generator=g).to(device)
        m = ops.mean(x)
        assert m.dim() == 0
        assert abs(float(m) - float(x.double().mean())) < 1e-6
        assert float(ops.mean(x)) == float(m)  # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("M,ld,width", [(1, 4, 3), (257, 28, 26), (4096, 3344, 3341), (5, 8, 8)])
def test_zero_pad_columns_touches_only_the_pad(device, M, ld, width):
    """``mh_fill_columns``: the pad columns of an ld-pitched buffer become zero, nothing else changes."""
    from models_amd import ops
    buf = torch.full((M, ld), 3.5, device=device)
    ops.zero_pad_columns(buf, width)
    assert bool((buf[:, :width] == 3.5).all()) and bool((buf[:, width:] == 0).all())
