"""HIP dense / interaction kernels vs the oracle."""
import numpy as np
import pytest
import torch

from models_amd import ops
from oracle import cbind, oracle as O

pytestmark = pytest.mark.gpu

# fp32 logits tolerance of the north star / reference (tf/utils/testing_utils.py:113-119)
ATOL = 1e-4


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("M,K,N", [(257, 13, 128), (1000, 128, 64), (300, 415, 128), (129, 64, 32),
                                   (64, 100, 200), (77, 33, 7), (515, 64, 1), (5, 3341, 96)])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid"])
def test_linear_matches_oracle(device, M, K, N, act):
    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = O.glorot_uniform(rng, K, N)
    b = rng.normal(size=N).astype(np.float32) * 0.1
    y = ops.linear(_t(x, device), _t(W, device), _t(b, device), act).cpu().numpy()
    ref = O.dense(x, W, b, act)
    np.testing.assert_allclose(y, ref, atol=ATOL, rtol=1e-5)


def test_linear_is_k_ascending_fmaf_chain(device):
    """Bit-exact against the C oracle's sequential fmaf chain (documented accumulation order)."""
    rng = np.random.default_rng(0)
    M, K, N = 200, 415, 128
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = rng.normal(size=(K, N)).astype(np.float32)
    y = ops.linear(_t(x, device), _t(W, device), None, None).cpu().numpy()
    np.testing.assert_array_equal(y, cbind.gemm_nn_fmaf(x, W))


def test_linear_transpose_detecting(device):
    """A = I-like input with an asymmetric W catches row/col swaps in the C/D layout."""
    K = N = 64
    x = np.eye(K, dtype=np.float32)
    W = (np.arange(K * N, dtype=np.float32).reshape(K, N)) / 100.0
    y = ops.linear(_t(x, device), _t(W, device)).cpu().numpy()
    np.testing.assert_array_equal(y, W)


def test_linear_strided_input_and_output(device):
    rng = np.random.default_rng(1)
    M, K, N = 130, 64, 32
    big = rng.normal(size=(M, 100)).astype(np.float32)
    W = O.glorot_uniform(rng, K, N)
    xb = _t(big, device)
    outb = torch.zeros(M, 50, device=device)
    ops.linear(xb[:, 4:4 + K], _t(W, device), None, "relu", out=outb[:, 8:8 + N])
    ref = O.dense(big[:, 4:4 + K], W, None, "relu")
    np.testing.assert_allclose(outb[:, 8:8 + N].cpu().numpy(), ref, atol=ATOL)
    assert torch.all(outb[:, :8] == 0) and torch.all(outb[:, 8 + N:] == 0)


@pytest.mark.parametrize("F,D", [(27, 64), (5, 8), (16, 32), (17, 128), (32, 16), (3, 40), (27, 256)])
@pytest.mark.parametrize("with_tail", [False, True])
def test_dot_interaction_matches_oracle(device, F, D, with_tail):
    rng = np.random.default_rng(F * D)
    B = 203
    X = rng.normal(size=(B, F, D)).astype(np.float32)
    tail = X[:, -1].copy() if with_tail else None
    out = ops.dot_interaction(_t(X, device), None if tail is None else _t(tail, device)).cpu().numpy()
    ref = O.dlrm_interaction_concat(X, tail)  # [bottom | interactions], the reference's order
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, atol=ATOL * max(1.0, D / 64), rtol=1e-5)
    if with_tail:  # the other column order of the C ABI (a shortcut key sorting behind the block's name): same values, swapped parts
        alt = ops.dot_interaction(_t(X, device), _t(tail, device), tail_first=False).cpu().numpy()
        np.testing.assert_array_equal(alt, np.concatenate([out[:, D:], out[:, :D]], axis=1))


def test_dot_interaction_asymmetric_order(device):
    """Distinct one-hot rows: pair (i,j) output is 1 iff rows i and j share a hot column."""
    F, D, B = 6, 8, 2
    X = np.zeros((B, F, D), np.float32)
    X[0, 0, 1] = X[0, 3, 1] = 1.0  # only pair (0,3)
    X[1, 2, 5] = 2.0
    X[1, 5, 5] = 3.0  # only pair (2,5) -> 6
    out = ops.dot_interaction(_t(X, device)).cpu().numpy()
    pairs = [(i, j) for i in range(F) for j in range(i + 1, F)]
    exp = np.zeros((B, len(pairs)), np.float32)
    exp[0, pairs.index((0, 3))] = 1.0
    exp[1, pairs.index((2, 5))] = 6.0
    np.testing.assert_array_equal(out, exp)


@pytest.mark.parametrize("F,D,dense_pos", [(27, 64, 26), (5, 16, 2), (17, 32, 0), (9, 128, None)])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_fused_gather_interaction_fwd_bwd(device, F, D, dense_pos, idt):
    """Fused segment == gather + stack + interaction (+ shortcut concat) of the unfused kernels/oracle."""
    rng = np.random.default_rng(F * D)
    B = 301
    tabs, ids, rows_np = [], [], []
    for s_ in range(F):
        if s_ == dense_pos:
            tabs.append(None)
            ids.append(None)
            rows_np.append(None)
            continue
        V = int(rng.integers(3, 400))
        t = rng.normal(size=(V, D)).astype(np.float32)
        i = rng.integers(0, V, size=B).astype(idt)
        i[:3] = [-1, V, 0]  # out-of-range ids -> zero rows
        tabs.append(_t(t, device))
        ids.append(_t(i, device))
        rows_np.append(O.embedding_lookup(t, i))
    dense = rng.normal(size=(B, D)).astype(np.float32) if dense_pos is not None else None
    X = np.stack([dense if r is None else r for r in rows_np], axis=1)
    ref = O.dlrm_interaction_concat(X, dense)
    out = ops.dlrm_interaction_fused(tabs, ids, None if dense is None else _t(dense, device)).cpu().numpy()
    np.testing.assert_allclose(out, ref, atol=ATOL * max(1.0, D / 64), rtol=1e-5)
    dout = rng.normal(size=ref.shape).astype(np.float32)
    dx = ops.dlrm_interaction_fused_backward(tabs, ids, None if dense is None else _t(dense, device), _t(dout, device))
    dx_ref = ops.dot_interaction_backward(_t(X, device), _t(dout, device), -1 if dense_pos is None else dense_pos,
                                          0 if dense_pos is None else D)
    torch.testing.assert_close(dx, dx_ref, atol=1e-5, rtol=1e-5)
    if dense is not None:  # appended order of the C ABI: same values with the two parts of every row swapped, forward and backward
        swap = lambda a: np.concatenate([a[:, D:], a[:, :D]], axis=1)
        alt = ops.dlrm_interaction_fused(tabs, ids, _t(dense, device), tail_first=False).cpu().numpy()
        np.testing.assert_array_equal(alt, swap(out))
        dx_alt = ops.dlrm_interaction_fused_backward(tabs, ids, _t(dense, device), _t(swap(dout), device), tail_first=False)
        assert torch.equal(dx_alt, dx)
        dx_alt_u = ops.dot_interaction_backward(_t(X, device), _t(swap(dout), device), dense_pos, D, tail_first=False)
        assert torch.equal(dx_alt_u, dx_ref)


@pytest.mark.parametrize("tail_first", [True, False])
@pytest.mark.parametrize("F,D,dense_pos", [(27, 64, 26), (5, 16, 2), (17, 32, 0), (9, 128, None), (32, 16, 31), (12, 128, 3), (2, 64, 0)])
def test_fused_forward_coalesced_output_path_equals_scattered_path(device, F, D, dense_pos, tail_first):
    """Output rows that are 16-byte aligned (ld % 4 == 0) leave the fused kernel as two float4 stores per lane staged through
    the LDS slab; any other row takes the scattered dword stores.  Same arithmetic: the two agree bit for bit, the columns
    behind the row (the ld padding) are never written, and a chunk boundary (B not a multiple of the wavefront count) works."""
    rng = np.random.default_rng(F + D)
    B = 1237
    tabs, ids = [], []
    for s_ in range(F):
        if s_ == dense_pos:
            tabs.append(None)
            ids.append(None)
            continue
        V = int(rng.integers(3, 400))
        tabs.append(_t(rng.normal(size=(V, D)).astype(np.float32), device))
        ids.append(_t(rng.integers(0, V, size=B).astype(np.int32), device))
    dense = None if dense_pos is None else _t(rng.normal(size=(B, D)).astype(np.float32), device)
    n = F * (F - 1) // 2 + (D if dense_pos is not None else 0)
    ld_a = (n + 3) // 4 * 4 + 4           # aligned rows with at least 4 floats of padding
    ld_u = ld_a + 1                       # rows that are not 16-byte aligned
    buf_a = torch.full((B, ld_a), -7.0, device=device)
    buf_u = torch.full((B, ld_u), -7.0, device=device)
    ops.dlrm_interaction_fused(tabs, ids, dense, out=buf_a[:, :n], tail_first=tail_first)
    ops.dlrm_interaction_fused(tabs, ids, dense, out=buf_u[:, :n], tail_first=tail_first)
    assert torch.equal(buf_a[:, :n], buf_u[:, :n])
    assert bool((buf_a[:, n:] == -7.0).all()) and bool((buf_u[:, n:] == -7.0).all())
    ref = ops.dlrm_interaction_fused(tabs, ids, dense, tail_first=tail_first)
    assert torch.equal(ref, buf_a[:, :n])
    if dense is not None:
        assert torch.equal(ref[:, :D] if tail_first else ref[:, n - D:], dense)


@pytest.mark.parametrize("tail_first", [True, False])
@pytest.mark.parametrize("F,D,T", [(27, 64, 64), (5, 16, 16), (17, 32, 0), (32, 16, 16), (12, 128, 40), (2, 64, 64), (27, 64, 100)])
def test_interaction_forward_coalesced_output_path_equals_scattered_path(device, F, D, T, tail_first):
    """mh_dot_interaction_fwd: aligned output rows take the staged float4 stores, other rows (and tails wider than 64) the
    scattered ones; bit-identical, the ld padding untouched."""
    g = torch.Generator().manual_seed(F * D + T)
    B = 1237
    x = torch.randn(B, F, D, generator=g).to(device)
    tail = torch.randn(B, T, generator=g).to(device) if T else None
    n = F * (F - 1) // 2 + T
    ld_a = (n + 3) // 4 * 4 + 4
    buf_a = torch.full((B, ld_a), -7.0, device=device)
    buf_u = torch.full((B, ld_a + 1), -7.0, device=device)
    ops.dot_interaction(x, tail, out=buf_a[:, :n], tail_first=tail_first)
    ops.dot_interaction(x, tail, out=buf_u[:, :n], tail_first=tail_first)
    assert torch.equal(buf_a[:, :n], buf_u[:, :n])
    assert bool((buf_a[:, n:] == -7.0).all()) and bool((buf_u[:, n:] == -7.0).all())
    ref = O.dot_interaction(x.cpu().numpy())
    pofs, tofs = (T, 0) if tail_first else (0, n - T)
    np.testing.assert_allclose(buf_a[:, pofs:pofs + n - T].cpu().numpy(), ref, atol=ATOL * max(1.0, D / 64) * 4, rtol=1e-5)
    if T:
        assert torch.equal(buf_a[:, tofs:tofs + T], tail)


@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_fused_segment_equals_unfused_pair_bitwise_long_runs(device, idt):
    """BASELINE configs[1] geometry (26 tables + bottom-MLP row, D = 64) at a batch where every wavefront walks a run of
    several samples (id prefetch two ahead, row prefetch one ahead): the fused kernels issue the same MFMA chains as
    gather + interaction, so forward and backward are bit-identical to the unfused pair."""
    g = torch.Generator().manual_seed(5)
    B, F, D = 20011, 27, 64
    rows = [3, 17, 1000, 250000] * 6 + [7, 90000]
    order = list(range(26))
    dense_slot = 11
    tabs = [torch.randn(r, D, generator=g).to(device) for r in rows]
    ids = [torch.randint(0, r, (B,), generator=g).to(idt).to(device) for r in rows]
    ids[3][5] = -1
    ids[0][B - 1] = 3  # out of range: zero rows
    dense = torch.randn(B, D, generator=g).to(device)
    slot_tabs = tabs[:dense_slot] + [None] + tabs[dense_slot:]
    slot_ids = ids[:dense_slot] + [None] + ids[dense_slot:]
    stack = torch.empty(B, F, D, device=device)
    ops.embedding_gather(tabs, ids, out=stack, out_slot=[s for s in range(F) if s != dense_slot])
    stack[:, dense_slot] = dense
    P = F * (F - 1) // 2
    ref = torch.empty(B, P + D, device=device)
    ops.dot_interaction(stack, dense, out=ref)
    out = ops.dlrm_interaction_fused(slot_tabs, slot_ids, dense)
    assert torch.equal(out, ref)
    dout = torch.randn(B, P + D, generator=g).to(device)
    dx_ref = ops.dot_interaction_backward(stack, dout, dense_slot, D)
    dx = ops.dlrm_interaction_fused_backward(slot_tabs, slot_ids, dense, dout)
    assert torch.equal(dx, dx_ref)


# ---- second-generation GEMM core (mh_gemm2.h: DMA tiles, 3-deep ring) -- taken for N >= 256, K % 4 == 0 -----------------
@pytest.mark.parametrize("M,K,N", [(300, 100, 256), (130, 132, 260), (1000, 64, 512), (257, 416, 384), (513, 3344, 256)])
@pytest.mark.parametrize("act", [None, "relu"])
def test_wide_linear_matches_oracle_and_first_generation_core(device, M, K, N, act, monkeypatch):
    rng = np.random.default_rng(M + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = O.glorot_uniform(rng, K, N)
    b = rng.normal(size=N).astype(np.float32) * 0.1
    y = ops.linear(_t(x, device), _t(W, device), _t(b, device), act).cpu().numpy()
    np.testing.assert_allclose(y, O.dense(x, W, b, act), atol=ATOL, rtol=1e-5)
    # same k-ascending fmaf chain as the C oracle: identical bits (K tail through the zero chunks, M / N tails clamped)
    y0 = ops.linear(_t(x, device), _t(W, device), None, None).cpu().numpy()
    np.testing.assert_array_equal(y0, cbind.gemm_nn_fmaf(x, W))


def test_wide_linear_strided_operands(device):
    rng = np.random.default_rng(7)
    M, K, N = 260, 128, 256
    big = rng.normal(size=(M, 200)).astype(np.float32)
    W = O.glorot_uniform(rng, K, N)
    outb = torch.full((M, 300), -3.0, device=device)
    ops.linear(_t(big, device)[:, 8:8 + K], _t(W, device), None, "relu", out=outb[:, 12:12 + N])
    ref = O.dense(big[:, 8:8 + K], W, None, "relu")
    np.testing.assert_allclose(outb[:, 12:12 + N].cpu().numpy(), ref, atol=ATOL)
    assert float(outb[:, :12].max()) == -3.0 and float(outb[:, 12 + N:].max()) == -3.0


@pytest.mark.parametrize("d", [256, 260, 388])
def test_wide_cross_layer_matches_oracle(device, d):
    rng = np.random.default_rng(d)
    M = 333
    x0 = rng.normal(size=(M, d)).astype(np.float32)
    x = rng.normal(size=(M, d)).astype(np.float32)
    W = O.glorot_uniform(rng, d, d)
    b = rng.normal(size=d).astype(np.float32) * 0.1
    y = ops.cross_layer(_t(x0, device), _t(x, device), _t(W, device), _t(b, device)).cpu().numpy()
    np.testing.assert_allclose(y, O.cross_layer(x0, x, W, b), atol=ATOL, rtol=1e-5)


@pytest.mark.parametrize("M,d", [(333, 260), (44032, 260)])  # small: first-generation fallback; large: fused store of p
def test_cross_layer_with_saved_preactivation(device, M, d):
    rng = np.random.default_rng(d + M)
    x0 = _t(rng.normal(size=(M, d)).astype(np.float32), device)
    x = _t(rng.normal(size=(M, d)).astype(np.float32), device)
    W = _t(O.glorot_uniform(rng, d, d), device)
    b = _t(rng.normal(size=d).astype(np.float32) * 0.1, device)
    out, p = ops.cross_layer(x0, x, W, b, save_p=True)
    arith = ops.gemm_arith()
    ref = ops.linear(x, W, b, None)
    if arith == "f32":
        assert torch.equal(p, ref)                                  # p = x W + b, the same fmaf chains
    else:
        # bf16x6 (default): fp32-grade, the dropped terms are <= 2^-25 of a product; bf16x3 (opt-in): 2^-17 per operand.
        # ops.linear may run another arithmetic at this shape, so the comparison is by tolerance against p's scale.
        tol = 1e-4 if arith == "bf16x3" else 2e-6
        torch.testing.assert_close(p, ref, atol=tol * float(ref.abs().max()), rtol=tol)
    torch.testing.assert_close(out, ops.cross_layer(x0, x, W, b), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(out, x0 * p + x, atol=1e-5, rtol=1e-5)
