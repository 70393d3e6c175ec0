"""HIP embedding kernels vs the oracle (bit-exact for one-hot; fp32-roundoff for combiners)."""
import numpy as np
import pytest
import torch

from models_amd import ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("D", [8, 32, 64, 128, 40])
@pytest.mark.parametrize("idt", [np.int32, np.int64])
def test_gather_bit_exact(device, D, idt):
    rng = np.random.default_rng(10 + D)
    rows = [4, 1000, 37, 5003]
    B = 1037
    tabs = [rng.normal(size=(r, D)).astype(np.float32) for r in rows]
    ids = [rng.integers(0, r, size=B).astype(idt) for r in rows]
    ids[1][:7] = [-1, rows[1], rows[1] + 5, 0, rows[1] - 1, 2**31 - 1 if idt == np.int32 else 2**40, -7]  # OOR -> zero rows
    out = ops.embedding_gather([_t(t, device) for t in tabs], [_t(i, device) for i in ids])
    ref = np.stack([O.embedding_lookup(t, i) for t, i in zip(tabs, ids)], axis=1)
    assert out.shape == (B, len(rows), D)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_gather_sorted_slots_and_column_ids(device):
    rng = np.random.default_rng(3)
    names = ["C1", "C10", "C2", "b"]
    D, B = 16, 300
    tabs = {n: rng.normal(size=(50, D)).astype(np.float32) for n in names}
    ids = {n: rng.integers(0, 50, size=(B, 1)).astype(np.int64) for n in names}
    order = sorted(names)
    slots = [order.index(n) for n in names]
    out = ops.embedding_gather([_t(tabs[n], device) for n in names], [_t(ids[n], device) for n in names],
                               out_slot=slots, n_slots=len(names) + 1)
    ref = O.stack_features({n: O.embedding_lookup(tabs[n], ids[n]) for n in names})
    np.testing.assert_array_equal(out[:, : len(names)].cpu().numpy(), ref)


def test_gather_many_features_chunks(device):
    rng = np.random.default_rng(4)
    F, D, B = 70, 8, 129  # > MH_MAX_FEATURES -> two launches
    tabs = [rng.normal(size=(11 + f, D)).astype(np.float32) for f in range(F)]
    ids = [rng.integers(0, 11 + f, size=B).astype(np.int32) for f in range(F)]
    out = ops.embedding_gather([_t(t, device) for t in tabs], [_t(i, device) for i in ids])
    ref = np.stack([O.embedding_lookup(t, i) for t, i in zip(tabs, ids)], axis=1)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_gather_empty_batch(device):
    t = torch.zeros(5, 8, device=device)
    out = ops.embedding_gather([t], [torch.zeros(0, dtype=torch.int64, device=device)])
    assert out.shape == (0, 1, 8)


@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("D,mean_len", [(64, 3), (32, 20), (40, 5)])
def test_bag_matches_oracle(device, combiner, D, mean_len):
    rng = np.random.default_rng(7)
    V, B = 997, 513
    W = rng.normal(size=(V, D)).astype(np.float32)
    lens = rng.poisson(mean_len, size=B)
    lens[::17] = 0  # empty bags
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    values = rng.integers(0, V, size=int(offsets[-1])).astype(np.int64)
    values[::13] = -1  # pruned ids
    out = ops.embedding_bag(_t(W, device), _t(values, device), _t(offsets, device), combiner)
    ref = O.embedding_bag(W, values, offsets, combiner)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert np.all(out.cpu().numpy()[lens == 0] == 0)


@pytest.mark.parametrize("D", [16, 32, 64, 128])
@pytest.mark.parametrize("dense_list", [False, True])
def test_bag_lengths_that_end_inside_a_partial_group_round(device, D, dense_list):
    """The wave-cooperative kernel hands ids to its lane groups by shuffle: a chunk of 33 / 34 ids at D = 64 (4 groups) or 33 / 49 at
    D = 128 (2 groups) once made a group read the id out of a lane whose group had already left the loop (reads as 0: row 0 summed in
    place of the real row -- round-5 advisor finding).  Row 0 is poisoned here so that such a read cannot hide."""
    rng = np.random.default_rng(41)
    V = 2003
    W = rng.normal(size=(V, D)).astype(np.float32)
    W[0] = 1.0e6
    lens = np.array([33, 34, 35, 49, 50, 63, 64, 65, 97, 98, 113, 127, 128, 129, 161, 162, 1, 2, 31, 32] * 3, dtype=np.int64)
    for combiner in ("sum", "mean"):
        if dense_list:
            for L in (33, 34, 49, 97, 98):
                ids = rng.integers(1, V, size=(37, L)).astype(np.int32)
                out = ops.embedding_dense_list(_t(W, device), _t(ids, device), combiner)
                np.testing.assert_allclose(out.cpu().numpy(), O.embedding_dense_list(W, ids, combiner), rtol=1e-5, atol=1e-4)
        else:
            offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            values = rng.integers(1, V, size=int(offsets[-1])).astype(np.int64)   # never row 0
            out = ops.embedding_bag(_t(W, device), _t(values, device), _t(offsets, device), combiner)
            np.testing.assert_allclose(out.cpu().numpy(), O.embedding_bag(W, values, offsets, combiner), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("combiner", ["sum", "mean"])
@pytest.mark.parametrize("L", [1, 5, 24])
def test_dense_list_matches_oracle(device, combiner, L):
    rng = np.random.default_rng(8)
    V, B, D = 301, 257, 64
    W = rng.normal(size=(V, D)).astype(np.float32)
    ids = rng.integers(0, V, size=(B, L)).astype(np.int32)
    out = ops.embedding_dense_list(_t(W, device), _t(ids, device), combiner)
    ref = O.embedding_dense_list(W, ids, combiner)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


def test_gather_full_size_properties(device):
    """BASELINE config-2 size: 26 tables (Criteo cardinalities capped at 1M), D=64, B=65536.
    Size-independent property: out[b, f] == table_f[ids_f[b]] checked on device by a
    torch index_select (plumbing) of a sampled subset + checksum of checksums."""
    from models_amd.synthetic import CRITEO_CARDINALITIES

    g = torch.Generator(device="cpu").manual_seed(0)
    D, B = 64, 65536
    tabs = [torch.rand((v, D), generator=g).to(device) for v in CRITEO_CARDINALITIES]
    ids = [torch.randint(0, v, (B,), generator=g, dtype=torch.int32).to(device) for v in CRITEO_CARDINALITIES]
    out = ops.embedding_gather(tabs, ids)
    for f in (0, 5, 20, 25):
        ref = tabs[f].index_select(0, ids[f].long())
        assert torch.equal(out[:, f], ref)
    total = sum(t.index_select(0, i.long()).double().sum() for t, i in zip(tabs, ids))
    assert abs(out.double().sum().item() - total.item()) < 1e-6 * abs(total.item())
