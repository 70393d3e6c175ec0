"""models_amd.Loader in DEVICE-CHUNK mode (GPU): pinned host columns -> chunk copies on a copy stream -> on-device shuffle ->
batches as views.  Same PrepareFeatures contract as the per-batch host path (tests/test_loader.py), checked row by row."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import schema as S

pytestmark = pytest.mark.gpu


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 5, size=n)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    vals = rng.integers(0, 50, size=int(offs[-1])).astype(np.int64)
    return {"user": np.arange(n, dtype=np.int64),  # a row id: lets the test re-identify every shuffled row
            "item": rng.integers(0, 200, size=n).astype(np.int32), "genres": (vals, offs),
            "price": rng.random(n).astype(np.float32), "click": rng.integers(0, 2, size=n).astype(np.float32)}


def _schema(with_list=True):
    cols = [S.categorical("user", 100000), S.categorical("item", 200)]
    if with_list:
        cols.append(S.categorical("genres", 50, is_list=True, is_ragged=True))
    return mm.Schema(cols + [S.continuous("price"), S.binary_target("click")])


def _check_rows(data, inputs, y, with_list=True):
    u = inputs["user"].cpu().numpy().reshape(-1)
    np.testing.assert_array_equal(inputs["item"].cpu().numpy().reshape(-1), data["item"][u])
    np.testing.assert_array_equal(inputs["price"].cpu().numpy()[:, 0], data["price"][u])
    np.testing.assert_array_equal(y.cpu().numpy()[:, 0], data["click"][u])
    assert inputs["price"].shape == (len(u), 1) and y.shape == (len(u), 1)
    if with_list:
        offs, vals = inputs["genres__offsets"].cpu().numpy(), inputs["genres__values"].cpu().numpy()
        assert offs[0] == 0 and offs.shape[0] == len(u) + 1 and offs[-1] == vals.shape[0] and offs.dtype == vals.dtype
        gv, go = data["genres"]
        for b, r in enumerate(u):
            np.testing.assert_array_equal(vals[offs[b]:offs[b + 1]], gv[go[r]:go[r + 1]])
    return u


@pytest.mark.parametrize("with_list", [False, True])
@pytest.mark.parametrize("shuffle", [False, True])
def test_device_chunk_batches_are_the_dataset_rows(device, shuffle, with_list):
    n = 1003
    data = _data(n)
    if not with_list:
        data = {k: v for k, v in data.items() if k != "genres"}
    ld = mm.Loader(data, _schema(with_list), batch_size=64, shuffle=shuffle, seed=5, device=device, device_chunk_rows=256)
    assert ld._pinned is not None and ld.device_chunk_rows == 256 and len(ld) == 16
    for epoch in range(2):
        seen = []
        for inputs, y in ld:
            assert all(t.is_cuda for t in inputs.values()) and y.is_cuda
            seen.append(_check_rows(data, inputs, y, with_list))
        sizes = [len(s) for s in seen]
        assert sum(sizes) == n and sorted(sizes)[1:] == [64] * 15 and min(sizes) == n % 64
        allu = np.concatenate(seen)
        assert np.array_equal(np.sort(allu), np.arange(n))          # every row exactly once per epoch
        if not shuffle:
            assert np.array_equal(allu, np.arange(n))
        else:
            assert not np.array_equal(allu, np.arange(n))
            if epoch == 0:
                first = allu
            else:
                assert not np.array_equal(allu, first)              # re-seeded per epoch


def test_device_chunk_mode_feeds_fit(device):
    """Model.fit over the chunk loader: the captured step replays with the batch views copied into its static inputs."""
    n, B = 4096, 256
    rng = np.random.default_rng(1)
    data = {"a": rng.integers(0, 500, size=n).astype(np.int32), "b": rng.integers(0, 50, size=n).astype(np.int32),
            "x": rng.random(n).astype(np.float32), "y": rng.integers(0, 2, size=n).astype(np.float32)}
    schema = mm.Schema([S.categorical("a", 500), S.categorical("b", 50), S.continuous("x"), S.binary_target("y")])
    model = mm.DLRMModel(schema, embedding_dim=16, bottom_block=mm.MLPBlock([16], device=device),
                         top_block=mm.MLPBlock([16, 8], device=device), device=device)
    model.compile(optimizer="adagrad", learning_rate=0.05)
    ld = mm.Loader(data, schema, batch_size=B, shuffle=True, seed=2, device=device, device_chunk_rows=1024, drop_last=True)
    h = model.fit(ld, epochs=3)
    assert len(h["loss"]) == 3 and all(np.isfinite(h["loss"])) and h["examples_per_sec"][-1] > 0
