"""mh_route_build / mh_route_local_rows (row-sharded exchange) against the
framework-op statement ``distributed.route_build_torch`` -- bit-exact (index work)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("W", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("F,B", [(1, 1), (3, 257), (8, 4099), (26, 2048)])
def test_route_build_matches_stable_sort(W, dtype, F, B):
    from models_amd import ops
    from models_amd.distributed import route_build_torch

    dev = _dev()
    g = torch.Generator().manual_seed(F * 1000 + B + W)
    ids = [torch.randint(0, 1 << 20, (B,), generator=g).to(dtype).to(dev) for _ in range(F)]
    slots = torch.randperm(F + 2, generator=g)[:F].tolist()
    got = ops.route_build(ids, W, slots, F + 2)
    want = route_build_torch(ids, W, slots, F + 2)
    for a, b, name in zip(got, want, ("send_keys", "pos_of", "src_row", "counts")):
        assert torch.equal(a, b), name


def test_route_build_skewed_and_local_rows():
    from models_amd import ops
    from models_amd.distributed import route_build_torch, route_local_rows_torch

    dev = _dev()
    W = 8
    ids = [(torch.arange(5000, device=dev) * W), torch.full((5000,), 7, device=dev, dtype=torch.int64)]  # owners 0 / 7 only
    got = ops.route_build(ids, W)
    want = route_build_torch(ids, W, [0, 1], 2)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert got[3].tolist() == [5000, 0, 0, 0, 0, 0, 0, 5000]
    base = torch.tensor([0, 123456], dtype=torch.int64, device=dev)
    assert torch.equal(ops.route_local_rows(got[0], base), route_local_rows_torch(got[0], base))


def test_route_build_empty_and_errors():
    from models_amd import ops
    from models_amd._lib import MerlinHipError

    dev = _dev()
    out = ops.route_build([torch.empty(0, dtype=torch.int64, device=dev)], 4)
    assert out[3].tolist() == [0, 0, 0, 0] and out[0].numel() == 0
    with pytest.raises(MerlinHipError):
        ops.route_build([torch.zeros(4, dtype=torch.int64, device=dev)], 65)
    with pytest.raises(MerlinHipError):
        ops.route_build([torch.zeros(4, dtype=torch.int64)], 2)  # CPU tensor: no fallback
