"""The multi-GPU DLRM / TwoTower / DCN steps at WORLD SIZE 2 on the HIP kernels, on ONE GPU: two processes share cuda:0 and
talk over gloo (all-to-alls staged through host memory -- RCCL refuses two ranks on one device).  What N > 1 adds to the
forced-shard W = 1 tests: a real `row % W` ownership split, requests and rows that really cross ranks, the fixed-capacity
windows after calibration, the dense-bucket reduction -- all through libmerlin_hip.so (route build, local rows, fused
gather -> interaction over the RETURNED rows, both sparse updates with their id-only halves on side streams)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _patch_gloo_for_device_tensors():
    """gloo has no all-to-all for device tensors: stage through the host (test transport only)."""
    import torch.distributed as dist

    orig = dist.all_to_all_single

    def a2a(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if not out.is_cuda:
            return orig(out, inp, output_split_sizes, input_split_sizes, group=group, async_op=async_op)
        o = torch.empty(out.shape, dtype=out.dtype)
        orig(o, inp.cpu(), output_split_sizes, input_split_sizes, group=group)
        out.copy_(o)
        return None

    dist.all_to_all_single = a2a


def _dlrm(device):
    import models_amd as mm
    from models_amd import schema as S

    cards = {"C1": 4001, "C2": 7, "C3": 2500, "C4": 33, "C5": 1200}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous(f"I{i}") for i in range(1, 4)]
    cols.append(S.binary_target("label"))
    m = mm.DLRMModel(mm.Schema(cols), embedding_dim=16, bottom_block=mm.MLPBlock([32, 16], device=device, seed=7),
                     top_block=mm.MLPBlock([32, 16], device=device, seed=17), device=device)
    m.output.to_call.seed = 99
    m.compile(optimizer="adagrad", learning_rate=0.05)
    return m, cards


def _batches(cards, world, B, steps, seed=11, skew_from=None):
    """``skew_from``: from that step on every id of the row-sharded features is even: all requests go to rank 0 (row % 2)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for s_ in range(steps):
        x = {n: torch.randint(0, v, (world, B), generator=g, dtype=torch.int32) for n, v in cards.items()}
        if skew_from is not None and s_ >= skew_from:
            for n in ("C1", "C3", "C5"):
                x[n] = x[n] // 2 * 2
        x.update({f"I{i}": torch.rand(world, B, 1, generator=g) for i in range(1, 4)})
        out.append((x, torch.randint(0, 2, (world, B, 1), generator=g).float()))
    return out


def _worker(rank, world, port, q, skew=False, dedup=False):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        _patch_gloo_for_device_tensors()
        from models_amd import distributed as D

        dev = torch.device("cuda:0")
        B, steps = 512, (7 if skew else 5)  # two calibration steps (dense exchange), then fixed-capacity windows
        model, cards = _dlrm(dev)  # full tables, sliced by DistributedDLRM: the same values as the one-process reference
        batches = _batches(cards, world, B, steps, skew_from=3 if skew else None)
        mine = lambda x: {k: v[rank].to(dev) for k, v in x.items()}
        model(mine(batches[0][0]))
        dd = D.DistributedDLRM(model, shard_threshold=1000, dedup=dedup, capacity_factor=1.25)  # the window the skew scenario assumes
        assert sorted(dd.sharded) == ["C1", "C3", "C5"] and dd.group_sh.dedup == dedup
        losses = [float(dd.train_step(mine(x), y[rank].to(dev))) for x, y in batches]
        dd.check_overflow()
        assert dd.group_sh.spills == (1 if skew else 0), dd.group_sh.spills  # the first skewed call overflowed its window: served densely
        if dedup and not skew:  # the windows hold DISTINCT keys: 1.25 x ~half of them, below the request count 3 x 512
            assert dd.group_sh.capacity < 3 * B, dd.group_sh.capacity
        if os.environ.get("MERLIN_HIP_FUSED_DLRM", "1") != "0":  # the default configuration
            assert model.body._fused, "the sharded step should run the fused gather -> interaction kernels"
        pred = dd(mine(batches[0][0])).cpu().numpy()
        state = {"loss": losses, "pred": pred,
                 "dense": [p.data.cpu().numpy().copy() for p in model.parameters() if not p.sparse],
                 "rep": {n: model.body.embeddings.feature_table[n].table.data.cpu().numpy().copy() for n in dd.replicated},
                 "shard": {n: dd.sharded[n].cpu().numpy().copy() for n in dd.sharded}}
        q.put((rank, "ok", state))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc(), None))


@pytest.mark.parametrize("dedup", [False, True])
@pytest.mark.parametrize("skew", [False, True])
def test_sharded_dlrm_step_world2_on_hip_matches_the_full_batch_model(device, skew, dedup):
    """dedup: mh_route_build_dedup + the sender-side segment sum of the gradient rows (the fused sparse update onto zeros).
    skew: after the windows are frozen every sharded id becomes even (all requests to rank 0, twice the calibrated count):
    the overflowing call must be served without losing a request (dense exchange, window re-derived) -- through the real route
    kernels, with a rank that owns NONE of the requested rows (zero-row gathers and updates)."""
    import torch.multiprocessing as mp

    from models_amd import distributed as D

    world, B, steps = 2, 512, (7 if skew else 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, skew, dedup)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(m == "ok" for _, m, _ in res), [m for _, m, _ in res]
    # reference: ONE model on the concatenated batch (plain RankingModel.train_step, same kernels, one process)
    model, cards = _dlrm(device)
    batches = _batches(cards, world, B, steps, skew_from=3 if skew else None)
    full = lambda x: {k: v.reshape(world * B, *v.shape[2:]).to(device) for k, v in x.items()}
    model(full(batches[0][0]))
    ref_losses = [float(model.train_step(full(x), y.reshape(world * B, 1).to(device))) for x, y in batches]
    ref_pred = model(full(batches[0][0])).cpu().numpy().reshape(world, B, 1)
    ref_dense = [p.data.cpu().numpy() for p in model.parameters() if not p.sparse]
    tab = lambda n: model.body.embeddings.feature_table[n].table.data.cpu()
    for rank, _, st in res:
        np.testing.assert_allclose(st["loss"], ref_losses, rtol=2e-5, atol=2e-6)   # every rank reports the GLOBAL loss
        np.testing.assert_allclose(st["pred"], ref_pred[rank], atol=1e-4)           # forward of the trained sharded model
        for a, b in zip(st["dense"], ref_dense):
            np.testing.assert_allclose(a, b, atol=5e-5, rtol=1e-4)
        for n, t in st["rep"].items():
            np.testing.assert_allclose(t, tab(n).numpy(), atol=5e-5, rtol=1e-4)
        for n, t in st["shard"].items():
            np.testing.assert_allclose(t, D.shard_table(tab(n), rank, world).numpy(), atol=5e-5, rtol=1e-4)


# ---- generic DistributedModel (TwoTower configs[2], DCN-v2 configs[4]) at world size 2 on the HIP kernels -------------------
def _tt(device):
    import models_amd as mm
    from models_amd import schema as S

    cards = {"user_id": 3001, "user_age": 9, "item_id": 2003, "item_cat": 17}
    schema = mm.Schema([S.categorical("user_id", 3001, [S.Tags.USER, S.Tags.USER_ID]), S.categorical("user_age", 9, [S.Tags.USER]),
                        S.categorical("item_id", 2003, [S.Tags.ITEM, S.Tags.ITEM_ID]), S.categorical("item_cat", 17, [S.Tags.ITEM])])
    m = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device, seed=3), embedding_dim=16, device=device)
    m.compile(optimizer="adagrad", learning_rate=0.05)
    return m, cards, ()


def _dcn(device):
    import models_amd as mm
    from models_amd import schema as S

    cards = {"C1": 4001, "C2": 7, "C3": 2500}
    cols = [S.categorical(n, v) for n, v in cards.items()] + [S.continuous("I1"), S.continuous("I2"), S.binary_target("label")]
    m = mm.DCNModel(mm.Schema(cols), depth=2, deep_block=mm.MLPBlock([32, 16], device=device, seed=5), embedding_dim=16, device=device)
    m.compile(optimizer="adagrad", learning_rate=0.05)
    return m, cards, ("I1", "I2")


def _reseed(model):
    """Identical dense weights in every process (some blocks draw their seeds from a construction counter)."""
    for i, p in enumerate(q for q in model.parameters() if not q.sparse):
        g = torch.Generator().manual_seed(1000 + i)
        p.data.copy_((torch.randn(p.data.shape, generator=g) * (0.0 if p.data.dim() == 1 else 0.15)).to(p.data.device))


def _gbatches(cards, conts, world, B, steps, same, seed=23):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(steps):
        x = {n: torch.randint(0, v, (world, B, 1), generator=g) for n, v in cards.items()}
        x.update({c: torch.rand(world, B, 1, generator=g) for c in conts})
        y = torch.randint(0, 2, (world, B, 1), generator=g).float()
        if same:
            x = {k: v[:1].expand(world, *v.shape[1:]).contiguous() for k, v in x.items()}
        out.append((x, y))
    return out


def _generic_worker(rank, world, port, q, kind):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        _patch_gloo_for_device_tensors()
        from models_amd import distributed as D
        from models_amd.inputs import EmbeddingsBlock

        dev = torch.device("cuda:0")
        B, steps = 256, 4
        model, cards, conts = (_tt if kind == "twotower" else _dcn)(dev)
        batches = _gbatches(cards, conts, world, B, steps, same=kind == "twotower")
        mine = lambda x: {k: v[rank].to(dev) for k, v in x.items()}
        model(mine(batches[0][0]))
        _reseed(model)
        dm = D.DistributedModel(model, shard_threshold=1000)
        losses = [float(dm.train_step(mine(x), None if kind == "twotower" else y[rank].to(dev))) for x, y in batches]
        dm.check_overflow()
        tabs = {n: (t.table.data.cpu().numpy().copy(), getattr(t, "shard", None))
                for emb in model.blocks_of_type(EmbeddingsBlock) for n, t in emb.feature_table.items()}
        q.put((rank, "ok", {"loss": losses, "tabs": tabs,
                            "n_sharded": sum(len(ns) for sh in dm.shards for _, ns in sh.groups.values()),
                            "dense": [p.data.cpu().numpy().copy() for p in model.parameters() if not p.sparse]}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, "FAIL: " + traceback.format_exc(), None))


@pytest.mark.parametrize("kind", ["twotower", "dcn"])
def test_distributed_model_world2_on_hip(device, kind):
    """dcn: two ranks with half a batch each end with the parameters of ONE model trained on the concatenated batch;
    twotower: in-batch negatives are rank-local (tf/blocks/retrieval/base.py:329-375), so both ranks get the SAME batch and
    the summed, 1/W-scaled gradients must reproduce the one-process model trained on it (tests/test_distributed.py states
    the same on CPU with framework-op kernels; here the HIP kernels run)."""
    import torch.multiprocessing as mp

    from models_amd.inputs import EmbeddingsBlock

    world, B, steps = 2, 256, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_generic_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert all(m == "ok" for _, m, _ in res), [m for _, m, _ in res]
    model, cards, conts = (_tt if kind == "twotower" else _dcn)(device)
    batches = _gbatches(cards, conts, world, B, steps, same=kind == "twotower")
    model({k: v[0].to(device) for k, v in batches[0][0].items()})
    _reseed(model)
    ref_losses = []
    for x, y in batches:
        if kind == "twotower":
            ref_losses.append(float(model.train_step({k: v[0].to(device) for k, v in x.items()})))
        else:
            ref_losses.append(float(model.train_step({k: v.reshape(world * B, 1).to(device) for k, v in x.items()},
                                                     y.reshape(world * B, 1).to(device))))
    ref_tabs = {n: t.table.data.cpu().numpy() for emb in model.blocks_of_type(EmbeddingsBlock) for n, t in emb.feature_table.items()}
    ref_dense = [p.data.cpu().numpy() for p in model.parameters() if not p.sparse]
    for rank, _, st in res:
        assert st["n_sharded"] == 2
        np.testing.assert_allclose(st["loss"], ref_losses, rtol=5e-5, atol=5e-6)
        for a, b in zip(st["dense"], ref_dense):
            np.testing.assert_allclose(a, b, atol=1e-4, rtol=5e-4)
        for n, (t, shard) in st["tabs"].items():
            want = ref_tabs[n] if shard is None else ref_tabs[n][rank::world]
            np.testing.assert_allclose(t, want, atol=1e-4, rtol=5e-4, err_msg=n)
