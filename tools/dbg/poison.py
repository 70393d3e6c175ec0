"""Debug: does any kernel of a train step read memory it did not write?  torch.empty / empty_like return NaN-filled
(float) or 0x7f-filled (integer) tensors and every cached C-ABI workspace is filled with 0xFF before each step;
the step must give bit-identical, finite results."""
import contextlib
import sys
import torch
import models_amd as mm
from models_amd import ops, schema as S

device = torch.device("cuda:0")


@contextlib.contextmanager
def poisoned():
    real_empty, real_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0xFF)
            else:
                t.fill_(0x7F7F7F7F if t.dtype == torch.int32 else 0x7F7F7F7F7F7F7F7F if t.dtype == torch.int64 else 1)
        return t

    torch.empty = lambda *a, **k: poison(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: poison(real_like(*a, **k))
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real_empty, real_like


def poison_workspaces():
    for k, buf in ops._WS.items():
        buf.fill_(0xFF)


def run(kind, poison):
    mm.set_seed(3)
    g = torch.Generator().manual_seed(9)
    if kind == "dlrm":
        schema = mm.Schema([S.categorical("a", 40), S.categorical("b", 17), S.continuous("x"), S.binary_target("y")])
        m = mm.DLRMModel(schema, embedding_dim=8, bottom_block=mm.MLPBlock([8], device=device, seed=1),
                         top_block=mm.MLPBlock([8], device=device, seed=2), device=device)
        def batch(B):
            x = {"a": torch.randint(0, 40, (B, 1), generator=g).to(device), "b": torch.randint(0, 17, (B, 1), generator=g).to(device),
                 "x": torch.rand(B, 1, generator=g).to(device)}
            return x, torch.randint(0, 2, (B, 1), generator=g).float().to(device)
    elif kind == "dlrm_big":
        cols = [S.categorical(f"c{i}", r) for i, r in enumerate([100000, 37, 5000, 3, 1000000, 250])]
        schema = mm.Schema(cols + [S.continuous("x0"), S.continuous("x1"), S.binary_target("y")])
        m = mm.DLRMModel(schema, embedding_dim=64, bottom_block=mm.MLPBlock([128, 64], device=device, seed=1),
                         top_block=mm.MLPBlock([128, 64, 32], device=device, seed=2), device=device)
        rows = [100000, 37, 5000, 3, 1000000, 250]
        def batch(B):
            x = {f"c{i}": torch.randint(0, r, (B, 1), generator=g).to(device) for i, r in enumerate(rows)}
            x["x0"] = torch.rand(B, 1, generator=g).to(device); x["x1"] = torch.rand(B, 1, generator=g).to(device)
            return x, torch.randint(0, 2, (B, 1), generator=g).float().to(device)
    else:
        schema = mm.Schema([S.categorical("user_id", 500, [S.Tags.USER, S.Tags.USER_ID]), S.categorical("item_id", 300, [S.Tags.ITEM, S.Tags.ITEM_ID]),
                            S.categorical("item_cat", 12, [S.Tags.ITEM])])
        m = mm.TwoTowerModel(schema, mm.MLPBlock([32, 16], device=device), embedding_dim=16, device=device)
        def batch(B):
            x = {"user_id": torch.randint(0, 500, (B, 1), generator=g).to(device), "item_id": torch.randint(0, 300, (B, 1), generator=g).to(device),
                 "item_cat": torch.randint(0, 12, (B, 1), generator=g).to(device)}
            return x, None
    m.compile(optimizer="adagrad", learning_rate=0.05)
    ops.SIDE.enabled = False
    Bs = [64, 64, 100, 4099] if kind != "dlrm_big" else [4096, 5000]
    losses = []
    for B in Bs:
        x, y = batch(B)
        if poison:
            poison_workspaces()
            with poisoned():
                losses.append(float(m.train_step(x, y)))
        else:
            losses.append(float(m.train_step(x, y)))
    torch.cuda.synchronize()
    return losses, [p.data.clone() for p in m.parameters()]


for kind in sys.argv[1:] or ["dlrm", "dlrm_big", "tt"]:
    l0, p0 = run(kind, False)
    l1, p1 = run(kind, True)
    print(kind, "losses", l0, l1)
    print(kind, "param diffs", ["%.1e" % float((a - b).abs().max()) for a, b in zip(p0, p1)], "nan:", [bool(torch.isnan(b).any()) for b in p1], flush=True)
