"""Debug: graph replays after an eager first step with side streams vs eager steps without (tiny DLRM).
usage: side_race.py <eager steps before capture with side streams on/off, e.g. "1", "0", "01", "10">"""
import sys
import torch
import models_amd as mm
from models_amd import ops, schema as S
from models_amd.graph import GraphedStep, PackedBatch

device = torch.device("cuda:0")
schema = mm.Schema([S.categorical("a", 40), S.categorical("b", 17), S.continuous("x"), S.binary_target("y")])
pre = sys.argv[1] if len(sys.argv) > 1 else "1"
snap_mode = sys.argv[2] if len(sys.argv) > 2 else "snap"
do_snap = snap_mode == "snap"


def build():
    mm.set_seed(3)
    m = mm.DLRMModel(schema, embedding_dim=8, bottom_block=mm.MLPBlock([8], device=device, seed=1),
                     top_block=mm.MLPBlock([8], device=device, seed=2), device=device)
    m.compile(optimizer="adagrad", learning_rate=0.05)
    return m


def batches(sizes):
    g = torch.Generator().manual_seed(9)
    out = []
    for B in sizes:
        x = {"a": torch.randint(0, 40, (B, 1), generator=g).to(device), "b": torch.randint(0, 17, (B, 1), generator=g).to(device),
             "x": torch.rand(B, 1, generator=g).to(device)}
        out.append((x, torch.randint(0, 2, (B, 1), generator=g).float().to(device)))
    return out


def snap(m):
    torch.cuda.synchronize()
    return [p.data.clone() for p in m.parameters()] + [p.state["accumulator"].clone() for p in m.parameters() if "accumulator" in p.state]


def pack(x, y):
    d = dict(x); d["__targets__"] = y
    return d


data = batches([64] * 8)
m1, m2 = build(), build()


def fn(d):
    d = dict(d)
    y = d.pop("__targets__")
    return m1.train_step(d, y)


n0 = len(pre)
for i, c in enumerate(pre):
    ops.SIDE.enabled = c == "1"
    m1.train_step(*data[i])
ops.SIDE.enabled = False
for i in range(n0):
    m2.train_step(*data[i])
print("pre", ["%.1e" % float((x - y).abs().max()) for x, y in zip(snap(m1), snap(m2))][:4], flush=True)
ops.SIDE.enabled = True
g = GraphedStep(fn, PackedBatch(pack(*data[n0])), warmup=0)
ops.SIDE.enabled = False
for i in range(n0, 8):
    g.replay(PackedBatch(pack(*data[i])))
    if do_snap:
        a = snap(m1)
    if snap_mode == "sync":
        torch.cuda.synchronize()
    if snap_mode == "alloc":
        junk = [torch.full((n,), 1e30, device=device) for n in (320, 136, 8, 64, 320, 136)]
    m2.train_step(*data[i])
    if snap_mode == "sync":
        torch.cuda.synchronize()
    if do_snap:
        b = snap(m2)
        print(f"step{i}", ["%.1e" % float((x - y).abs().max()) for x, y in zip(a, b)][:10], "acc max m1 %.2e m2 %.2e" % (float(a[8].max()), float(b[8].max())), flush=True)
print("end", ["%.1e" % float((x - y).abs().max()) for x, y in zip(snap(m1), snap(m2))][:10], flush=True)
