"""Debug: where do graph-replayed train steps diverge from eager ones (tiny DLRM)?  usage: fit_graph.py [interleave|serial]"""
import sys
import torch
import models_amd as mm
from models_amd import schema as S
from models_amd.graph import GraphedStep, PackedBatch

device = torch.device("cuda:0")
schema = mm.Schema([S.categorical("a", 40), S.categorical("b", 17), S.continuous("x"), S.binary_target("y")])
mode = sys.argv[1] if len(sys.argv) > 1 else "interleave"


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


def diff(tag, a, b):
    print(tag, ["%.1e" % float((x - y).abs().max()) for x, y in zip(a, b)], flush=True)


def pack(x, y):
    d = dict(x); d["__targets__"] = y
    return d


data = batches([64] * 6)
m1, m2 = build(), build()


def fn(d):
    d = dict(d)
    y = d.pop("__targets__")
    return m1.train_step(d, y)


if mode == "serial":
    s1, s2 = [], []
    m1.train_step(*data[0]); s1.append(snap(m1))
    g = GraphedStep(fn, PackedBatch(pack(*data[1])), warmup=0)
    for i in range(1, 6):
        g.replay(PackedBatch(pack(*data[i]))); s1.append(snap(m1))
    for i in range(6):
        m2.train_step(*data[i]); s2.append(snap(m2))
    for i in range(6):
        diff(f"serial step{i}", s1[i], s2[i])
else:
    m1.train_step(*data[0]); m2.train_step(*data[0])
    diff("step0", snap(m1), snap(m2))
    g = GraphedStep(fn, PackedBatch(pack(*data[1])), warmup=0)
    for i in range(1, 6):
        g.replay(PackedBatch(pack(*data[i])))
        if mode == "sync":
            torch.cuda.synchronize()
        m2.train_step(*data[i])
        if mode == "sync":
            torch.cuda.synchronize()
        diff(f"{mode} step{i}", snap(m1), snap(m2))
