#!/usr/bin/env python
"""Which torch (aten) kernels run inside a train step, and from which line of models_amd?  A TorchDispatchMode logs every aten op
that touches a tensor of >= --min-numel elements together with the innermost models_amd / bench.py frames.

    python tools/dbg/find_torch_kernels.py --workload dcn --batch 8192
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench


class Log(TorchDispatchMode):
    def __init__(self, min_numel):
        super().__init__()
        self.min_numel = min_numel
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        big = 0
        for a in list(args) + [out]:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                big = max(big, a.numel())
        name = str(func)
        if big >= self.min_numel and not any(k in name for k in ("view", "as_strided", "slice", "select", "detach", "alias", "t.default",
                                                                     "reshape", "unsqueeze", "squeeze", "expand", "permute", "transpose")):
            frames = [f for f in traceback.extract_stack() if "models_amd" in f.filename or f.filename.endswith("bench.py")]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1])
            self.hits[(name, big, where)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dcn", choices=["dcn", "dlrm"])
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--min-numel", type=int, default=1 << 20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev, emb_dim=128 if a.workload == "dcn" else 64, dcn=a.workload == "dcn")
    model.compile(optimizer="adagrad", learning_rate=0.01)
    t = bench.make_batch(dev, a.batch, 1)
    x = {k: v for k, v in t.items() if k != "__label__"}
    y = t["__label__"]
    for _ in range(3):
        model.train_step(x, y)
    torch.cuda.synchronize()
    with Log(a.min_numel) as log:
        model.train_step(x, y)
    torch.cuda.synchronize()
    for (name, n, where), c in sorted(log.hits.items(), key=lambda kv: -kv[0][1] * kv[1]):
        print(f"{c:3d} x {name:40s} numel {n:>12d}  {where}")


if __name__ == "__main__":
    main()
