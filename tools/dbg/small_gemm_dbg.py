import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from models_amd import ops
dev = torch.device("cuda")
rng = np.random.default_rng(17)
B, D, V = 1500, 64, 37
lens = rng.poisson(30, size=B); offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
vals = rng.integers(0, V, size=int(offs[-1])).astype(np.int64)
g = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32)).to(dev)
for comb in ("mean", "sqrtn", "sum"):
    for opt in ("sgd", "adagrad", "adam"):
        W0 = rng.standard_normal((V, D)).astype(np.float32)
        def fresh():
            W = torch.from_numpy(W0.copy()).to(dev)
            S = torch.full((V, D), 0.1, device=dev) if opt != "sgd" else None
            S2 = torch.full((V, D), 0.1, device=dev) if opt == "adam" else None
            return W, S, S2
        v, o = torch.from_numpy(vals).to(dev), torch.from_numpy(offs).to(dev)
        Wm, Sm, S2m = fresh()
        ops.embedding_bag_backward_multi([Wm], None if Sm is None else [Sm], [v], [o], g, [0], comb, opt, 0.05, 1e-7, None if S2m is None else [S2m])
        Ws, Ss, S2s = fresh()
        ops.embedding_bag_backward(Ws, Ss, v, o, g, comb, opt, 0.05, 1e-7, S2s)
        d = (Wm - Ws).abs()
        bad_rows = (d.max(1).values > 1e-4).nonzero().flatten().tolist()
        bad_cols = (d.max(0).values > 1e-4).nonzero().flatten().tolist()
        print(comb, opt, "max", float(d.max()), "bad rows", bad_rows[:12], "bad cols", bad_cols[:8], len(bad_cols))
