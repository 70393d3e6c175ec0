import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from models_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
bad = 0
for E in (128, 64):
    for (B, Nn, ids, logq) in [(10, 1000, True, False), (65, 64, True, True), (1, 64, False, False), (31, 97, True, True), (64, 64, False, False),
                               (257, 2049, True, False), (1000, 65, True, True), (33, 4097, False, True)]:
        q, it, neg = unit(rng.normal(size=(B, E))), unit(rng.normal(size=(B, E))), unit(rng.normal(size=(Nn, E)))
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        pid = rng.integers(0, 50, size=B).astype(np.int64) if ids else None
        nid = rng.integers(0, 50, size=Nn).astype(np.int64) if ids else None
        kw = {}
        if logq:
            kw = dict(pos_logq=t(np.log(rng.random(B) * 0.3 + 1e-3).astype(np.float32)), neg_logq=t(np.log(rng.random(Nn) * 0.3 + 1e-3).astype(np.float32)),
                      logq_after_mask=bool(B % 2))
        args = (t(q), t(it), t(neg), t(pid), t(nid), 0.05, -655.04)
        outs = {}
        for mode in ("f32", "bf16x6"):
            os.environ["MERLIN_HIP_SCORER_ARITH"] = mode
            res, dq, ditem = ops.inbatch_softmax_train(*args, **kw)
            _, _, dneg = ops.inbatch_softmax_backward(args[0], args[1], args[2], res.lse, args[3], args[4], 0.05, -655.04, need_dq=False, **kw)
            fw = ops.inbatch_softmax(*args, materialize=False, **kw)
            outs[mode] = [x.double().cpu() for x in (res.loss, res.lse, dq, ditem, dneg, fw.lse)]
        errs = [float((a - b).abs().max() / max(float(a.abs().max()), 1e-30)) for a, b in zip(outs["f32"], outs["bf16x6"])]
        ok = all(e < 2e-4 for e in errs) and all(torch.isfinite(x).all() for x in outs["bf16x6"])
        bad += not ok
        print(E, B, Nn, ids, logq, "OK" if ok else "MISMATCH", ["%.1e" % e for e in errs])
print("bad:", bad)
