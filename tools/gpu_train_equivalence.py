"""End-to-end check of the default (fp32-grade six-term) arithmetics against the exact fp32 chains: the SAME model, the same batches, N train
steps under each setting -- per-step loss differences and the distance of the final parameters.  A learnable signal is planted (labels /
positives depend on the features) so that the losses fall and a wrong gradient would show.  All rows but the last of a model run the sparse
update WITHOUT float atomics (MERLIN_HIP_DETERMINISTIC=1): two runs of one arithmetic are then bit-identical and a row's distance is its
arithmetic's alone; the last row is the yardstick -- the SAME exact-chain arithmetic with the default float atomics (another summation order).
Usage (GPU box): python tools/gpu_train_equivalence.py > profiles/r6_train_equivalence.txt"""
import os
import sys

import numpy as np
import torch

os.environ["MERLIN_HIP_DETERMINISTIC"] = "1"  # no float atomics in the sparse update: two runs of one arithmetic are bit-identical,
#                                                what differs between the rows below is the arithmetic alone
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import models_amd as mm  # noqa: E402

LR = [0.05]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def run(build, batches, steps, env, lr=0.05, opt="adagrad"):
    lr = LR[0]
    for k in ("MERLIN_HIP_GEMM_ARITH", "MERLIN_HIP_SCORER_ARITH"):
        os.environ.pop(k, None)
    os.environ["MERLIN_HIP_DETERMINISTIC"] = "1"
    os.environ.update(env)
    mm.set_seed(11)
    model = build()
    model.compile(optimizer=opt, learning_rate=lr)
    losses = []
    for i in range(steps):
        b = batches[i % len(batches)]
        if "__label__" in b:
            x = {k: v for k, v in b.items() if k != "__label__"}
            losses.append(float(model.train_step(x, b["__label__"])))
        else:
            losses.append(float(model.train_step(b)))
    params = [p.data.double().cpu() for p in model.parameters()]  # by position: layer names carry a process-wide counter
    return np.array(losses), params


def compare(name, ref, other):
    (l0, p0), (l1, p1) = ref, other
    dl = np.abs(l0 - l1)
    gn = lambda ps: float(sum(float(a.norm()) ** 2 for a in ps) ** 0.5)
    rel = gn([a - b for a, b in zip(p0, p1)]) / gn(p0)
    moved = gn([a - b for a, b in zip(p0, pinit)]) / gn(pinit)
    print(f"  {name:34s} loss first / last {l1[0]:.6f} / {l1[-1]:.6f}   max |loss - loss_f32| {dl.max():.3e} (last step {dl[-1]:.3e})   "
          f"|params - params_f32| / |params| {rel:.3e}   (training moved them by {moved:.3e} of their norm)")


# ---- DLRM configs[1], B = 65 536, 60 steps over 4 batches whose label depends on two dense features ----------------------------------
B, steps = 65536, 60
batches = [bench.make_batch(dev, B, s) for s in range(4)]
for b in batches:
    from models_amd.synthetic import CRITEO_CONT_NAMES

    sig = (b[CRITEO_CONT_NAMES[0]] + b[CRITEO_CONT_NAMES[1]] > 1.0).float()
    b["__label__"] = sig
build = lambda: bench.build_model(dev)[0]
pinit = None
mm.set_seed(11)
pinit = [p.data.double().cpu() for p in (lambda m: (m(dict((k, v) for k, v in batches[0].items() if k != "__label__")), m)[1])(build()).parameters()]
print(f"DLRM configs[1] (26 tables + 13 dense, bottom [128,64], top [128,64,32]), B = {B}, Adagrad lr 0.05, {steps} train steps:")
ref = run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32"})
print(f"  {'exact fp32 chains (reference run)':34s} loss first / last {ref[0][0]:.6f} / {ref[0][-1]:.6f}")
compare("default (tower layers bf16x6)", ref, run(build, batches, steps, {}))
compare("exact chains, second run", ref, run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32"}))
compare("exact chains, float atomics", ref, run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32", "MERLIN_HIP_DETERMINISTIC": "0"}))

LR[0] = 1.0
# ---- TwoTower configs[2], B = 32 768, 40 steps ------------------------------------------------------------------------------------------
B, steps = 32768, 40
batches = [bench.make_twotower_batch(dev, B, s) for s in range(4)]
for b in batches:  # a learnable signal: the item's category is a function of the user's city
    b["item_category"] = (b["user_city"] % 1000).to(torch.int32)
build = lambda: bench.build_twotower(dev)[0]
mm.set_seed(11)
pinit = [p.data.double().cpu() for p in (lambda m: (m(batches[0]), m)[1])(build()).parameters()]
print(f"\nTwoTower configs[2] (towers [256,128], in-batch sampled softmax), B = {B}, Adagrad lr 1.0, {steps} train steps:")
ref = run(build, batches, steps, {"MERLIN_HIP_SCORER_ARITH": "f32", "MERLIN_HIP_GEMM_ARITH": "f32"})
print(f"  {'exact fp32 chains (reference run)':34s} loss first / last {ref[0][0]:.6f} / {ref[0][-1]:.6f}")
compare("default (scorer + towers bf16x6)", ref, run(build, batches, steps, {}))
compare("opt-in bf16x3 scorer", ref, run(build, batches, steps, {"MERLIN_HIP_SCORER_ARITH": "bf16x3"}))
compare("exact chains, second run", ref, run(build, batches, steps, {"MERLIN_HIP_SCORER_ARITH": "f32", "MERLIN_HIP_GEMM_ARITH": "f32"}))
compare("exact chains, float atomics", ref, run(build, batches, steps, {"MERLIN_HIP_SCORER_ARITH": "f32", "MERLIN_HIP_GEMM_ARITH": "f32",
                                                                       "MERLIN_HIP_DETERMINISTIC": "0"}))

# ---- DCN-v2 configs[4], B = 16 384, 20 steps ----------------------------------------------------------------------------------------------
LR[0] = 0.05
B, steps = 16384, 20
batches = [bench.make_batch(dev, B, s) for s in range(4)]
for b in batches:
    b["__label__"] = (b[CRITEO_CONT_NAMES[0]] + b[CRITEO_CONT_NAMES[1]] > 1.0).float()
build = lambda: bench.build_model(dev, dcn=True)[0]
mm.set_seed(11)
pinit = [p.data.double().cpu() for p in (lambda m: (m(dict((k, v) for k, v in batches[0].items() if k != "__label__")), m)[1])(build()).parameters()]
print(f"\nDCN-v2 configs[4] (3 full-rank cross layers at d = 3341, deep tower [512, 256]), B = {B}, Adagrad lr 0.05, {steps} train steps:")
ref = run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32"})
print(f"  {'exact fp32 chains (reference run)':34s} loss first / last {ref[0][0]:.6f} / {ref[0][-1]:.6f}")
compare("default (cross / 3341->512 bf16x6)", ref, run(build, batches, steps, {}))
compare("opt-in bf16x3", ref, run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "bf16x3"}))
compare("exact chains, second run", ref, run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32"}))
compare("exact chains, float atomics", ref, run(build, batches, steps, {"MERLIN_HIP_GEMM_ARITH": "f32", "MERLIN_HIP_DETERMINISTIC": "0"}))
