#!/usr/bin/env python
"""Generate golden vectors by EXECUTING the reference's own Python (run in the authoring
container, where /root/reference exists; the vectors are committed, the reference is not).

The reference packages cannot be imported here (TensorFlow / merlin-core are absent), so the few
self-contained functions of its torch backend that restate the hot-path math are extracted from
the reference SOURCE at run time with ``ast`` and executed with torch-CPU:

  * merlin/models/torch/outputs/contrastive.py: ContrastiveOutput.contrastive_outputs +
    rescore_false_negatives                      (scorer)
  * merlin/models/torch/blocks/dlrm.py: DLRMInteraction.forward       (interaction, a7)
  * merlin/models/torch/blocks/cross.py: CrossBlock.forward loop      (cross, a9)
  * merlin/models/utils/schema_utils.py: get_embedding_size_from_cardinality (a2)
  * merlin/models/torch/outputs/classification.py: BinaryOutput.DEFAULT_LOSS_CLS (nn.BCELoss) and
    merlin/models/torch/outputs/contrastive.py: ContrastiveOutput.__init__'s default `loss` (nn.CrossEntropyLoss()): the
    loss classes the reference's torch backend instantiates, evaluated on fixed inputs (BCE, softmax-CE a13)
  * the DLRM top-MLP input layout (a5 / a8), from BOTH statements the reference holds:
      - merlin/models/torch/blocks/dlrm.py: InteractionBlock.forward (+ torch/transforms/agg.py: Stack.forward)
      - the TF key logic itself: WithShortcut.__init__'s branch dict, ParallelBlock.call (tf/core/combinators.py),
        Filter.call / check_feature (tf/core/tabular.py), ConcatFeatures.call / StackFeatures.call
        (tf/core/aggregation.py), executed over a numpy shim for tf.cast / tf.concat / tf.stack
  * merlin/models/torch/blocks/mlp.py: MLPBlock.__init__ (the Dense -> activation sequence the reference builds), executed
    with torch.nn modules and fixed weights (Dense a6)

Nothing is copied into this repository: only inputs/outputs land in tests/golden/*.npz.
"""
import ast
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent


def extract(path: Path, name: str, cls: str = None):
    """Return the source of function `name` (optionally a method of class `cls`) in `path`."""
    src = path.read_text()
    tree = ast.parse(src)
    nodes = tree.body
    if cls is not None:
        nodes = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in nodes if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    fn.returns = None
    return ast.unparse(fn)


def load(path, name, cls=None, extra=None):
    from typing import Dict, Optional, Tuple, Union

    ns = {"torch": torch, "Optional": Optional, "Tuple": Tuple, "Dict": Dict, "Union": Union, "math": __import__("math")}
    ns.update(extra or {})
    exec(extract(REF / path, name, cls), ns)
    return ns[name]


def class_attr(path, cls: str, attr: str, ns):
    """Evaluate the class attribute `cls.attr` of the reference source file `path` (e.g. DEFAULT_LOSS_CLS)."""
    tree = ast.parse((REF / path).read_text())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    asg = next(n for n in node.body if isinstance(n, ast.Assign) and any(getattr(t, "id", None) == attr for t in n.targets))
    return eval(compile(ast.Expression(asg.value), str(path), "eval"), ns)


def init_default(path, cls: str, arg: str, ns):
    """Evaluate the default value of `cls.__init__`'s argument `arg` in the reference source file `path`."""
    tree = ast.parse((REF / path).read_text())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    init = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    names = [a.arg for a in init.args.args]
    dflt = init.args.defaults[names.index(arg) - (len(names) - len(init.args.defaults))]
    return eval(compile(ast.Expression(dflt), str(path), "eval"), ns)


def assign_value(path, cls: str, fn: str, target: str, ns):
    """Evaluate the right-hand side of `target = ...` inside `cls.fn` of the reference source file `path`."""
    tree = ast.parse((REF / path).read_text())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    f = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == fn)
    asg = next(n for n in ast.walk(f) if isinstance(n, ast.Assign) and any(getattr(t, "id", None) == target for t in n.targets))
    return eval(compile(ast.Expression(asg.value), str(path), "eval"), ns)


class _TFShim:
    """The three TensorFlow calls the key logic makes, on numpy arrays."""
    float32 = np.float32
    SparseTensor = type("SparseTensor", (), {})

    @staticmethod
    def cast(x, dtype):
        return np.asarray(x).astype(dtype)

    @staticmethod
    def concat(tensors, axis):
        return np.concatenate(tensors, axis=axis)

    @staticmethod
    def stack(tensors, axis):
        return np.stack(tensors, axis=axis)


def dlrm_layout(out, g):
    """What the reference feeds its top MLP: `[bottom_block | interactions]`.

    TF statement (the canonical one): DLRMBlock ends in `connect_with_shortcut(DotProductInteractionBlock(),
    shortcut_filter=Filter("bottom_block"), aggregation="concat")` (tf/blocks/dlrm.py:126-130).  Its pieces are executed here
    FROM THE REFERENCE SOURCE: the branch dict WithShortcut.__init__ builds, ParallelBlock.call's merge loop, Filter.call,
    ConcatFeatures.call.  What is NOT in the reference tree is Keras' automatic layer name of the interaction branch
    (`block.name`): Keras names a layer `to_snake_case(class name)` + a uniquifying suffix -- the reference's own tests show
    it for this class: "sequential_block", "sequential_block_1" in the model summaries of
    tests/unit/tf/models/test_base.py:342-345.  (SequentialBlock._get_name, tf/core/combinators.py:137-138, is used by
    __repr__ only.)  The fixture stores the order for that name and asserts that EVERY name Keras can generate for the class
    gives the same order; the reference's torch backend states the order directly, and agrees."""
    tfs = _TFShim()
    comb = "merlin/models/tf/core/combinators.py"
    B, Fc, D = 6, 5, 8
    # upper case sorts before lower case (ASCII); every name sorts before "bottom_block" (TF) AND before "continuous" (the
    # key the torch twin stacks the bottom output under), so both statements stack it in the last slot and agree column by column
    names = ["C1", "C10", "C2", "I_emb", "a_cat"][:Fc]
    emb = {n: torch.randn(B, D, generator=g).numpy() for n in names}
    bottom = torch.randn(B, D, generator=g).numpy()
    interaction_inputs = {**emb, "bottom_block": bottom}  # what ParallelBlock{"embeddings", "bottom_block"} hands on (dlrm.py:110-113)

    # -- the interaction branch: SequentialBlock(StackFeatures(axis=1), DotProductInteraction()) (dlrm.py:169-170)
    stack_call = load("merlin/models/tf/core/aggregation.py", "call", "StackFeatures", {"tf": tfs, "TabularData": dict})
    stack_self = types.SimpleNamespace(axis=1, output_dtype=np.float32, _check_concat_shapes=lambda inputs: None)
    inter_fwd = load("merlin/models/torch/blocks/dlrm.py", "forward", "DLRMInteraction")
    Inter = type("Inter", (torch.nn.Module,), {"forward": inter_fwd})

    def interaction_branch(inputs):
        return Inter()(torch.from_numpy(stack_call(stack_self, inputs))).numpy()

    # -- the shortcut branch: Filter("bottom_block")
    filt = types.SimpleNamespace(feature_names=["bottom_block"], exclude=False, pop=False, add_to_context=False)
    filt.check_feature = types.MethodType(load("merlin/models/tf/core/tabular.py", "check_feature", "Filter"), filt)
    filter_call = load("merlin/models/tf/core/tabular.py", "call", "Filter", {"TabularData": dict})

    # -- WithShortcut.__init__: inputs = {block_outputs_name: block, "shortcut": shortcut}, block_outputs_name = block.name
    def run(block_name):
        branches = assign_value(comb, "WithShortcut", "__init__", "inputs",
                                {"block_outputs_name": block_name, "block": interaction_branch,
                                 "shortcut": lambda inputs: filter_call(filt, inputs)})
        me = types.SimpleNamespace(strict=False, parallel_dict=branches,
                                   _maybe_filter_layer_inputs_using_schema=lambda name, layer, inputs: inputs)
        par_call = load(comb, "call", "ParallelBlock", {"call_layer": lambda layer, inputs, **kw: layer(inputs)})
        merged = par_call(me, dict(interaction_inputs))
        concat_call = load("merlin/models/tf/core/aggregation.py", "call", "ConcatFeatures", {"tf": tfs, "TabularData": dict})
        cself = types.SimpleNamespace(axis=-1, output_dtype=np.float32, _check_concat_shapes=lambda inputs: None)
        return list(merged.keys()), sorted(merged.keys()), concat_call(cself, merged)

    keys, order, top_in = run("sequential_block_7")
    assert "shortcut" not in keys and sorted(keys) == ["bottom_block", "sequential_block_7"], keys
    for nm in ["sequential_block"] + [f"sequential_block_{i}" for i in (1, 2, 9, 10, 123)]:
        assert run(nm)[1][0] == "bottom_block" and np.array_equal(run(nm)[2], top_in), nm
    P = (Fc + 1) * Fc // 2
    assert top_in.shape == (B, D + P) and np.array_equal(top_in[:, :D], bottom)

    # -- torch twin: InteractionBlock.forward over [Stack(dim=1), DLRMInteraction()] with the bottom output under "continuous"
    stack_fwd = load("merlin/models/torch/transforms/agg.py", "forward", "Stack")
    TStack = type("TStack", (torch.nn.Module,), {"forward": lambda self, inputs, batch=None: stack_fwd(self, inputs), "dim": 1})
    TInter = type("TInter", (torch.nn.Module,), {"forward": lambda self, inputs, batch=None: inter_fwd(self, inputs)})
    ib_fwd = load("merlin/models/torch/blocks/dlrm.py", "forward", "InteractionBlock", {"Batch": object})
    tin = {**{k: torch.from_numpy(v) for k, v in emb.items()}, "continuous": torch.from_numpy(bottom)}
    twin = ib_fwd(types.SimpleNamespace(values=[TStack(), TInter()]), tin).numpy()
    assert np.array_equal(twin[:, :D], bottom), "torch twin: continuous first"
    assert np.allclose(twin, top_in, atol=1e-5), "the two statements of the reference disagree"

    out.update(dl_names=np.array(names), dl_bottom=bottom, dl_top_in=top_in, dl_key_order=np.array(order),
               dl_merged_keys=np.array(keys), dl_twin_top_in=twin,
               **{f"dl_emb_{n}": v for n, v in emb.items()})


def dense_mlp(out, g):
    """Dense (a6) through the reference's torch MLPBlock: its __init__ (torch/blocks/mlp.py) is executed from source and
    builds `Linear -> activation` per layer; the resulting modules run with fixed weights.  Keras' Dense is
    act(x @ kernel + bias) with kernel [in, out] = Linear.weight.T."""
    tree = ast.parse((REF / "merlin/models/torch/blocks/mlp.py").read_text())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MLPBlock")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    init.decorator_list = []
    for a in init.args.args + init.args.kwonlyargs:
        a.annotation = None
    # the body's `super().__init__(*modules)` hands the module list to the container: capture it
    captured = {}
    src = ast.unparse(init).replace("super().__init__", "_capture")
    ns = {"nn": torch.nn, "_capture": lambda *m, **kw: captured.__setitem__("modules", list(m))}
    exec(src, ns)
    units, din = [24, 16, 8], 19
    ns["__init__"](types.SimpleNamespace(), units, pre_agg=torch.nn.Identity())  # default activation: the reference's (nn.ReLU)
    mods = captured["modules"]
    x = torch.randn(13, din, generator=g)
    h, li = x, 0
    for m in mods:
        if isinstance(m, torch.nn.LazyLinear):
            lin = torch.nn.Linear(h.shape[1], m.out_features)
            with torch.no_grad():
                lin.weight.copy_(torch.randn(m.out_features, h.shape[1], generator=g) * 0.3)
                lin.bias.copy_(torch.randn(m.out_features, generator=g) * 0.1)
            out[f"mlp_W{li}"] = lin.weight.detach().T.contiguous()
            out[f"mlp_b{li}"] = lin.bias.detach()
            li += 1
            m = lin
        h = m(h)
    out.update(mlp_x=x, mlp_y=h.detach(), mlp_layers=np.array([type(m).__name__ for m in mods]))


def batchnorm_mlp(out, g):
    """Dense -> BatchNormalization (``MLPBlock(normalization=...)``, tf/blocks/mlp.py:129-135) through the reference's torch
    MLPBlock, whose __init__ places the normalization module it is handed behind `Linear -> activation` (torch/blocks/mlp.py:
    70-76).  The module carries Keras' constants (epsilon 1e-3; Keras momentum 0.99 = torch momentum 0.01): the training-mode
    output, its gradients under autograd, the moving mean after one training call and the inference-mode output are then the
    Keras layer's.  One documented difference: torch feeds the UNBIASED batch variance into running_var, Keras the biased one --
    the fixture stores torch's running_var and the batch size; the test converts."""
    tree = ast.parse((REF / "merlin/models/torch/blocks/mlp.py").read_text())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MLPBlock")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    init.decorator_list = []
    for a in init.args.args + init.args.kwonlyargs:
        a.annotation = None
    captured = {}
    src = ast.unparse(init).replace("super().__init__", "_capture")
    ns = {"nn": torch.nn, "_capture": lambda *m, **kw: captured.__setitem__("modules", list(m))}
    exec(src, ns)
    din, n, M = 11, 24, 29
    bn = torch.nn.BatchNorm1d(n, eps=1e-3, momentum=0.01)
    ns["__init__"](types.SimpleNamespace(), [n], normalization=bn, pre_agg=torch.nn.Identity())
    mods = captured["modules"]
    lin = torch.nn.Linear(din, n)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(n, din, generator=g) * 0.4)
        lin.bias.copy_(torch.randn(n, generator=g) * 0.1)
        bn.weight.copy_(1.0 + 0.3 * torch.randn(n, generator=g))
        bn.bias.copy_(0.2 * torch.randn(n, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(n, generator=g))
        bn.running_var.copy_(1.0 + 0.2 * torch.rand(n, generator=g))
    out.update(bn_W=lin.weight.detach().T.contiguous(), bn_b=lin.bias.detach().clone(), bn_gamma=bn.weight.detach().clone(),
               bn_beta=bn.bias.detach().clone(), bn_mean0=bn.running_mean.clone(), bn_var0=bn.running_var.clone())
    mods = [lin if isinstance(m, torch.nn.LazyLinear) else m for m in mods]
    x = torch.randn(M, din, generator=g, requires_grad=True)
    dy = torch.randn(M, n, generator=g)

    def run(inp):
        h = inp
        for m in mods:
            h = m(h)
        return h

    bn.train()
    y = run(x)
    y.backward(dy)
    out.update(bn_x=x.detach().clone(), bn_dy=dy, bn_y_train=y.detach().clone(), bn_dx=x.grad.clone(),
               bn_dW=lin.weight.grad.T.contiguous().clone(), bn_db=lin.bias.grad.clone(), bn_dgamma=bn.weight.grad.clone(),
               bn_dbeta=bn.bias.grad.clone(), bn_mean1=bn.running_mean.clone(), bn_var1_unbiased=bn.running_var.clone(),
               bn_layers=np.array([type(m).__name__ for m in captured["modules"]]))
    bn.eval()
    with torch.no_grad():
        out["bn_y_infer"] = run(x.detach()).clone()


def adagrad_steps(out, g):
    """Adagrad through the optimizer the reference's `create_optimizer(module, "adagrad")` builds (torch/models/base.py:443-480,
    executed from source: `optim.Adagrad(params, lr=0.01)`), with Keras' initial accumulator 0.1 and epsilon 1e-7 written into
    its param group / state: two steps on a table whose gradient comes from autograd of a lookup with DUPLICATE ids (their
    rows' gradients are summed before the update: Keras' _deduplicate_indexed_slices; rows that were not looked up have
    gradient 0 and do not move)."""
    import typing

    V, D = 17, 8
    create_optimizer = load("merlin/models/torch/models/base.py", "create_optimizer", None,
                            {"optim": torch.optim, "nn": torch.nn, "Iterator": typing.Iterator})
    table = torch.nn.Embedding(V, D)
    with torch.no_grad():
        table.weight.copy_(torch.randn(V, D, generator=g))
    opt = create_optimizer(table, "adagrad")
    assert type(opt).__name__ == "Adagrad" and opt.param_groups[0]["lr"] == 0.01
    opt.param_groups[0]["eps"] = 1e-7                      # Keras' epsilon
    opt.state[table.weight]["sum"].fill_(0.1)             # Keras' initial_accumulator_value
    out.update(ada_w0=table.weight.detach().clone(), ada_lr=np.float32(0.01))
    ids, dys = [], []
    for step in range(2):
        i = torch.randint(0, V, (12,), generator=g)
        i[5] = i[0]
        i[7] = i[0]                                        # a row looked up three times
        dy = torch.randn(12, D, generator=g)
        opt.zero_grad()
        table(i).backward(dy)
        opt.step()
        ids.append(i)
        dys.append(dy)
        out[f"ada_w{step + 1}"] = table.weight.detach().clone()
        out[f"ada_acc{step + 1}"] = opt.state[table.weight]["sum"].clone()
    out.update(ada_ids=torch.stack(ids), ada_dy=torch.stack(dys))


def main():
    g = torch.Generator().manual_seed(20260925)
    out = {}

    # --- scorer ---------------------------------------------------------------------------------
    rescore = load("merlin/models/torch/outputs/contrastive.py", "rescore_false_negatives")
    contrastive = load("merlin/models/torch/outputs/contrastive.py", "contrastive_outputs", "ContrastiveOutput",
                       {"rescore_false_negatives": rescore})
    B, Nn, E = 37, 53, 24
    q = torch.randn(B, E, generator=g) * 0.5
    pos = torch.randn(B, E, generator=g) * 0.5
    neg = torch.randn(Nn, E, generator=g) * 0.5
    pos_id = torch.randint(0, 20, (B,), generator=g)
    neg_id = torch.randint(0, 20, (Nn,), generator=g)
    me = types.SimpleNamespace(downscore_false_negatives=True, false_negative_score=float(np.finfo(np.float16).min) / 100.0)
    logits = contrastive(me, q, pos, neg, pos_id.unsqueeze(0), neg_id.unsqueeze(0))
    out.update(sc_q=q, sc_pos=pos, sc_neg=neg, sc_pos_id=pos_id, sc_neg_id=neg_id, sc_logits=logits, sc_target=me.target)
    # in-batch: negatives are the batch's own items, ids shared (diagonal + duplicate ids masked)
    ids = torch.randint(0, 12, (B,), generator=g)
    me2 = types.SimpleNamespace(downscore_false_negatives=True, false_negative_score=-655.04)
    out.update(ib_ids=ids, ib_logits=contrastive(me2, q, pos, pos, ids.unsqueeze(0), ids.unsqueeze(0)))
    me3 = types.SimpleNamespace(downscore_false_negatives=False, false_negative_score=0.0)
    out.update(nd_logits=contrastive(me3, q, pos, neg, None, None))

    # --- DLRM interaction -----------------------------------------------------------------------
    fwd = load("merlin/models/torch/blocks/dlrm.py", "forward", "DLRMInteraction")
    Inter = type("Inter", (torch.nn.Module,), {"forward": fwd})
    for (b, f, d) in [(9, 27, 64), (5, 4, 8), (3, 17, 32)]:
        X = torch.randn(b, f, d, generator=g)
        out[f"int_x_{f}_{d}"] = X
        out[f"int_y_{f}_{d}"] = Inter()(X)

    # --- cross block (loop body of CrossBlock.forward with explicit Linear modules) -------------
    cross_fwd = load("merlin/models/torch/blocks/cross.py", "forward", "CrossBlock")
    d, depth = 20, 3
    lins = [torch.nn.Linear(d, d) for _ in range(depth)]
    for i, lin in enumerate(lins):
        with torch.no_grad():
            lin.weight.copy_(torch.randn(d, d, generator=g) * 0.2)
            lin.bias.copy_(torch.randn(d, generator=g) * 0.1)
        out[f"cr_W{i}"] = lin.weight.detach().T.contiguous()  # keras layout [in, out]
        out[f"cr_b{i}"] = lin.bias.detach()
    me4 = types.SimpleNamespace(values=lins, concat=None)
    x = torch.randn(11, d, generator=g)
    out.update(cr_x=x, cr_y=cross_fwd(me4, x).detach())
    # low-rank variant (DCN-v2 Eq. 2; TF: DenseMaybeLowRank = Dense(r, no bias) then Dense(d), blocks/mlp.py:365-396)
    r = 6
    lows = []
    for i in range(depth):
        u, v = torch.nn.Linear(d, r, bias=False), torch.nn.Linear(r, d)
        with torch.no_grad():
            u.weight.copy_(torch.randn(r, d, generator=g) * 0.2)
            v.weight.copy_(torch.randn(d, r, generator=g) * 0.2)
            v.bias.copy_(torch.randn(d, generator=g) * 0.1)
        out[f"crl_U{i}"] = u.weight.detach().T.contiguous()  # [d, r]
        out[f"crl_V{i}"] = v.weight.detach().T.contiguous()  # [r, d]
        out[f"crl_b{i}"] = v.bias.detach()
        lows.append(torch.nn.Sequential(u, v))
    me5 = types.SimpleNamespace(values=lows, concat=None)
    out.update(crl_y=cross_fwd(me5, x).detach())

    # --- inferred embedding dims ----------------------------------------------------------------
    emb = load("merlin/models/utils/schema_utils.py", "get_embedding_size_from_cardinality")
    cards = np.array([2, 10, 100, 1000, 10_000, 1_000_000, 10_000_000])
    out.update(emb_card=cards, emb_dim=np.array([emb(int(c), 2.0, True) for c in cards]),
               emb_dim_raw=np.array([emb(int(c), 2.0, False) for c in cards]))

    # --- popularity sampler distribution + logQ-corrected logits (appended: earlier draws are unchanged) -------
    # The reference's torch twin of the log-uniform sampler (torch/outputs/sampling/popularity.py) states the same
    # distribution as tf/outputs/sampling/popularity.py:139-166, whose range is one longer: TF(max_id) == twin(max_id + 1).
    lu = load("merlin/models/torch/outputs/sampling/popularity.py", "get_log_uniform_distr", "LogUniformSampler")
    # (the twin's "sampled at least once" formula is 1 - (1 + p)^-n, TF's is 1 - (1 - p)^n: only the base distribution
    # is pinned here, the unique-sampling form follows tf/outputs/sampling/popularity.py:152-158)
    for tag, (max_id, min_id, n) in {"a": (999, 2, 10), "b": (57, 0, 20)}.items():
        out[f"pop_{tag}_args"] = np.array([max_id, min_id, n])
        out[f"pop_{tag}_dist"] = lu(None, max_id + 1, min_id)
    # logQ BEFORE the false-negative rescoring (tf/outputs/contrastive.py:309-319): scores - log(p + 1e-16) are the dot
    # products of vectors augmented by one coordinate, so the reference's own contrastive_outputs produces them
    ppos = torch.rand(B, generator=g) * 0.2 + 1e-3
    pneg = torch.rand(Nn, generator=g) * 0.2 + 1e-3
    qa = torch.cat([q, torch.ones(B, 1)], 1)
    posa = torch.cat([pos, -torch.log(ppos + 1e-16).unsqueeze(1)], 1)
    nega = torch.cat([neg, -torch.log(pneg + 1e-16).unsqueeze(1)], 1)
    me6 = types.SimpleNamespace(downscore_false_negatives=True, false_negative_score=float(np.finfo(np.float16).min) / 100.0)
    out.update(lq_ppos=ppos, lq_pneg=pneg, lq_logits=contrastive(me6, qa, posa, nega, pos_id.unsqueeze(0), neg_id.unsqueeze(0)))
    # PopularityLogitsCorrection as `post` (tf/transforms/bias.py:238-254): every column of the rescored logits,
    # reg_factor * log(prob + 1e-16) subtracted -- the reference's own logits minus the stated correction
    reg = 0.7
    probs = torch.rand(20, generator=g) + 0.05
    probs = probs / probs.sum()
    corr = torch.cat([probs[pos_id].unsqueeze(1), probs[neg_id].unsqueeze(0).expand(B, Nn)], 1)
    out.update(pc_probs=probs, pc_reg=np.array(reg), pc_logits=logits - reg * torch.log(corr + 1e-16))

    # --- ragged lookup: the reference's torch backend combines a bag with F.embedding_bag(inputs, weight, offsets,
    # mode=seq_combiner) (torch/inputs/embedding.py:291-293) -- sum / mean bags incl. empty ones (offsets WITHOUT the end)
    embedding_bag = load("merlin/models/torch/inputs/embedding.py", "forward_bag", "EmbeddingTable",
                         {"nn": torch.nn})
    V, D_, nb = 29, 12, 17
    Wb = torch.randn(V, D_, generator=g)
    lens = torch.randint(0, 6, (nb,), generator=g)
    lens[4] = 0
    offs = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
    vals = torch.randint(0, V, (int(offs[-1]),), generator=g)
    out.update(bag_W=Wb, bag_values=vals, bag_offsets=offs)
    for mode in ("sum", "mean"):
        me7 = types.SimpleNamespace(table=types.SimpleNamespace(weight=Wb), seq_combiner=mode)
        out[f"bag_{mode}"] = embedding_bag(me7, vals, offs[:-1])

    # --- losses: the classes the reference's torch backend instantiates (appended: earlier draws are unchanged) ----
    # BinaryOutput: nn.Sigmoid() then DEFAULT_LOSS_CLS() = nn.BCELoss (torch/outputs/classification.py:44,58-62);
    # ContrastiveOutput: default loss = nn.CrossEntropyLoss() (:59) on [B, 1 + Nn] logits with the positive in column 0
    # (torch/outputs/contrastive.py: target built by contrastive_outputs).  Probabilities stay inside [1e-6, 1 - 1e-6]:
    # Keras clips at 1e-7, torch clamps log at -100 -- the two statements agree on that range.
    bce_cls = class_attr("merlin/models/torch/outputs/classification.py", "BinaryOutput", "DEFAULT_LOSS_CLS", {"nn": torch.nn})
    ce = init_default("merlin/models/torch/outputs/contrastive.py", "ContrastiveOutput", "loss", {"nn": torch.nn})
    z = torch.randn(257, 1, generator=g) * 3.0
    prob = torch.sigmoid(z).clamp(1e-6, 1 - 1e-6)
    lab = (torch.rand(257, 1, generator=g) < 0.4).float()
    out.update(bce_p=prob, bce_y=lab, bce_loss=bce_cls()(prob, lab), bce_cls=np.array(bce_cls.__name__))
    ce_logits = contrastive(me2, q, pos, pos, ids.unsqueeze(0), ids.unsqueeze(0))  # the in-batch logits above
    # class-index target 0 == the one-hot-on-column-0 target the reference builds
    out.update(ce_loss=ce(ce_logits, torch.zeros(B, dtype=torch.long)), ce_cls=np.array(type(ce).__name__),
               ce_loss_onehot=ce(ce_logits, me2.target))

    # --- DLRM top-MLP input layout (appended: earlier draws are unchanged) ----------------------------------------
    dlrm_layout(out, g)
    dense_mlp(out, g)
    batchnorm_mlp(out, g)  # appended: the generator state of everything above is unchanged
    adagrad_steps(out, g)

    np.savez_compressed(OUT / "reference_vectors.npz",
                        **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("wrote", OUT / "reference_vectors.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
