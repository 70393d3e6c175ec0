"""torch-CPU (all host threads) statement of the DLRM step -- TEST INFRASTRUCTURE / ``bench.py`` CPU baseline only.

Same algorithm as ``oracle.dlrm_forward`` / ``oracle.dlrm_train_step`` (which follow
merlin/models/tf/blocks/dlrm.py:110-131, tf/blocks/interaction.py:86-116, tf/blocks/mlp.py:275-280,
tf/models/base.py:1121-1174), written with framework ops so that every stage runs on all host cores
(``torch.set_num_threads(os.cpu_count())``): this is the CPU baseline SURVEY.md section 8(d) asks for -- "oracle
(restated reference), not TensorFlow".  ``tests/test_oracle_torch.py`` pins it to the numpy oracle.  Nothing in
``models_amd/`` imports this module.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


def usable_cpus() -> int:
    """CPUs this process may actually run on: the affinity mask, clipped by the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001 -- no cgroup v2 file: keep the affinity count
        pass
    return n


def use_all_threads(probe=None) -> int:
    """Use every usable host thread -- unless ``probe`` (a callable running a short sample of the workload) shows that
    fewer threads are FASTER on this box (SMT siblings / container quotas can make 256 OpenMP threads crawl on 65 536
    tiny batched matmuls): the candidates are all, half and a quarter of the usable CPUs, the fastest one is kept and
    reported as ``cores``."""
    import time

    n = usable_cpus()
    if probe is None:
        torch.set_num_threads(n)
        return torch.get_num_threads()
    best, best_t = n, None
    for cand in sorted({n, max(1, n // 2), max(1, n // 4), min(n, 32)}, reverse=True):
        torch.set_num_threads(cand)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = cand, dt
    torch.set_num_threads(best)
    return torch.get_num_threads()


def _act(x, name):
    if name in (None, "linear"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


def _mlp(x, layers):
    for W, b, act in layers:
        x = _act(x @ W + b, act)
    return x


class DLRMState:
    """Host copies of the model state as torch tensors (tables, MLP layers, head, Adagrad accumulators)."""

    def __init__(self, tables: Dict[str, np.ndarray], bottom, top, head, initial_accumulator_value: float = 0.1):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).clone()
        self.names = sorted(tables)
        self.tables = {n: t(tables[n]) for n in self.names}
        self.bottom = [(t(W), t(b), a) for W, b, a in bottom]
        self.top = [(t(W), t(b), a) for W, b, a in top]
        self.head = (t(head[0]), t(head[1]))
        self.iav = initial_accumulator_value
        self.acc: Dict[object, torch.Tensor] = {}

    def dense_params(self) -> List[Tuple[object, torch.Tensor]]:
        out = []
        for i, (W, b, _) in enumerate(self.bottom):
            out += [(("bottom", i, "W"), W), (("bottom", i, "b"), b)]
        for i, (W, b, _) in enumerate(self.top):
            out += [(("top", i, "W"), W), (("top", i, "b"), b)]
        out += [(("head", "W"), self.head[0]), (("head", "b"), self.head[1])]
        return out


def _forward(state: DLRMState, cat: Dict[str, torch.Tensor], cont_x: torch.Tensor, rows: Optional[Dict[str, torch.Tensor]] = None):
    emb = rows if rows is not None else {n: state.tables[n][cat[n]] for n in state.names}
    bottom_out = _mlp(cont_x, state.bottom)
    feats = {**emb, "bottom_block": bottom_out}
    order = sorted(feats)  # StackFeatures: sorted keys (core/aggregation.py:101-108)
    X = torch.stack([feats[k] for k in order], dim=1)
    F = X.shape[1]
    iu = torch.triu_indices(F, F, offset=1)
    inter = torch.bmm(X, X.transpose(1, 2))[:, iu[0], iu[1]]  # strict upper triangle, row-major (interaction.py:107-112)
    top_in = torch.cat([bottom_out, inter], dim=1)            # [bottom | interactions] (see oracle.dlrm_interaction_concat)
    p = torch.sigmoid(_mlp(top_in, state.top) @ state.head[0] + state.head[1])
    return p


def _inputs(state, cat_ids, cont):
    cat = {n: torch.from_numpy(np.asarray(cat_ids[n]).reshape(-1).astype(np.int64)) for n in state.names}
    B = next(iter(cat.values())).shape[0]
    cont_x = torch.cat([torch.from_numpy(np.asarray(cont[k], dtype=np.float32).reshape(B, -1)) for k in sorted(cont)], dim=1)
    return cat, cont_x, B


def dlrm_forward(state: DLRMState, cat_ids, cont) -> np.ndarray:
    cat, cont_x, _ = _inputs(state, cat_ids, cont)
    with torch.no_grad():
        return _forward(state, cat, cont_x).numpy()


def dlrm_train_step(state: DLRMState, cat_ids, cont, labels, optimizer: str = "adagrad", lr: float = 0.01,
                    eps: float = 1e-7) -> float:
    """fwd + mean BCE (Keras clip 1e-7) + bwd + SGD / Adagrad; embedding rows as IndexedSlices: duplicate ids summed,
    only touched rows (and their accumulators) updated."""
    cat, cont_x, B = _inputs(state, cat_ids, cont)
    y = torch.from_numpy(np.asarray(labels, dtype=np.float32).reshape(B, 1))
    rows = {n: state.tables[n][cat[n]].requires_grad_() for n in state.names}
    dense = state.dense_params()
    for _, w in dense:
        w.requires_grad_()
    p = _forward(state, cat, cont_x, rows)
    pc = p.clamp(1e-7, 1 - 1e-7)
    loss = -(y * torch.log(pc) + (1 - y) * torch.log(1 - pc)).mean()
    grads = torch.autograd.grad(loss, [w for _, w in dense] + [rows[n] for n in state.names])
    with torch.no_grad():
        for (key, w), g in zip(dense, grads[: len(dense)]):
            w.requires_grad_(False)
            if optimizer == "adagrad":
                a = state.acc.setdefault(key, torch.full_like(w, state.iav))
                a.add_(g * g)
                w.sub_(lr * g / (a.sqrt() + eps))
            else:
                w.sub_(lr * g)
        for n, g in zip(state.names, grads[len(dense):]):
            uniq, inv = torch.unique(cat[n], return_inverse=True)
            gsum = torch.zeros((uniq.shape[0], g.shape[1]), dtype=torch.float32).index_add_(0, inv, g)
            if optimizer == "adagrad":
                a = state.acc.setdefault(("table", n), torch.full_like(state.tables[n], state.iav))
                ar = a[uniq] + gsum * gsum
                a[uniq] = ar
                state.tables[n][uniq] -= lr * gsum / (ar.sqrt() + eps)
            else:
                state.tables[n][uniq] -= lr * gsum
    return float(loss.detach())
