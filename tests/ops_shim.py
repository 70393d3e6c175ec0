"""Framework-op (torch, any device) statements of the ``models_amd.ops`` entry points the DLRM step uses.

TEST INFRASTRUCTURE ONLY: ``install()`` monkeypatches them over the HIP wrappers so that the HOST logic of
``models_amd`` (block wiring, gradient scaling, the sharded multi-GPU step, bucket layout) can be exercised on CPU
with ``gloo`` at world size > 1, where no HIP device exists.  The numerics of the kernels themselves are checked
by the ``-m gpu`` tests against ``oracle/``; nothing here is reachable from the product."""
from __future__ import annotations

import torch

from models_amd import distributed as D


def _act(x, name):
    if name in (None, "linear"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


def _act_grad(y, g, name):
    if name in (None, "linear"):
        return g
    if name == "relu":
        return g * (y > 0).to(g.dtype)
    if name == "sigmoid":
        return g * y * (1 - y)
    raise ValueError(name)


def embedding_gather(tables, ids, out=None, out_slot=None, n_slots=None, out_offset=None):
    F, D = len(tables), tables[0].shape[1]
    B = ids[0].reshape(-1).shape[0]
    slots = list(range(F)) if out_slot is None else list(out_slot)
    if n_slots is None:
        n_slots = (max(slots) + 1) if out is None else out.shape[1]
    if out is None:
        out = torch.empty((B, n_slots, D), dtype=torch.float32, device=tables[0].device)
    if B == 0:  # a rank that owns none of the requested rows (all ids of a skewed batch on one owner)
        return out
    flat = out.view(B, -1)
    offs = [s * D for s in slots] if out_offset is None else list(out_offset)
    for t, i, o in zip(tables, ids, offs):
        idx = i.reshape(-1).long()
        ok = (idx >= 0) & (idx < t.shape[0])  # ids outside the table read as a zero row (mh_embedding_gather_fwd)
        flat[:, o:o + D] = t[idx.clamp(0, max(t.shape[0] - 1, 0))] * ok.unsqueeze(1).to(t.dtype)
    return out


def linear(x, W, b=None, activation=None, out=None):
    y = x @ W
    if b is not None:
        y = y + b
    y = _act(y, activation)
    if out is None:
        return y
    out.copy_(y)
    return out


def dot_interaction(x, tail=None, out=None, tail_first=True):
    B, F, _ = x.shape
    iu = torch.triu_indices(F, F, offset=1)
    inter = torch.bmm(x, x.transpose(1, 2))[:, iu[0], iu[1]]
    res = inter if tail is None else torch.cat([tail, inter] if tail_first else [inter, tail], dim=1)
    if out is None:
        return res
    out.copy_(res)
    return out


def dot_interaction_backward(x, dout, tail_slot=-1, tail_width=0, tail_first=True):
    B, F, D = x.shape
    P = F * (F - 1) // 2
    T = tail_width if tail_slot >= 0 else 0
    pofs, tofs = (T, 0) if tail_first else (0, P)
    iu = torch.triu_indices(F, F, offset=1)
    G = torch.zeros(B, F, F, dtype=x.dtype, device=x.device)
    G[:, iu[0], iu[1]] = dout[:, pofs:pofs + P]
    dx = torch.bmm(G + G.transpose(1, 2), x)
    if T > 0:
        dx[:, tail_slot, :T] += dout[:, tofs:tofs + T]
    return dx


def linear_backward(x, W, y, dy, activation=None, need_dx=True, need_db=True, x_activation=None, zero_pad=True, late_dw=None):
    if activation not in (None, "linear"):
        dy.copy_(_act_grad(y, dy, activation))  # in place, like the kernel
    dx = None
    if need_dx:
        dx = _act_grad(x, dy @ W.t(), x_activation)
    return dx, x.t() @ dy, (dy.sum(0) if need_db else None)


def embedding_gather_backward(tables, states, ids, grad, grad_offset, optimizer="sgd", lr=0.01, eps=1e-7,
                              states2=None, beta1=0.9, beta2=0.999, lr_device=None, prepared=None):
    if optimizer not in ("sgd", "adagrad"):
        raise NotImplementedError("shim: sgd / adagrad only")
    B = grad.shape[0]
    if B == 0:  # nothing arrived for this rank's shards (mh_embedding_gather_bwd returns at once for B <= 0)
        return
    g2 = grad.reshape(B, -1)
    D = tables[0].shape[1]
    seen = {}
    for f, t in enumerate(tables):
        key = t.data_ptr()
        if key not in seen:
            seen[key] = (t, None if states is None else states[f], torch.zeros_like(t))
        idx = ids[f].reshape(-1).long()
        ok = (idx >= 0) & (idx < t.shape[0])
        seen[key][2].index_add_(0, idx[ok], g2[:, grad_offset[f]:grad_offset[f] + D][ok])
    for t, st, g in seen.values():
        if optimizer == "adagrad":
            st += g * g
            t -= lr * g / (st.sqrt() + eps)
        else:
            t -= lr * g


def _bag_ids(values, offsets, B):
    lens = (offsets[1:] - offsets[:-1]).long()
    return torch.repeat_interleave(torch.arange(B, device=values.device), lens)


def _bag_div(combiner, n):
    n = n.to(torch.float32)
    one = torch.ones_like(n)
    if combiner == "mean":
        return torch.where(n > 0, n, one)
    if combiner == "sqrtn":
        return torch.where(n > 0, n.sqrt(), one)
    return one


def embedding_bag(table, values, offsets, combiner="mean", out=None):
    B, V = offsets.shape[0] - 1, table.shape[0]
    values = values.reshape(-1).long()
    bag = _bag_ids(values, offsets.reshape(-1), B)
    kept = values >= 0
    valid = kept & (values < V)
    acc = torch.zeros((B, table.shape[1]), dtype=torch.float32, device=table.device)
    acc.index_add_(0, bag[valid], table[values[valid]])
    acc = acc / _bag_div(combiner, torch.bincount(bag[kept], minlength=B)).unsqueeze(1)
    if out is None:
        return acc
    out.copy_(acc)
    return out


def embedding_dense_list(table, ids, combiner="mean", out=None):
    if ids.dim() == 3:
        ids = ids.squeeze(-1)
    B, L = ids.shape
    idx = ids.reshape(-1).long()
    ok = (idx >= 0) & (idx < table.shape[0])
    g = (table[idx.clamp(0, table.shape[0] - 1)] * ok.unsqueeze(1).to(table.dtype)).reshape(B, L, -1)
    res = {"mean": lambda: g.mean(1), "sum": lambda: g.sum(1), "max": lambda: g.max(1).values}[combiner]()
    if out is None:
        return res
    out.copy_(res)
    return out


def embedding_bag_expand(table, values, offsets, grad, combiner="mean"):
    B = grad.shape[0]
    if offsets is None:
        if values.dim() == 3:
            values = values.squeeze(-1)
        L = values.shape[1]
        flat = values.reshape(-1)
        bag = torch.arange(B, device=grad.device).repeat_interleave(L)
        if combiner == "max":
            idx = flat.long()
            ok = (idx >= 0) & (idx < table.shape[0])
            rows = (table[idx.clamp(0, table.shape[0] - 1)] * ok.unsqueeze(1).to(table.dtype)).reshape(B, L, -1)
            sel = (rows == rows.max(1, keepdim=True).values).to(grad.dtype)
            return flat, (sel / sel.sum(1, keepdim=True) * grad.unsqueeze(1)).reshape(B * L, -1)
        div = torch.full((B,), float(L) if combiner == "mean" else 1.0, device=grad.device)
    else:
        flat = values.reshape(-1)
        bag = _bag_ids(flat, offsets.reshape(-1), B)
        div = _bag_div(combiner, torch.bincount(bag[flat >= 0], minlength=B))
    return flat, grad[bag] / div[bag].unsqueeze(1)


def embedding_bag_backward(table, state, values, offsets, grad, combiner="mean", optimizer="sgd", lr=0.01, eps=1e-7,
                           state2=None, beta1=0.9, beta2=0.999, lr_device=None):
    flat, gexp = embedding_bag_expand(table, values, offsets, grad, combiner)
    embedding_gather_backward([table], None if state is None else [state], [flat], gexp.unsqueeze(1), [0], optimizer, lr, eps)


def bce(p, label, need_grad=True):
    p, label = p.reshape(-1), label.reshape(-1)
    pc = p.clamp(1e-7, 1 - 1e-7)
    loss = -(label * pc.log() + (1 - label) * (1 - pc).log())
    dlogit = ((p - label) / p.shape[0]).reshape(-1, 1) if need_grad else None
    return loss.mean(), dlogit


def dense_optimizer_step_multi(opt, params):
    for p in params:
        if p.grad is None:
            continue
        g = p.grad.reshape(p.data.shape)
        if opt.name == "adagrad":
            if "accumulator" not in p.state:
                p.state["accumulator"] = torch.full_like(p.data, opt.initial_accumulator_value)
            p.state["accumulator"] += g * g
            p.data -= opt.learning_rate * g / (p.state["accumulator"].sqrt() + opt.epsilon)
        elif opt.name == "sgd":
            p.data -= opt.learning_rate * g
        elif opt.name == "adam":  # keras Adam on a dense gradient (mh_dense_optimizer_step with the bias-corrected lr of adam_tick)
            if "m" not in p.state:
                p.state["m"], p.state["v"] = torch.zeros_like(p.data), torch.zeros_like(p.data)
            p.state["m"].mul_(opt.beta_1).add_(g, alpha=1 - opt.beta_1)
            p.state["v"].mul_(opt.beta_2).addcmul_(g, g, value=1 - opt.beta_2)
            p.data -= float(opt.lr_device) * p.state["m"] / (p.state["v"].sqrt() + opt.epsilon)
        else:
            raise NotImplementedError("shim: sgd / adagrad / adam")
        p.grad = None


def adam_tick(opt):
    """mh_adam_tick: step += 1, lr_device = lr sqrt(1 - b2^t) / (1 - b1^t)"""
    opt._step_dev += 1.0
    t = float(opt._step_dev)
    opt.lr_device.fill_(opt.learning_rate * (1 - opt.beta_2 ** t) ** 0.5 / (1 - opt.beta_1 ** t))


def route_build(ids, world_size, slots=None, n_slots=None, capacity=0, overflow=None, dedup=False):
    F = len(ids)
    slots = list(range(F)) if slots is None else list(slots)
    return D.route_build_torch(ids, world_size, slots, (max(slots) + 1) if n_slots is None else n_slots, capacity, overflow,
                               dedup=dedup)


def eltwise(op, a, b, c=None):
    return {"mul": lambda: a * b, "add": lambda: a + b, "fma": lambda: a * b + c}[op]()


def rowwise_dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def cross_layer(x0, x, W, b, save_p=False):
    p = x @ W + (0 if b is None else b)
    return (x0 * p + x, p) if save_p else x0 * p + x


def cross_layer_backward(x0, x, p, dout, W, dx0_acc=None):
    g = dout * x0
    share = dout * p
    dx0_acc = share if dx0_acc is None else dx0_acc.add_(share)
    return dx0_acc, g @ W.t() + dout, x.t() @ g, g.sum(0)


def cross_lowrank_dx(dh, U, dout):
    return dh @ U.t() + dout


def _scorer_terms(q, item, neg, pos_ids, neg_ids, T, fns, pos_logq, neg_logq, after):
    pos = (q * item).sum(-1, keepdim=True)
    ng = q @ neg.t()
    if pos_logq is not None:
        pos = pos - pos_logq.reshape(-1, 1)
        if not after:
            ng = ng - neg_logq.reshape(1, -1)
    if pos_ids is not None:
        ng = torch.where(pos_ids.reshape(-1, 1) == neg_ids.reshape(1, -1), torch.full_like(ng, fns), ng)
    if pos_logq is not None and after:
        ng = ng - neg_logq.reshape(1, -1)
    return torch.cat([pos, ng], 1) / T


def inbatch_softmax(q, item, neg_item, pos_ids=None, neg_ids=None, temperature=1.0, false_neg_score=-655.04,
                    materialize=True, pos_logq=None, neg_logq=None, logq_after_mask=False):
    from models_amd.ops import ScorerResult

    z = _scorer_terms(q, item, neg_item, pos_ids, neg_ids, temperature, false_neg_score, pos_logq, neg_logq, logq_after_mask)
    lse = torch.logsumexp(z, 1)
    return ScorerResult(z if materialize else None, lse - z[:, 0], lse)


def _scorer_grads(q, item, neg_item, pos_ids, neg_ids, T, fns, grad_scale, pos_logq, neg_logq, after):
    qq, ii, nn = (t.detach().clone().requires_grad_() for t in (q, item, neg_item))
    z = _scorer_terms(qq, ii, nn, pos_ids, neg_ids, T, fns, pos_logq, neg_logq, after)
    per = torch.logsumexp(z, 1) - z[:, 0]
    (per.sum() * (1.0 / q.shape[0] if grad_scale is None else grad_scale)).backward()
    return per.detach(), torch.logsumexp(z, 1).detach(), qq.grad, ii.grad, nn.grad


def inbatch_softmax_train(q, item, neg_item, pos_ids=None, neg_ids=None, temperature=1.0, false_neg_score=-655.04,
                          grad_scale=None, pos_logq=None, neg_logq=None, logq_after_mask=False):
    from models_amd.ops import ScorerResult

    # neg_item aliases item for in-batch negatives: differentiate the two roles separately, like the kernels
    loss, lse, dq, ditem, _ = _scorer_grads(q, item, neg_item.detach().clone(), pos_ids, neg_ids, temperature,
                                            false_neg_score, grad_scale, pos_logq, neg_logq, logq_after_mask)
    return ScorerResult(None, loss, lse), dq, ditem


def inbatch_softmax_backward(q, item, neg_item, lse, pos_ids=None, neg_ids=None, temperature=1.0,
                             false_neg_score=-655.04, grad_scale=None, need_dq=True, pos_logq=None, neg_logq=None,
                             logq_after_mask=False):
    _, _, dq, ditem, dneg = _scorer_grads(q, item.detach().clone(), neg_item.detach().clone(), pos_ids, neg_ids,
                                          temperature, false_neg_score, grad_scale, pos_logq, neg_logq, logq_after_mask)
    return (dq, ditem, dneg) if need_dq else (None, None, dneg)


def l2norm(x, eps=1e-6):
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def zero_pad_columns(buf, width):
    if buf.shape[1] != width:
        buf[:, width:].zero_()


def install():
    """Replace the HIP wrappers of ``models_amd.ops`` by the statements above (this process only)."""
    from models_amd import ops

    for name in ("embedding_gather", "linear", "dot_interaction", "dot_interaction_backward", "linear_backward",
                 "embedding_gather_backward", "bce", "dense_optimizer_step_multi", "adam_tick", "route_build", "eltwise", "rowwise_dot",
                 "cross_layer", "cross_layer_backward", "cross_lowrank_dx", "inbatch_softmax", "inbatch_softmax_train", "inbatch_softmax_backward", "l2norm", "embedding_bag",
                 "embedding_dense_list", "embedding_bag_expand", "embedding_bag_backward", "zero_pad_columns"):
        setattr(ops, name, globals()[name])
    ops.route_local_rows = D.route_local_rows_torch
