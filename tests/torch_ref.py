"""Plain PyTorch fp32 CPU reference (autograd) for the floating-point backward kernels.
Test infrastructure only."""
import numpy as np
import torch


def act(x, name):
    if name in (None, "linear"):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


def dot_interaction(X, tail=None):
    F = X.shape[1]
    Z = torch.bmm(X, X.transpose(1, 2))
    iu = torch.triu_indices(F, F, offset=1)
    out = Z[:, iu[0], iu[1]]
    return out if tail is None else torch.cat([tail, out], dim=1)  # [bottom | interactions]: the reference's order


def dlrm_forward(cat_ids, cont, tables, bottom, top, head):
    """Same math as oracle.dlrm_forward, in torch (differentiable)."""
    names = sorted(cat_ids)
    x = torch.cat([cont[k].reshape(len(cont[k]), -1) for k in sorted(cont)], dim=1)
    for W, b, a in bottom:
        x = act(x @ W + b, a)
    bottom_out = x
    feats = {n: tables[n][torch.as_tensor(cat_ids[n]).reshape(-1).long()] for n in names}
    feats["bottom_block"] = bottom_out
    stacked = torch.stack([feats[k] for k in sorted(feats)], dim=1)
    h = dot_interaction(stacked, bottom_out)
    for W, b, a in top:
        h = act(h @ W + b, a)
    p = torch.sigmoid(h @ head[0] + head[1])
    return p


def keras_bce(p, y):
    eps = 1e-7
    p = p.reshape(-1).clamp(eps, 1 - eps)
    y = y.reshape(-1)
    return -(y * torch.log(p) + (1 - y) * torch.log(1 - p)).mean()


def adagrad_update(w, g, acc, lr, eps=1e-7):
    acc = acc + g * g
    return w - lr * g / (acc.sqrt() + eps), acc
