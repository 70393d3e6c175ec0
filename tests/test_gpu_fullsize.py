"""BASELINE.json full-size configurations: size-independent properties + oracle checks on a row subset."""
import numpy as np
import pytest
import torch

import models_amd as mm
from models_amd import ops, schema as S
from models_amd.synthetic import CRITEO_CARDINALITIES, CRITEO_CAT_NAMES, CRITEO_CONT_NAMES
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _c2_model(device):
    cols = [S.categorical(n, v) for n, v in zip(CRITEO_CAT_NAMES, CRITEO_CARDINALITIES)]
    cols += [S.continuous(n) for n in CRITEO_CONT_NAMES] + [S.binary_target("label")]
    schema = mm.Schema(cols)
    return mm.DLRMModel(schema, embedding_dim=64, bottom_block=mm.MLPBlock([128, 64], device=device),
                        top_block=mm.MLPBlock([128, 64, 32], device=device), device=device)


def test_c2_dlrm_forward_full_batch(device):
    """configs[1] at B = 65536: (a) rows are independent -> a permuted batch gives the permuted output,
    bit for bit; (b) the first 512 rows match the numpy oracle within 1e-4; (c) output in (0, 1)."""
    model = _c2_model(device)
    g = torch.Generator().manual_seed(0)
    B = 65536
    batch = {n: torch.randint(0, v, (B, 1), generator=g, dtype=torch.int32).to(device) for n, v in zip(CRITEO_CAT_NAMES, CRITEO_CARDINALITIES)}
    batch.update({n: torch.rand(B, 1, generator=g).to(device) for n in CRITEO_CONT_NAMES})
    p = model(batch)
    assert p.shape == (B, 1) and bool(((p > 0) & (p < 1)).all())
    perm = torch.randperm(B, generator=g).to(device)
    p2 = model({k: v[perm] for k, v in batch.items()})
    assert torch.equal(p2, p[perm])
    n = 512
    body = model.body
    lay = lambda blk: [(l.kernel.numpy(), l.bias.numpy(), l.activation) for l in blk.layers]
    hd = model.output.to_call
    sub_ids = {k: batch[k][:n].cpu().numpy() for k in CRITEO_CAT_NAMES}
    tables = {k: body.embeddings.feature_table[k].table.data[torch.unique(batch[k][:n].long())] for k in CRITEO_CAT_NAMES}
    # compact tables for the oracle: remap ids to the unique set (avoids copying 1.6 GB to the host)
    remap = {}
    for k in CRITEO_CAT_NAMES:
        u, inv = np.unique(sub_ids[k].reshape(-1), return_inverse=True)
        remap[k] = inv.reshape(-1, 1)
    ref = O.dlrm_forward(remap, {k: batch[k][:n].cpu().numpy() for k in CRITEO_CONT_NAMES},
                         {k: v.cpu().numpy() for k, v in tables.items()}, lay(body.bottom_block), lay(body.top_block),
                         (hd.kernel.numpy(), hd.bias.numpy()))
    np.testing.assert_allclose(p[:n].cpu().numpy(), ref["prob"], atol=1e-4)


def test_c3_scorer_full_batch_properties(device):
    """configs[2] at B = 32768, E = 128: fused loss == loss recomputed from the materialised logits,
    column 0 == row-wise dot, diagonal == false-negative score, lse >= max logit."""
    g = torch.Generator().manual_seed(1)
    B, E = 32768, 128
    q = (torch.randn(B, E, generator=g) * 0.1).to(device)
    it = (torch.randn(B, E, generator=g) * 0.1).to(device)
    ids = torch.randperm(1_000_000, generator=g)[:B].to(torch.int32).to(device)
    fused = ops.inbatch_softmax(q, it, it, ids, ids, materialize=False)
    full = ops.inbatch_softmax(q, it, it, ids, ids, materialize=True)
    assert full.logits.shape == (B, B + 1)
    torch.testing.assert_close(fused.loss, full.loss, atol=0, rtol=0)
    torch.testing.assert_close(full.logits[:, 0], (q * it).sum(-1), atol=1e-5, rtol=1e-5)
    diag = full.logits[:, 1:].diagonal()
    assert bool((diag == torch.tensor(O.MIN_FLOAT, dtype=torch.float32, device=device)).all())
    ref_lse = torch.logsumexp(full.logits[:1024].double(), dim=1).float()
    torch.testing.assert_close(full.lse[:1024], ref_lse, atol=1e-4, rtol=1e-5)
    assert bool((full.lse >= full.logits.max(dim=1).values - 1e-6).all())
