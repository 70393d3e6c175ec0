"""The torch-CPU all-threads statement of the DLRM step (bench.py's cpu_baseline) against the numpy oracle."""
import copy

import numpy as np

from oracle import oracle as O
from oracle import oracle_torch as OT


def _setup(seed=0, B=257, D=16, F=5):
    rng = np.random.default_rng(seed)
    names = [f"C{i}" for i in range(1, F + 1)]
    cards = [7, 1000, 33, 4, 5000][:F]
    tables = {n: O.embedding_uniform(rng, v, D) for n, v in zip(names, cards)}
    cat = {n: rng.integers(0, v, size=(B, 1)) for n, v in zip(names, cards)}
    cont = {f"I{i}": rng.random((B, 1), dtype=np.float32) for i in range(1, 4)}
    lay = lambda dims: [(O.glorot_uniform(rng, a, b), np.zeros(b, np.float32), "relu") for a, b in zip(dims[:-1], dims[1:])]
    bottom = lay([3, 24, D])
    P = (F + 1) * F // 2
    top = lay([P + D, 32, 8])
    head = (O.glorot_uniform(rng, 8, 1), np.zeros(1, np.float32))
    y = rng.integers(0, 2, size=(B, 1)).astype(np.float32)
    return tables, cat, cont, bottom, top, head, y


def test_forward_matches_numpy_oracle():
    tables, cat, cont, bottom, top, head, _ = _setup()
    ref = O.dlrm_forward(cat, cont, tables, bottom, top, head)["prob"]
    st = OT.DLRMState(tables, bottom, top, head)
    np.testing.assert_allclose(OT.dlrm_forward(st, cat, cont), ref, atol=1e-6)


def test_train_steps_match_numpy_oracle():
    for opt in ("adagrad", "sgd"):
        tables, cat, cont, bottom, top, head, y = _setup(seed=3)
        st = OT.DLRMState(tables, bottom, top, head)
        t2, b2, p2, h2 = copy.deepcopy((tables, bottom, top, head))
        states = None
        for _ in range(3):
            l_np, states = O.dlrm_train_step(cat, cont, y, t2, b2, p2, h2, states, opt, 0.05)
            l_t = OT.dlrm_train_step(st, cat, cont, y, opt, 0.05)
            assert abs(l_np - l_t) < 1e-5
        for n in t2:
            np.testing.assert_allclose(st.tables[n].numpy(), t2[n], atol=2e-5)
        for (W, b, _), (Wt, bt, _) in zip(b2 + p2, st.bottom + st.top):
            np.testing.assert_allclose(Wt.detach().numpy(), W, atol=2e-5)
            np.testing.assert_allclose(bt.detach().numpy(), b, atol=2e-5)
        np.testing.assert_allclose(st.head[0].detach().numpy(), h2[0], atol=2e-5)
