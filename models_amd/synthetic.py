"""Synthetic inputs of the BASELINE.json configurations.

Cardinalities come from the reference's Criteo schema
(merlin/datasets/advertising/criteo/transformed/schema.pbtxt: int_domain.max + 1), capped at
1 000 000 rows as BASELINE.json config 2 states.
"""
from __future__ import annotations

import numpy as np

_CRITEO_MAX = [
    9_999_999, 29_427, 15_127, 7_295, 19_901, 3, 6_465, 1_310, 61, 9_999_999, 622_921, 219_556, 10,
    2_209, 9_779, 71, 4, 963, 14, 9_999_999, 4_384_510, 9_999_999, 290_588, 10_829, 95, 34,
]
CRITEO_CAT_NAMES = [f"C{i}" for i in range(1, 27)]
CRITEO_CONT_NAMES = [f"I{i}" for i in range(1, 14)]
CRITEO_CARDINALITIES = [min(m + 1, 1_000_000) for m in _CRITEO_MAX]


def lognormal_ids(rng: np.random.Generator, n: int, max_id: int) -> np.ndarray:
    """The reference generator's id recipe: clip(int(lognormal(3, 1)), 1, max)
    (merlin/datasets/synthetic.py:218-222, 244-248) -- heavily skewed towards small ids."""
    return np.clip(rng.lognormal(3.0, 1.0, size=n).astype(np.int64), 1, max_id)


def uniform_ids(rng: np.random.Generator, n: int, rows: int) -> np.ndarray:
    return rng.integers(0, rows, size=n, dtype=np.int64)
