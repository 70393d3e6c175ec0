"""Schema-lite: the subset of ``merlin.schema`` (external to the reference, not installed here)
that the hot path consults -- column name, tags, integer domain, list-ness.

Mirrors the calls the reference makes: ``schema.select_by_tag(Tags.CATEGORICAL)``,
``.excluding_by_tag(Tags.TARGET)`` (tf/blocks/dlrm.py:90-91), ``col.int_domain.max``
(tf/inputs/embedding.py:91-93), ``categorical_cardinalities`` / ``infer_embedding_dim``
(utils/schema_utils.py:105-113, 198-207).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from enum import Enum
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Union


class Tags(str, Enum):
    CATEGORICAL = "categorical"
    CONTINUOUS = "continuous"
    LIST = "list"
    SEQUENCE = "sequence"
    TARGET = "target"
    BINARY_CLASSIFICATION = "binary_classification"
    REGRESSION = "regression"
    ID = "id"
    ITEM = "item"
    ITEM_ID = "item_id"
    USER = "user"
    USER_ID = "user_id"
    CONTEXT = "context"
    EMBEDDING = "embedding"


TagsType = Union[Tags, str, Sequence[Union[Tags, str]]]


@dataclass(frozen=True)
class Domain:
    min: Optional[float] = None
    max: Optional[float] = None
    name: Optional[str] = None


@dataclass(frozen=True)
class ColumnSchema:
    name: str
    tags: frozenset = field(default_factory=frozenset)
    dtype: str = "int64"
    is_list: bool = False
    is_ragged: bool = False
    int_domain: Optional[Domain] = None
    float_domain: Optional[Domain] = None
    value_count_max: Optional[int] = None

    def __post_init__(self):
        object.__setattr__(self, "tags", frozenset(_norm_tags(self.tags)))

    def with_tags(self, tags: TagsType) -> "ColumnSchema":
        return _replace(self, tags=self.tags | frozenset(_norm_tags(tags)))

    def with_name(self, name: str) -> "ColumnSchema":
        return _replace(self, name=name)


def _replace(col: ColumnSchema, **kw) -> ColumnSchema:
    d = {f: getattr(col, f) for f in col.__dataclass_fields__}
    d.update(kw)
    return ColumnSchema(**d)


def _norm_tags(tags: TagsType) -> List[Tags]:
    if isinstance(tags, (Tags, str)):
        tags = [tags]
    out = []
    for t in tags or []:
        out.append(t if isinstance(t, Tags) else Tags(str(t).lower()))
    return out


def categorical(name: str, cardinality: int, tags: TagsType = (), *, domain_name: Optional[str] = None,
                is_list: bool = False, is_ragged: bool = False, dtype: str = "int64") -> ColumnSchema:
    """Categorical column with ids in ``[0, cardinality)`` (``int_domain.max = cardinality - 1``)."""
    return ColumnSchema(name, frozenset(_norm_tags(tags)) | {Tags.CATEGORICAL}, dtype, is_list, is_ragged,
                        Domain(0, cardinality - 1, domain_name))


def continuous(name: str, tags: TagsType = ()) -> ColumnSchema:
    return ColumnSchema(name, frozenset(_norm_tags(tags)) | {Tags.CONTINUOUS}, "float32")


def binary_target(name: str) -> ColumnSchema:
    return ColumnSchema(name, frozenset({Tags.TARGET, Tags.BINARY_CLASSIFICATION}), "float32")


class Schema:
    """Ordered collection of :class:`ColumnSchema` keyed by name."""

    def __init__(self, columns: Iterable[Union[ColumnSchema, str]] = ()):
        self._cols: Dict[str, ColumnSchema] = {}
        for c in columns:
            if isinstance(c, str):
                c = ColumnSchema(c)
            if c.name in self._cols:
                raise ValueError(f"duplicate column {c.name!r}")
            self._cols[c.name] = c

    # --- selection (merlin.schema.Schema API used by the reference) ---
    def select_by_tag(self, tags: TagsType) -> "Schema":
        want = set(_norm_tags(tags))
        return Schema(c for c in self if c.tags & want)

    def excluding_by_tag(self, tags: TagsType) -> "Schema":
        drop = set(_norm_tags(tags))
        return Schema(c for c in self if not (c.tags & drop))

    remove_by_tag = excluding_by_tag

    def select_by_name(self, names: Union[str, Sequence[str]]) -> "Schema":
        if isinstance(names, str):
            names = [names]
        return Schema(self._cols[n] for n in names if n in self._cols)

    def excluding_by_name(self, names: Union[str, Sequence[str]]) -> "Schema":
        if isinstance(names, str):
            names = [names]
        return Schema(c for c in self if c.name not in set(names))

    @property
    def column_names(self) -> List[str]:
        return list(self._cols)

    @property
    def first(self) -> ColumnSchema:
        return next(iter(self._cols.values()))

    def get(self, name: str, default=None):
        return self._cols.get(name, default)

    def __getitem__(self, name: str) -> ColumnSchema:
        return self._cols[name]

    def __contains__(self, name: str) -> bool:
        return name in self._cols

    def __iter__(self) -> Iterator[ColumnSchema]:
        return iter(self._cols.values())

    def __len__(self) -> int:
        return len(self._cols)

    def __add__(self, other: "Schema") -> "Schema":
        cols = dict(self._cols)
        for c in other:
            cols[c.name] = c
        return Schema(cols.values())

    def __repr__(self) -> str:
        return f"Schema({self.column_names})"


# --- utils/schema_utils.py -----------------------------------------------------------------
def categorical_cardinalities(schema: Schema) -> Dict[str, int]:
    """utils/schema_utils.py:105-113: cardinality = int_domain.max + 1."""
    out = {}
    for col in schema.select_by_tag(Tags.CATEGORICAL):
        if col.int_domain is None or col.int_domain.max is None:
            raise ValueError(f"categorical column {col.name!r} needs an int_domain")
        out[col.name] = int(col.int_domain.max) + 1
    return out


def infer_embedding_dim(cardinality: int, multiplier: float = 2.0, ensure_multiple_of_8: bool = True) -> int:
    """utils/schema_utils.py:198-207: ceil(cardinality ** 0.25 * multiplier), rounded up to x8."""
    dim = int(math.ceil(math.pow(cardinality, 0.25) * multiplier))
    if ensure_multiple_of_8:
        dim = int(math.ceil(dim / 8) * 8)
    return dim
