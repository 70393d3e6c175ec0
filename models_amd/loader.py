"""Batches for the hot path from a Parquet dataset / DataFrame / dict of arrays.

The reference feeds its models through ``merlin.models.tf.loader.Loader`` (tf/loader.py:247-333, a thin wrapper of
the external ``merlin.dataloader``): per batch a dict ``name -> tensor`` in the ``PrepareFeatures`` input contract
(tf/transforms/features.py:143-379) plus the target column(s):

  * scalar categorical column  -> ``[B]`` int32/int64 ids (``prepare_features`` makes them ``[B, 1]``);
  * scalar continuous column   -> ``[B, 1]`` float32;
  * list column (ragged or fixed length) -> CSR ``name__values [nnz]`` + ``name__offsets [B + 1]``;
  * target column(s)           -> ``[B, 1]`` float32 (a dict of them if there are several);
  * ranks of a multi-GPU job read contiguous, equally sized slices of the row range (loader.py:308-311).

This is IO plumbing, not a kernel: pyarrow decodes the columns once into host arrays; every batch is staged through
pinned host memory and copied on a dedicated HIP stream, one batch ahead of the consumer, so the copy of batch i + 1
overlaps the train step of batch i (`MI355X: overlap copies with compute on separate streams`).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

from .schema import Schema, Tags


def _table_to_columns(table, names: List[str]) -> Dict[str, Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]]:
    """pyarrow Table -> name -> numpy array (scalar column) or (values, offsets) (list column)."""
    import pyarrow as pa

    out = {}
    for n in names:
        col = table.column(n).combine_chunks()
        if pa.types.is_list(col.type) or pa.types.is_large_list(col.type) or pa.types.is_fixed_size_list(col.type):
            if pa.types.is_fixed_size_list(col.type):
                L = col.type.list_size
                values = col.flatten().to_numpy(zero_copy_only=False)
                offsets = np.arange(len(col) + 1, dtype=np.int64) * L
            else:
                if col.null_count:
                    raise ValueError(f"list column {n!r} has null rows; fill them with empty lists first")
                offsets = col.offsets.to_numpy(zero_copy_only=False).astype(np.int64)
                values = col.values.to_numpy(zero_copy_only=False)[offsets[0]:offsets[-1]]
                offsets = offsets - offsets[0]
            out[n] = (values, offsets)
        else:
            if col.null_count:
                raise ValueError(f"column {n!r} has nulls; the hot path expects dense, preprocessed features")
            out[n] = col.to_numpy(zero_copy_only=False)
    return out


def _column_arrays(data, names: List[str]) -> Dict[str, Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]]:
    """Whole dataset (Parquet path / DataFrame / dict of arrays) -> columns."""
    try:
        import pyarrow as pa
        import pyarrow.parquet as pq
    except ImportError as e:  # pragma: no cover
        raise ImportError("models_amd.loader needs pyarrow to read Parquet / DataFrames") from e
    if isinstance(data, (str, Path)):
        p = Path(data)
        files = sorted(p.glob("*.parquet")) if p.is_dir() else [p]
        if not files:
            raise FileNotFoundError(f"no parquet files under {p}")
        table = pa.concat_tables([pq.read_table(f, columns=names) for f in files])
    elif isinstance(data, dict):
        out = {}
        for n in names:
            v = data[n]
            out[n] = (np.asarray(v[0]), np.asarray(v[1])) if isinstance(v, tuple) else np.asarray(v)
        return out
    else:  # pandas DataFrame (or anything pyarrow can convert)
        table = pa.Table.from_pandas(data[names], preserve_index=False) if hasattr(data, "columns") else pa.table(data)
    return _table_to_columns(table, names)


Columns = Dict[str, Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]]


def _n_rows(cols: Columns) -> int:
    v = next(iter(cols.values()))
    return len(v[1]) - 1 if isinstance(v, tuple) else len(v)


def _slice_rows(cols: Columns, a: int, b: int) -> Columns:
    out: Columns = {}
    for n, v in cols.items():
        if isinstance(v, tuple):
            vals, offs = v
            o = offs[a:b + 1]
            out[n] = (vals[o[0]:o[-1]], o - o[0])
        else:
            out[n] = v[a:b]
    return out


def _take_rows(cols: Columns, idx: np.ndarray) -> Columns:
    out: Columns = {}
    for n, v in cols.items():
        if isinstance(v, tuple):
            vals, offs = v
            lens = (offs[idx + 1] - offs[idx]).astype(np.int64)
            new_offs = np.zeros(len(idx) + 1, dtype=offs.dtype)
            np.cumsum(lens, out=new_offs[1:])
            starts = np.repeat(offs[idx].astype(np.int64) - new_offs[:-1].astype(np.int64), lens)  # value ranges of the rows
            out[n] = (vals[starts + np.arange(int(new_offs[-1]), dtype=np.int64)], new_offs)
        else:
            out[n] = v[idx]
    return out


def _concat_rows(parts: List[Columns]) -> Columns:
    if len(parts) == 1:
        return parts[0]
    out: Columns = {}
    for n in parts[0]:
        if isinstance(parts[0][n], tuple):
            vals = np.concatenate([p[n][0] for p in parts])
            offs, base = [np.zeros(1, dtype=parts[0][n][1].dtype)], 0
            for p in parts:
                o = p[n][1]
                offs.append(o[1:] + base)
                base += int(o[-1])
            out[n] = (vals, np.concatenate(offs).astype(parts[0][n][1].dtype))
        else:
            out[n] = np.concatenate([p[n] for p in parts])
    return out


class Loader:
    """``for inputs, targets in Loader(path_or_df, schema, batch_size): model.train_step(inputs, targets)``.

    ``buffer_rows`` (Parquet paths only) switches to STREAMING: instead of decoding the dataset into host memory
    once, row groups are read on the fly and batches are cut from chunks of at least ``buffer_rows`` rows (shuffled
    inside a chunk, row groups visited in a shuffled order -- the ``buffer_size`` / ``parts_per_chunk`` scheme of
    the reference loader, tf/loader.py:247-333).  Ranks take every W-th row group and all stop after the same number
    of rows, so that collective train steps stay aligned."""

    def __init__(self, paths_or_dataset, schema: Schema, batch_size: int, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False, device=None, global_rank: Optional[int] = None,
                 global_size: Optional[int] = None, prefetch: bool = True, buffer_rows: Optional[int] = None,
                 device_chunk_rows: Optional[int] = 8_388_608, device_resident_bytes: int = 48 << 30):
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        self.schema, self.batch_size, self.shuffle, self.seed, self.drop_last = schema, int(batch_size), shuffle, seed, drop_last
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.prefetch = prefetch and self.device.type == "cuda"
        if global_size is None:
            import torch.distributed as dist

            ready = dist.is_available() and dist.is_initialized()
            global_rank, global_size = (dist.get_rank(), dist.get_world_size()) if ready else (0, 1)
        self.rank, self.world = int(global_rank or 0), int(global_size)
        self.label_names = [c.name for c in schema.select_by_tag(Tags.TARGET)]
        self.cat_names = [c.name for c in schema.select_by_tag(Tags.CATEGORICAL).excluding_by_tag(Tags.TARGET)]
        self.cont_names = [c.name for c in schema.select_by_tag(Tags.CONTINUOUS).excluding_by_tag(Tags.TARGET)]
        names = self.cat_names + self.cont_names + self.label_names
        if not names:
            raise ValueError("the schema selects no columns")
        self._names = names
        self._epoch = 0
        self._stream: Optional[List[Tuple[str, int, int]]] = None  # (file, row group, rows) of this rank
        if buffer_rows is not None:
            self._init_streaming(paths_or_dataset, int(buffer_rows))
            return
        cols = _column_arrays(paths_or_dataset, names)
        rows = {n: (len(v[1]) - 1 if isinstance(v, tuple) else len(v)) for n, v in cols.items()}
        if len(set(rows.values())) != 1:
            raise ValueError(f"columns have different lengths: {rows}")
        total = next(iter(rows.values()))
        per = total // self.world  # equal contiguous slices; the remainder rows are dropped so that ranks stay in step
        self.lo, self.n_rows = self.rank * per, per
        self.columns = self._typed(cols)
        # DEVICE-CHUNK mode (GPU, dataset decoded into host memory): the columns of this rank's slice live in PINNED host memory;
        # an epoch copies them to the GPU in chunks of `device_chunk_rows` rows (whole columns, one async copy each, on a copy
        # stream, one chunk ahead of the consumer), shuffles a chunk ON THE DEVICE (one permutation shared by all columns: the
        # reference loader's shuffle-inside-a-buffer scheme, tf/loader.py:247-333), and hands out batches as VIEWS of the chunk --
        # no per-batch host gather, no per-batch pinned allocation, no per-batch copy.  The per-batch host path below (one fancy
        # index + pin + copy per column) feeds ~2 M samples/s; a DLRM step consumes 68 M samples/s (160 B per sample).
        self._pinned = None
        self._dev_cache = None
        if device_chunk_rows and self.device.type == "cuda":
            self.device_chunk_rows = max(int(device_chunk_rows) // self.batch_size, 1) * self.batch_size
            self._pinned = {}
            for n, v in self.columns.items():
                if isinstance(v, tuple):
                    vals, offs = v
                    o = offs[self.lo:self.lo + self.n_rows + 1]
                    self._pinned[n] = (self._pin(vals[int(o[0]):int(o[-1])]), (o - o[0]).astype(np.int64))  # offsets stay on the host
                else:
                    self._pinned[n] = self._pin(v[self.lo:self.lo + self.n_rows])
            # DEVICE-RESIDENT: a rank's slice that fits `device_resident_bytes` (default 48 GB of the 288 GB of HBM) is uploaded
            # ONCE -- the chunks of the first epoch stay on the device, later epochs only shuffle there.  The host link feeds
            # ~5 GB/s of pinned columns on the boxes this was measured on; a DLRM step consumes 10.9 GB/s (68 M samples/s x 160 B).
            nbytes = sum((v[0].numel() * v[0].element_size() + 8 * len(v[1])) if isinstance(v, tuple) else v.numel() * v.element_size()
                         for v in self._pinned.values())
            self.dataset_bytes = int(nbytes)
            # the pinned copy IS the dataset from here on: the decoded arrays are released (they doubled the host footprint)
            self.columns = {}
            # never more than half of what is FREE on this GPU right now (round-5 advisor finding: 48 GB assumed a 288 GB device
            # with nothing else on it; beside large embedding tables the cache would have run the device out of memory mid-epoch)
            budget = int(device_resident_bytes or 0)
            if budget:
                try:
                    budget = min(budget, torch.cuda.mem_get_info(self.device)[0] // 2)
                except Exception:  # noqa: BLE001 -- no usable figure: stream the chunks
                    budget = 0
            self.device_resident = bool(budget and nbytes <= budget)
            if self.device_resident:
                self._dev_cache = {}

    def _init_streaming(self, path, buffer_rows: int) -> None:
        import pyarrow.parquet as pq

        if not isinstance(path, (str, Path)):
            raise TypeError("buffer_rows (streaming) needs a Parquet file or directory")
        p = Path(path)
        files = sorted(p.glob("*.parquet")) if p.is_dir() else [p]
        if not files:
            raise FileNotFoundError(f"no parquet files under {p}")
        groups = []
        for f in files:
            md = pq.ParquetFile(f).metadata
            groups += [(str(f), g, md.row_group(g).num_rows) for g in range(md.num_row_groups)]
        per_rank = [groups[r::self.world] for r in range(self.world)]
        totals = [sum(g[2] for g in gs) for gs in per_rank]
        if min(totals) == 0:
            raise ValueError(f"{len(groups)} row group(s) cannot feed {self.world} ranks: rewrite the dataset with more "
                             "row groups or load it without buffer_rows")
        self._stream = per_rank[self.rank]
        self.n_rows = min(totals)  # every rank stops after the same number of rows
        self.lo = 0
        self.buffer_rows = max(buffer_rows, self.batch_size)
        self.columns = {}

    @staticmethod
    def _pin(a: np.ndarray) -> torch.Tensor:
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        return torch.from_numpy(a).pin_memory()

    def _device_chunk_batches(self, epoch: int) -> Iterator:
        """Batches as views of device-resident, device-shuffled chunks (see __init__)."""
        dev, B, C, n = self.device, self.batch_size, self.device_chunk_rows, self.n_rows

        def plan(ep):
            st = list(range(0, n, C))
            r = np.random.default_rng(self.seed + ep)
            if self.shuffle and len(st) > 1:
                # the short last chunk STAYS last: in the middle of an epoch its partial batch would run eagerly inside fit
                # (a captured step has one static shape) and every later batch of the epoch would sit at a shifted offset
                tail = [st.pop()] if n % C else []
                st = [st[i] for i in r.permutation(len(st))] + tail
            return st, r

        starts, rng = plan(epoch)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        copy_stream = self._copy_stream
        cur_stream = torch.cuda.current_stream(dev)
        lists = [k for k, v in self._pinned.items() if isinstance(v, tuple)]

        cache = self._dev_cache
        # STREAMING (the dataset does not stay on the device): the chunks land in two PERSISTENT buffer sets (raw columns + shuffled columns,
        # allocated once for the largest chunk) that alternate -- no allocation, no allocator bookkeeping across streams, nothing for the
        # caching allocator to give back and fetch again while an epoch runs.  Set j is refilled on the copy stream once the compute stream
        # has passed the event recorded behind the last batch that was cut from it.
        persistent = cache is None and not lists
        if persistent and getattr(self, "_sets", None) is None:
            rows = min(C, n)
            self._sets = [{"raw": {k: torch.empty((rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev) for k, v in self._pinned.items()},
                           "out": {k: torch.empty((rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev) for k, v in self._pinned.items()}
                           if self.shuffle else None, "free": None} for _ in range(2)]
            self._set_next = 0

        def upload(key, host_slice):
            """The chunk's part of one column on the device: copied from pinned memory, or -- device-resident mode -- the copy
            made by an earlier epoch (read-only: shuffled batches are index_select results, unshuffled ones views)."""
            if cache is not None and key in cache:
                return cache[key]
            t = host_slice().to(dev, non_blocking=True)
            if cache is not None:
                cache[key] = t
            return t

        PIECE = 1 << 20  # rows per host-to-device copy: ~4 MB, ~70 us on the link

        class _Staging:
            """A chunk on its way into buffer set `st`, as a sequence of SMALL steps the consumer loop advances a few at a time between
            batches.  Measured (round 6, rocprofv3 kernel + copy trace of a streaming fit): while one of the 32 MB column copies of the
            earlier all-at-once staging was in flight the train step took 1.8 ms instead of 0.99 -- kernels of the step queue behind the
            0.6 ms copy that happens to share their hardware queue (stream priorities and GPU_MAX_HW_QUEUES changed nothing); a
            chunk's 23 ms of copies cost 12 ms of step time: 0.79-0.87 of the resident rate.  In pieces of 4 MB a kernel waits 70 us
            at most, and spread over the batches of the previous chunk the host never spends more than a few calls at once."""

            def __init__(self_, a, rng):
                self_.b = min(a + C, n)
                self_.a, self_.m = a, min(a + C, n) - a
                self_.st = self._sets[self._set_next]
                self._set_next ^= 1
                self_.rng, self_.cols, self_.done, self_.ev = rng, {}, False, None
                self_.steps = self_._run()
                self_.n_steps = len(self._pinned) * (-(-self_.m // PIECE) + 1) + 1

            def _run(self_):
                st, a, m = self_.st, self_.a, self_.m
                with torch.cuda.stream(copy_stream):
                    if st["free"] is not None:
                        copy_stream.wait_event(st["free"])  # the batches cut from this set have all been consumed
                    perm = None
                    if self.shuffle:
                        g = torch.Generator(device=dev).manual_seed(int(self_.rng.integers(0, 2 ** 62)))
                        perm = torch.randperm(m, device=dev, generator=g)
                yield
                for k, v in self._pinned.items():
                    raw = st["raw"][k][:m]
                    for p0 in range(0, m, PIECE):
                        p1 = min(p0 + PIECE, m)
                        with torch.cuda.stream(copy_stream):
                            raw[p0:p1].copy_(v[a + p0:a + p1], non_blocking=True)
                        yield
                    with torch.cuda.stream(copy_stream):
                        if perm is not None:
                            out = st["out"][k][:m]
                            torch.index_select(raw, 0, perm, out=out)
                            self_.cols[k] = out
                        else:
                            self_.cols[k] = raw
                    yield
                with torch.cuda.stream(copy_stream):
                    self_.ev = torch.cuda.Event()
                    self_.ev.record(copy_stream)
                self_.done = True

            def advance(self_, k):
                for _ in range(k):
                    if self_.done or next(self_.steps, StopIteration) is StopIteration:
                        return

            def result(self_):
                while not self_.done:
                    if next(self_.steps, StopIteration) is StopIteration:
                        break
                return self_.cols, {}, self_.ev, self_.m, self_.st

        def stage(a: int, rng=rng):
            if persistent:
                return _Staging(a, rng)
            b = min(a + C, n)
            m = b - a
            host_offs = {}
            with torch.cuda.stream(copy_stream):
                perm = None
                if self.shuffle:
                    if lists:  # ragged columns need the permuted offsets on the HOST (batch bounds): the permutation is made there
                        perm_h = rng.permutation(m)
                        perm = torch.from_numpy(perm_h).pin_memory().to(dev, non_blocking=True)
                    else:
                        g = torch.Generator(device=dev).manual_seed(int(rng.integers(0, 2 ** 62)))
                        perm = torch.randperm(m, device=dev, generator=g)
                cols = {}
                for k, v in self._pinned.items():
                    if isinstance(v, tuple):
                        vals, offs = v
                        o = offs[a:b + 1]
                        dv = upload((k, a), lambda: vals[int(o[0]):int(o[-1])])
                        lens_h = np.diff(o)
                        if perm is not None:
                            new_lens = lens_h[perm_h]
                            new_o = np.zeros(m + 1, dtype=np.int64)
                            np.cumsum(new_lens, out=new_o[1:])
                            # value j of the permuted chunk comes from position src_start(row) + (j - new_start(row))
                            shift = torch.from_numpy((o[:-1][perm_h] - o[0]) - new_o[:-1]).pin_memory().to(dev, non_blocking=True)
                            reps = torch.from_numpy(new_lens).pin_memory().to(dev, non_blocking=True)
                            idx = torch.repeat_interleave(shift, reps, output_size=int(new_o[-1])) + torch.arange(int(new_o[-1]), device=dev)
                            dv = dv.index_select(0, idx)
                            o = new_o
                        else:
                            o = o - o[0]
                        host_offs[k] = o
                        cols[k] = (dv, torch.from_numpy(o.astype(np.int64 if dv.dtype == torch.int64 else np.int32)).pin_memory().to(dev, non_blocking=True))
                    else:
                        t = upload((k, a), lambda: v[a:b])
                        cols[k] = t.index_select(0, perm) if perm is not None else t
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return cols, host_offs, ev, m, None

        # the first chunk of THIS epoch may have been staged while the previous epoch's last chunk was being consumed
        pre = getattr(self, "_prestaged", None)
        self._prestaged = None
        if pre is not None and pre[0] == epoch:
            nxt, starts, rng = pre[1], pre[2], pre[3]  # the plan and the generator that staged it: later chunks continue its stream
        else:
            nxt = stage(starts[0])
        for ci in range(len(starts)):
            cols, host_offs, ev, m, bufset = nxt.result() if hasattr(nxt, "result") else nxt  # (a staging of the previous epoch is an instance of ITS local class)
            pending = None  # the staging that advances while this chunk's batches are handed out
            if ci + 1 < len(starts):
                nxt = stage(starts[ci + 1], rng)  # the next chunk's copies overlap this chunk's steps
            else:
                # last chunk of the epoch: stage the first chunk of the NEXT epoch now (fit() iterates the loader once per epoch;
                # without this every epoch starts with an exposed chunk copy -- 336 MB per 2 M Criteo rows)
                nxt = None
                st2, r2 = plan(epoch + 1)
                self._prestaged = (epoch + 1, stage(st2[0], r2), st2, r2)
            pending = nxt if nxt is not None else self._prestaged[1]
            nbatch = max(-(-m // B), 1)
            per_batch = -(-pending.n_steps // nbatch) if hasattr(pending, "n_steps") else 0
            cur_stream.wait_event(ev)
            if bufset is None:
                for v in cols.values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        t.record_stream(cur_stream)  # allocated on the copy stream, consumed on the compute stream
            for a in range(0, m, B):
                b = min(a + B, m)
                if b - a < B and self.drop_last:
                    break
                dev_b = {}
                for k, v in cols.items():
                    if isinstance(v, tuple):
                        o = host_offs[k]
                        dev_b[k + "__values"] = v[0][int(o[a]):int(o[b])]
                        dev_b[k + "__offsets"] = v[1][a:b + 1] - v[1][a] if a else v[1][:b + 1]
                    elif k in self.cont_names or k in self.label_names:
                        dev_b[k] = v[a:b].reshape(-1, 1)
                    else:
                        dev_b[k] = v[a:b]
                if per_batch:
                    pending.advance(per_batch)  # a few small copies of the NEXT chunk behind every batch of this one
                if not self.label_names:
                    yield dev_b, None
                else:
                    labels = {nm: dev_b.pop(nm) for nm in self.label_names}
                    yield dev_b, (labels[self.label_names[0]] if len(labels) == 1 else labels)
            if hasattr(pending, "result"):
                pending.result()  # whatever is left of the next chunk's staging is enqueued before this chunk is given up
            if bufset is not None:  # everything the consumer enqueued for this chunk's batches lies before this point of its stream
                bufset["free"] = torch.cuda.Event()
                bufset["free"].record(cur_stream)

    def _typed(self, cols: Columns) -> Columns:
        """ids -> int32 / int64 (values and offsets of a list share the dtype), continuous / targets -> float32."""
        out: Columns = {}
        for n in self.cat_names:
            v = cols[n]
            if isinstance(v, tuple):
                vals, offs = v
                vals = vals.astype(np.int64 if vals.dtype.itemsize > 4 else np.int32, copy=False)
                out[n] = (vals, offs.astype(vals.dtype))  # kernels want one integer dtype for both
            else:
                if not np.issubdtype(v.dtype, np.integer):
                    raise TypeError(f"categorical column {n!r} must hold integer ids, got {v.dtype}")
                out[n] = v.astype(np.int64 if v.dtype.itemsize > 4 else np.int32, copy=False)
        for n in self.cont_names + self.label_names:
            v = cols[n]
            if isinstance(v, tuple):
                raise TypeError(f"list-valued continuous / target column {n!r} is outside the hot path")
            out[n] = v.astype(np.float32, copy=False)
        return out

    def __len__(self) -> int:
        full, rem = divmod(self.n_rows, self.batch_size)
        return full + (1 if rem and not self.drop_last else 0)

    # --- host side: one batch as numpy arrays -------------------------------------------------------------
    @staticmethod
    def _flatten(cols: Columns) -> Dict[str, np.ndarray]:
        out: Dict[str, np.ndarray] = {}
        for n, v in cols.items():
            if isinstance(v, tuple):
                out[n + "__values"], out[n + "__offsets"] = v
            else:
                out[n] = v
        return out

    def _host_batch(self, idx: Optional[np.ndarray], contiguous: Optional[Tuple[int, int]]):
        cols = _slice_rows(self.columns, *contiguous) if contiguous is not None else _take_rows(self.columns, idx)
        return self._flatten(cols)

    def _host_batches_streaming(self, epoch: int) -> Iterator[Dict[str, np.ndarray]]:
        """Row groups -> chunks of >= buffer_rows rows -> batches; the tail of a chunk is carried into the next."""
        import pyarrow.parquet as pq

        rng = np.random.default_rng(self.seed + epoch) if self.shuffle else None
        groups = list(self._stream)
        if rng is not None:
            groups = [groups[i] for i in rng.permutation(len(groups))]
        left = self.n_rows  # rows this rank may still emit (every rank emits the same number)
        carry: Optional[Columns] = None
        parts: List[Columns] = []
        have = 0
        files: Dict[str, "pq.ParquetFile"] = {}
        for gi, (f, g, _) in enumerate(groups):
            if left <= 0:
                break
            pf = files.get(f)
            if pf is None:
                pf = files[f] = pq.ParquetFile(f)
            parts.append(self._typed(_table_to_columns(pf.read_row_group(g, columns=self._names), self._names)))
            have += _n_rows(parts[-1])
            if have < self.buffer_rows and gi != len(groups) - 1:
                continue
            chunk = _concat_rows(([carry] if carry is not None else []) + parts)
            parts, have, carry = [], 0, None
            n = _n_rows(chunk)
            if rng is not None:
                chunk = _take_rows(chunk, rng.permutation(n))
            a = 0
            while left > 0:
                take = min(self.batch_size, left)
                if n - a < take:
                    break  # not enough rows left in this chunk: carry them over
                if take < self.batch_size and self.drop_last:
                    left = 0
                    break
                yield self._flatten(_slice_rows(chunk, a, a + take))
                left -= take
                a += take
            if left > 0 and a < n:
                carry = _slice_rows(chunk, a, n)

    def _to_device(self, host: Dict[str, np.ndarray]):
        dev = {}
        for k, a in host.items():
            a = np.ascontiguousarray(a)
            if self.device.type == "cuda":
                if not a.flags.writeable:  # arrow-backed views are read-only; torch only wraps writable arrays
                    a = a.copy()
                t = torch.from_numpy(a).pin_memory().to(self.device, non_blocking=True)
            else:
                t = torch.from_numpy(a.copy() if not a.flags.writeable else a)
            base = k.split("__")[0]
            if base in self.cont_names or base in self.label_names:
                t = t.reshape(-1, 1)
            dev[k] = t
        if not self.label_names:
            return dev, None
        labels = {n: dev.pop(n) for n in self.label_names}
        return dev, (labels[self.label_names[0]] if len(labels) == 1 else labels)

    def __iter__(self) -> Iterator:
        epoch = self._epoch
        self._epoch += 1
        if self._stream is None and self._pinned is not None:
            yield from self._device_chunk_batches(epoch)
            return
        order = None
        if self.shuffle and self._stream is None:  # host path only (a 4 M-row permutation is 43 ms of host time per epoch)
            order = np.random.default_rng(self.seed + epoch).permutation(self.n_rows) + self.lo
        nb = len(self)

        def in_memory():
            for i in range(nb):
                a = i * self.batch_size
                b = min(a + self.batch_size, self.n_rows)
                if order is None:
                    yield self._host_batch(None, (self.lo + a, self.lo + b))
                else:
                    yield self._host_batch(order[a:b], None)

        source = self._host_batches_streaming(epoch) if self._stream is not None else in_memory()
        if not self.prefetch:
            for host in source:
                yield self._to_device(host)
            return
        copy_stream = torch.cuda.Stream(device=self.device)
        cur_stream = torch.cuda.current_stream(self.device)

        def stage():
            host = next(source, None)
            if host is None:
                return None
            with torch.cuda.stream(copy_stream):
                batch = self._to_device(host)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return batch, ev

        nxt = stage()
        while nxt is not None:
            (inputs, targets), ev = nxt
            nxt = stage()  # copy of the next batch overlaps this batch's step
            cur_stream.wait_event(ev)
            for t in list(inputs.values()) + ([targets] if isinstance(targets, torch.Tensor) else list((targets or {}).values())):
                t.record_stream(cur_stream)  # allocated on the copy stream, consumed on the compute stream
            yield inputs, targets
