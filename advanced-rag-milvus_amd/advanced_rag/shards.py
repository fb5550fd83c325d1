"""A collection spread over several shard handles, searched as one.

The reference hides sharding behind the index manager: a collection is created
with `num_shards=4` and Milvus merges its segments server-side (reference
src/advanced_rag/indexing.py:234-239, :503-525).  `ShardSet` is the counterpart
for the in-HBM store: S shard handles (one per GPU of the node — or several on
one GPU, which is how the one-GPU tests exercise it), rows appended in balanced
pieces, every search run on all shards in parallel (one thread per shard: the
ctypes calls release the GIL and each handle owns its streams) and the S
per-shard lists merged by the same (score desc, global row asc) rule the kernels
use, so the answer is the one a single shard holding all rows would give.

Rows keep ONE global numbering (insertion order), which is what the host payload
columns are keyed by; each shard records which global rows it holds
(`rows_of[s]`, ascending — so a shard's local order is the global order and
per-shard lists stay exactly ordered after renumbering).  Row filters arrive as
a boolean array over global rows and are cut into per-shard bitmasks.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np


def merge_lists(ids: Sequence[np.ndarray], scores: Sequence[np.ndarray], k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge per-shard top-k lists ([B,k] each, -1 padded) -> [B,k] by (score desc, id asc)."""
    all_ids = np.concatenate(ids, axis=1)
    all_sc = np.concatenate(scores, axis=1)
    B = all_ids.shape[0]
    out_ids = np.full((B, k), -1, dtype=np.int64)
    out_sc = np.zeros((B, k), dtype=np.float32)
    for b in range(B):
        live = np.nonzero(all_ids[b] >= 0)[0]
        order = live[np.lexsort((all_ids[b][live], -all_sc[b][live].astype(np.float64)))][:k]
        out_ids[b, :len(order)] = all_ids[b][order]
        out_sc[b, :len(order)] = all_sc[b][order]
    return out_ids, out_sc


class ShardSet:
    def __init__(self, handles: list):
        if not handles:
            raise ValueError("a ShardSet needs at least one shard handle")
        self.handles = list(handles)
        self.rows_of: List[np.ndarray] = [np.zeros(0, np.int64) for _ in handles]  # global row of each local row
        self._n = 0
        self._pool = ThreadPoolExecutor(max_workers=len(handles), thread_name_prefix="shard-") if len(handles) > 1 else None

    # ------------------------------------------------------------------ shape
    @property
    def n_shards(self) -> int:
        return len(self.handles)

    @property
    def first(self):
        return self.handles[0]

    @property
    def device(self) -> int:
        return self.handles[0].device

    @property
    def num_rows(self) -> int:
        return sum(h.num_rows for h in self.handles)

    @property
    def num_sparse_rows(self) -> int:
        return sum(h.num_sparse_rows for h in self.handles)

    @property
    def device_bytes(self) -> int:
        return sum(h.device_bytes for h in self.handles)

    # ------------------------------------------------------------------ ingest
    def _pieces(self, n: int) -> List[Tuple[int, int, int]]:
        """Cut a batch of n rows into contiguous pieces, one per shard, filling the emptiest shards first:
        -> [(shard, lo, hi)] with the pieces in row order."""
        S = self.n_shards
        if S == 1:
            return [(0, 0, n)]
        have = np.array([len(r) for r in self.rows_of], dtype=np.int64)
        target = (have.sum() + n + S - 1) // S
        want = np.maximum(target - have, 0)
        pieces, lo = [], 0
        for s in np.argsort(have, kind="stable"):
            take = int(min(want[s], n - lo))
            if take > 0:
                pieces.append((int(s), lo, lo + take))
                lo += take
        if lo < n:  # rounding leftovers: to the emptiest shard
            s = int(np.argmin(have))
            pieces.append((s, lo, n))
        return pieces

    def add(self, dense: Optional[np.ndarray], sparse_csr=None, n: Optional[int] = None):
        """Append rows (dense [n, dim] and/or a CSR triple of n rows) to the shards; both parts of a row go to the
        same shard.  Returns (base, end, sparse_error): the global row range [base, end) and, if a shard refused
        the sparse part of its piece, that error — the piece then got EMPTY sparse rows instead, because the row
        number is the only join key between the dense rows, the sparse rows and the host payload columns."""
        n = dense.shape[0] if dense is not None else (len(sparse_csr[0]) - 1 if sparse_csr is not None else int(n or 0))
        base = self._n
        sparse_error = None
        for s, lo, hi in self._pieces(n):
            h = self.handles[s]
            if dense is not None:
                h.add_dense(dense[lo:hi])
            if sparse_csr is not None:
                ptr, idx, val = sparse_csr
                ptr = np.asarray(ptr, dtype=np.int64)
                try:
                    h.add_sparse(ptr[lo:hi + 1], idx, val)  # hr_add_sparse reads idx/val at absolute indptr positions
                except Exception as e:  # keep the numbering aligned, report
                    sparse_error = e
                    h.add_sparse(np.zeros(hi - lo + 1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))
            self.rows_of[s] = np.concatenate([self.rows_of[s], np.arange(base + lo, base + hi, dtype=np.int64)])
            self._n = base + hi  # pieces are in row order: a failure further on leaves a consistent prefix
        return base, base + n, sparse_error

    def finalize(self):
        for h in self.handles:
            h.finalize()

    def close(self):
        for h in self.handles:
            h.close()
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    # ------------------------------------------------------------------ search
    def _local_mask(self, s: int, keep: Optional[np.ndarray]) -> Optional[np.ndarray]:
        if keep is None:
            return None
        if self.n_shards == 1 and len(self.rows_of[0]) == len(keep):
            return np.packbits(keep, bitorder="little")
        return np.packbits(keep[self.rows_of[s]], bitorder="little")

    def _fan_out(self, fn):
        if self._pool is None:
            return [fn(0)]
        return list(self._pool.map(fn, range(self.n_shards)))

    def _gather(self, parts, k: int):
        if self.n_shards == 1:
            return parts[0]
        ids, scores = [], []
        for s, (li, sc) in enumerate(parts):
            gi = np.where(li >= 0, self.rows_of[s][np.maximum(li, 0)] if len(self.rows_of[s]) else -1, -1)
            ids.append(gi.astype(np.int64))
            scores.append(sc)
        return merge_lists(ids, scores, k)

    def search_dense(self, q: np.ndarray, k: int, keep: Optional[np.ndarray] = None):
        """q [B, dim] float32; keep = boolean filter over GLOBAL rows (or None) -> (ids [B,k] global rows, scores)."""
        def one(s):
            if self.handles[s].num_rows == 0:
                B = np.atleast_2d(q).shape[0]
                return np.full((B, k), -1, np.int64), np.zeros((B, k), np.float32)
            return self.handles[s].search_dense(q, k, self._local_mask(s, keep))
        return self._gather(self._fan_out(one), k)

    def search_sparse(self, queries, k: int, drop_ratio: float = 0.0, keep: Optional[np.ndarray] = None):
        def one(s):
            if self.handles[s].num_sparse_rows == 0:
                return np.full((len(queries), k), -1, np.int64), np.zeros((len(queries), k), np.float32)
            return self.handles[s].search_sparse(queries, k, drop_ratio, self._local_mask(s, keep))
        return self._gather(self._fan_out(one), k)

    # ------------------------------------------------------------------ snapshot
    def save(self, path_of_shard) -> None:
        """path_of_shard(s) -> file for shard s."""
        for s, h in enumerate(self.handles):
            h.save(path_of_shard(s))

    def row_maps(self) -> List[np.ndarray]:
        return [r.copy() for r in self.rows_of]

    def adopt(self, handles: list, rows_of: Sequence[np.ndarray]):
        """Replace the shards by loaded ones (snapshot resume)."""
        for h in self.handles:
            h.close()
        self.handles = list(handles)
        self.rows_of = [np.asarray(r, dtype=np.int64) for r in rows_of]
        self._n = int(sum(len(r) for r in self.rows_of))
