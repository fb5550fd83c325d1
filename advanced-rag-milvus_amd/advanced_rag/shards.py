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


class PartialAppend(RuntimeError):
    """A multi-shard append failed after some pieces had gone in: global rows [base, end) exist in the shards."""

    def __init__(self, base: int, end: int, cause: Exception):
        super().__init__(f"append stopped after rows [{base}, {end}): {cause}")
        self.base, self.end, self.cause = base, end, cause


class ShardSet:
    def __init__(self, handles: list):
        if not handles:
            raise ValueError("a ShardSet needs at least one shard handle")
        self.handles = list(handles)
        self.rows_of: List[np.ndarray] = [np.zeros(0, np.int64) for _ in handles]  # global row of each local row
        self._n = 0
        self._packed = None   # (filter array, {shard: packed mask}) of the filter used last
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

    def add(self, dense, sparse_csr=None, n: Optional[int] = None):
        """Append rows (dense [n, dim] and/or a CSR triple of n rows) to the shards; both parts of a row go to the
        same shard.  `dense` is a numpy array (host rows, uploaded) or a CUDA tensor already in the shard's storage
        dtype (an encoder's output: re-tiled device to device by hr_add_dense_raw_dev, no host hop).
        Returns (base, end, sparse_error): the global row range [base, end) and, if a shard refused
        the sparse part of its piece, that error — the piece then got EMPTY sparse rows instead, because the row
        number is the only join key between the dense rows, the sparse rows and the host payload columns.
        A dense failure part-way (say, one device out of memory) raises PartialAppend carrying the rows that did go
        in, so that the caller can keep its payload columns aligned with them."""
        n = dense.shape[0] if dense is not None else (len(sparse_csr[0]) - 1 if sparse_csr is not None else int(n or 0))
        base = self._n
        sparse_error = None
        on_device = dense is not None and hasattr(dense, "is_cuda")
        for s, lo, hi in self._pieces(n):
            h = self.handles[s]
            if dense is not None:
                try:
                    if on_device:
                        self._add_dense_device(h, dense[lo:hi])
                    else:
                        h.add_dense(dense[lo:hi])
                except Exception as e:
                    if self._n > base:   # earlier pieces are in: tell the caller how many rows the set gained
                        raise PartialAppend(base, self._n, e) from e
                    raise
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

    @staticmethod
    def _add_dense_device(h, rows):
        """rows: CUDA tensor [m, dim] in the shard's storage dtype (fp16 / fp32); copied to the shard's GPU if needed."""
        import torch
        want = torch.float16 if h.dtype == 1 else torch.float32
        if rows.dtype != want:
            raise ValueError(f"device rows are {rows.dtype}, the shard stores {want}")
        if rows.dim() != 2 or rows.shape[1] != h.dim:
            raise ValueError(f"rows must be [n,{h.dim}], got {tuple(rows.shape)}")
        if rows.device.index != h.device:
            rows = rows.to(f"cuda:{h.device}")
        rows = rows.contiguous()
        stream = torch.cuda.current_stream(rows.device)
        h.add_dense_dev(rows.data_ptr(), rows.shape[0], stream.cuda_stream)   # synchronises the stream before it returns

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
    def _local_mask(self, s: int, keep: Optional[np.ndarray]):
        """Shard s's packed row mask of a boolean filter over GLOBAL rows -> positional arguments (rowmask, d_rowmask)
        of the handle's search.  Cut out and packed ONCE per filter (the manager hands the same array object for the
        same expression) and, for real shard handles, uploaded once: later searches with that filter pass a device
        pointer — no gather over the rows and no N/8-byte upload per search."""
        if keep is None:
            return ()
        ent = self._packed
        if ent is None or ent[0] is not keep:
            self._packed = ent = (keep, {})
        hit = ent[1].get(s)
        if hit is None:
            own = keep if (self.n_shards == 1 and len(self.rows_of[0]) == len(keep)) else keep[self.rows_of[s]]
            packed = np.packbits(own, bitorder="little")
            h = self.handles[s]
            if getattr(h, "_h", None) is not None:   # a libhbmrag shard: keep the mask in its HBM
                import torch
                hit = (None, torch.from_numpy(packed).to(f"cuda:{h.device}"))
            else:
                hit = (packed, None)
            ent[1][s] = hit
        packed, dev = hit
        return (packed,) if dev is None else (None, dev.data_ptr())

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
            return self.handles[s].search_dense(q, k, *self._local_mask(s, keep))
        return self._gather(self._fan_out(one), k)

    def search_sparse(self, queries, k: int, drop_ratio: float = 0.0, keep: Optional[np.ndarray] = None):
        def one(s):
            if self.handles[s].num_sparse_rows == 0:
                return np.full((len(queries), k), -1, np.int64), np.zeros((len(queries), k), np.float32)
            return self.handles[s].search_sparse(queries, k, drop_ratio, *self._local_mask(s, keep))
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


class CollectiveShardSet:
    """The torchrun form: one PROCESS per GPU, each owning the shard(s) of a contiguous global row range, searched
    as one collection from rank 0.

    Rank 0 is the front end (it owns the payload columns and answers `search`); the other ranks run `serve()`.
    One search = broadcast of a fixed-size header, broadcast of the query (+ the packed filter mask, if any), the
    per-rank HIP search, ONE gather of the packed per-rank lists to rank 0, the same merge as `ShardSet`.  The
    collectives are torch.distributed's (backend "nccl" = RCCL over xGMI on a GPU node; "gloo" in the CPU tests and
    one-GPU rehearsals); searches are serialised by a lock, so every rank sees the same sequence of commands.

    The local shard must already carry its global row numbers (ShardHandle.set_row_offset(first_row)), i.e. a
    local ShardSet of one handle; `first_row`/`n_local` say which slice of a global filter mask is this rank's."""

    OP_STOP, OP_DENSE, OP_SPARSE = 0, 1, 2

    def __init__(self, local: ShardSet, first_row: int, dist, group=None, device=None):
        import threading

        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.local, self.first_row = local, int(first_row)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else (
            torch.device("cuda", local.device) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        self._lock = threading.Lock()
        n = torch.tensor([local.num_rows, local.num_sparse_rows], dtype=torch.int64, device=self.dev)
        dist.all_reduce(n, group=group)
        self._n_rows, self._n_sparse = int(n[0]), int(n[1])

    # shape, as ShardSet
    n_shards = property(lambda self: self.world)
    first = property(lambda self: self.local.first)
    device = property(lambda self: self.local.device)
    num_rows = property(lambda self: self._n_rows)
    num_sparse_rows = property(lambda self: self._n_sparse)
    rows_of = property(lambda self: self.local.rows_of)

    def finalize(self):
        self.local.finalize()

    def close(self):
        self.local.close()

    # ------------------------------------------------------------------ protocol
    def _bcast(self, t):
        self.dist.broadcast(t, src=0, group=self.group)
        return t

    def _header(self, values=None):
        t = self.torch
        h = t.zeros(8, dtype=t.int64, device=self.dev)
        if values is not None:
            h[: len(values)] = t.tensor(values, dtype=t.int64)
        return self._bcast(h).tolist()

    def _exchange_mask(self, keep, n_rows: int, has_mask: bool):
        """rank 0 broadcasts the packed filter over GLOBAL rows; every rank cuts out its own rows."""
        if not has_mask:
            return None
        t = self.torch
        packed = t.from_numpy(np.packbits(keep, bitorder="little")).to(self.dev) if self.rank == 0 else \
            t.empty((n_rows + 7) // 8, dtype=t.uint8, device=self.dev)
        bits = np.unpackbits(self._bcast(packed).cpu().numpy(), bitorder="little")[:n_rows].astype(bool)
        n_local = self.local.num_rows if self.local.num_rows else self.local.num_sparse_rows
        return bits[self.first_row:self.first_row + n_local]

    def _collect(self, ids: np.ndarray, scores: np.ndarray, k: int):
        """ONE gather: [B, 2k] int64 per rank = ids, then the fp32 score bits."""
        t = self.torch
        B = ids.shape[0]
        pack = np.concatenate([ids.astype(np.int64), scores.astype(np.float32).view(np.int32).astype(np.int64)], axis=1)
        mine = t.from_numpy(np.ascontiguousarray(pack)).to(self.dev)
        parts = [t.empty_like(mine) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(mine, parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        parts = [p.cpu().numpy() for p in parts]
        return merge_lists([p[:, :k] for p in parts],
                           [p[:, k:].astype(np.int32).view(np.float32).reshape(B, k) for p in parts], k)

    def _run(self, op: int, B: int, k: int, extra: int, has_mask: bool, drop_bits: int, payload, keep):
        t = self.torch
        if op == self.OP_DENSE:
            dim = extra
            q = payload if self.rank == 0 else t.empty((B, dim), dtype=t.float32, device=self.dev)
            q = self._bcast(q).cpu().numpy()
            local_keep = self._exchange_mask(keep, self._n_rows, has_mask)
            ids, sc = self.local.search_dense(q, k, local_keep)
        else:
            nnz = extra
            if self.rank == 0:
                ptr, idx, val = payload
            else:
                ptr = t.empty(B + 1, dtype=t.int64, device=self.dev)
                idx = t.empty(nnz, dtype=t.int32, device=self.dev)
                val = t.empty(nnz, dtype=t.float32, device=self.dev)
            ptr, idx, val = (self._bcast(x).cpu().numpy() for x in (ptr, idx, val))
            local_keep = self._exchange_mask(keep, self._n_sparse, has_mask)
            queries = [(idx[ptr[b]:ptr[b + 1]], val[ptr[b]:ptr[b + 1]]) for b in range(B)]
            drop = float(np.array([drop_bits], dtype=np.int64).view(np.float64)[0])
            ids, sc = self.local.search_sparse(queries, k, drop, local_keep)
        return self._collect(ids, sc, k)

    # ------------------------------------------------------------------ rank 0
    def search_dense(self, q: np.ndarray, k: int, keep: Optional[np.ndarray] = None):
        t = self.torch
        q = np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)
        with self._lock:
            self._header([self.OP_DENSE, q.shape[0], k, q.shape[1], int(keep is not None), 0])
            return self._run(self.OP_DENSE, q.shape[0], k, q.shape[1], keep is not None, 0,
                             t.from_numpy(q).to(self.dev), keep)

    def search_sparse(self, queries, k: int, drop_ratio: float = 0.0, keep: Optional[np.ndarray] = None):
        t = self.torch
        ptr = np.zeros(len(queries) + 1, dtype=np.int64)
        for b, (qi, _) in enumerate(queries):
            ptr[b + 1] = ptr[b] + len(qi)
        idx = np.concatenate([np.asarray(qi, np.int32) for qi, _ in queries]) if ptr[-1] else np.zeros(0, np.int32)
        val = np.concatenate([np.asarray(qv, np.float32) for _, qv in queries]) if ptr[-1] else np.zeros(0, np.float32)
        drop_bits = int(np.array([drop_ratio], dtype=np.float64).view(np.int64)[0])
        with self._lock:
            self._header([self.OP_SPARSE, len(queries), k, int(ptr[-1]), int(keep is not None), drop_bits])
            payload = tuple(t.from_numpy(x).to(self.dev) for x in (ptr, idx, val))
            return self._run(self.OP_SPARSE, len(queries), k, int(ptr[-1]), keep is not None, drop_bits, payload, keep)

    def stop_workers(self):
        with self._lock:
            self._header([self.OP_STOP])

    # ------------------------------------------------------------------ ranks > 0
    def serve(self):
        """Answer rank 0's searches until it sends OP_STOP."""
        while True:
            op, B, k, extra, has_mask, drop_bits, _, _ = self._header()
            if op == self.OP_STOP:
                return
            self._run(int(op), int(B), int(k), int(extra), bool(has_mask), int(drop_bits), None, None)
