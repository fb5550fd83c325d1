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
    """The torchrun form: one PROCESS per GPU, each owning the shard of a contiguous global row range, searched as one
    collection from rank 0 — the reference's `num_shards` are invisible to the caller and cost it one RPC
    (indexing.py:232-239, :439-551); here a round of searches costs TWO collectives:

      1. ONE broadcast of a fixed-size packet (header + dense queries + sparse queries as CSR + the id of the filter
         mask): every rank learns what to search;
      2. every rank runs the dense and / or the sparse search of the round on its shard (device forms on a GPU shard,
         unproven lists repaired locally through the host form);
      3. ONE gather of the packed per-rank lists (both modalities) to rank 0, which merges them by (score desc, row asc).

    A round carries the searches of one or many retrieve() calls (the manager's batching front packs concurrent
    callers), so a lone retrieve() = 1 broadcast + 1 gather.  A filter's packed row mask travels ONCE, in an extra
    broadcast of the round that first uses it; every rank keeps its slice under the mask's id (same LRU on all ranks).
    Rank 0 validates and packs the whole round BEFORE the first collective and every rank always reaches the gather — a
    failing rank contributes empty lists and an error flag that rank 0 raises — so nobody is left waiting in a
    collective.  Collectives are torch.distributed's (backend "nccl" = RCCL over xGMI on a GPU node; "gloo" in the CPU
    tests and one-GPU rehearsals); rounds are serialised by a lock, so every rank sees the same sequence.

    Two forms of the local shard (a local ShardSet of one handle).  Pre-built: the rank filled its shard itself with one
    contiguous global row range; the handle carries the numbers (ShardHandle.set_row_offset(first_row)) and `first_row`
    says where the rank's rows sit in a global filter mask.  Ingested (`local_ids=True`): the handle numbers its rows from
    0, `local.rows_of[0]` maps them to global rows, and the collection grows through the COLLECTIVE control operations —
    add (the batch is broadcast, every rank keeps one contiguous block of it), finalize, save (every rank writes its own
    shard file), row_maps (one gather) — which rank 0 calls while the other ranks sit in serve(); each is one control
    packet + its payload broadcasts + one all-reduce of a status flag, so that rank 0 hears of a failure anywhere."""

    OP_STOP, OP_ROUND, OP_ADD, OP_FLUSH, OP_SAVE, OP_ROWMAPS, OP_HYBRID = 0, 1, 2, 3, 4, 5, 6
    HEADER = 16                     # int64 words
    PACKET_BYTES = 1 << 20          # header + queries of one round (128 x (768-d dense + 100-term sparse) = 0.5 MB)
    MAX_MASKS = 8
    hybrid_on_device = True         # False: retrieve() takes the two-searches-then-fuse rounds through the host forms (round 3)

    def __init__(self, local: ShardSet, first_row: int, dist, group=None, device=None, local_ids: bool = False):
        import threading

        import torch
        self.torch, self.dist, self.group = torch, dist, group
        self.local, self.first_row = local, int(first_row)
        # local_ids: the local handle numbers its rows from 0 and `local.rows_of[0]` says which global row each one is — the
        # form a collection INGESTED through this set takes (round 4: `add` is collective, every batch is cut into one
        # block per rank, so a rank's rows are no longer one contiguous global range).  Otherwise (a shard built by the rank
        # itself, hr_set_row_offset(first_row)) the handle already returns global rows of a contiguous range.
        self.local_ids = bool(local_ids)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else (
            torch.device("cuda", local.device) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        self._lock = threading.Lock()
        n = torch.tensor([local.num_rows, local.num_sparse_rows], dtype=torch.int64, device=self.dev)
        dist.all_reduce(n, group=group)
        self._n_rows, self._n_sparse = int(n[0]), int(n[1])
        self._packet = torch.zeros(self.PACKET_BYTES, dtype=torch.uint8, device=self.dev)
        self._masks = {}            # mask id -> this rank's boolean slice (LRU, same order on every rank)
        self._mask_ids = {}         # rank 0: id(filter array) -> (mask id, the array: keeps the id alive)
        self._next_mask_id = 1
        self.n_collectives = 0      # broadcasts + gathers issued (tests count them)
        self._side = None           # one worker thread: the dense search of a round that also carries sparse queries
        self._hyb_engines = {}      # (top_k, rrf_k) -> engine.HybridSearchEngine over the group (hybrid rounds on the device)
        self._dev_masks = {}        # mask id -> this rank's packed row mask in HBM

    # shape, as ShardSet
    n_shards = property(lambda self: self.world)
    first = property(lambda self: self.local.first)
    device = property(lambda self: self.local.device)
    num_rows = property(lambda self: self._n_rows)
    num_sparse_rows = property(lambda self: self._n_sparse)
    rows_of = property(lambda self: self.local.rows_of)

    def close(self):
        self.local.close()

    def _global_rows(self) -> np.ndarray:
        """Global row of every local row of this rank."""
        if self.local_ids:
            return self.local.rows_of[0]
        return np.arange(self.first_row, self.first_row + max(self.local.num_rows, self.local.num_sparse_rows), dtype=np.int64)

    # ------------------------------------------------------------------ collective control operations (rank 0 calls, the others serve())
    def _control(self, op: int, words=(), blob: bytes = b""):
        """rank 0: send a control packet (header words + an optional byte blob) — the workers pick it up in serve()."""
        hdr = np.zeros(self.HEADER, dtype=np.int64)
        hdr[0] = op
        hdr[1: 1 + len(words)] = list(words)
        hdr[self.HEADER - 1] = len(blob)
        body = np.concatenate([hdr.view(np.uint8), np.frombuffer(blob, dtype=np.uint8)]) if blob else hdr.view(np.uint8)
        if body.size > self.PACKET_BYTES:
            raise ValueError("control packet too large")
        return self._send_packet(body, whole=True)

    def _status(self, err: Optional[Exception]):
        """Every rank reports whether its part of a control operation worked; rank 0 raises if any did not."""
        t = self.torch
        flag = t.tensor([0 if err is None else 1], dtype=t.int64, device=self.dev)
        self.n_collectives += 1
        self.dist.all_reduce(flag, group=self.group)
        if err is not None and self.rank != 0:
            import logging
            logging.getLogger(__name__).error("rank %d failed a collective control operation: %s", self.rank, err)
        if self.rank == 0:
            if err is not None:
                raise err
            if int(flag.item()):
                raise RuntimeError("a collective control operation failed on another rank (see its log)")

    def add(self, dense, sparse_csr=None, n: Optional[int] = None):
        """Collective append (rank 0 calls it with the batch, the other ranks are in serve()): the batch travels in ONE
        broadcast per part (dense rows, CSR), every rank keeps the contiguous block shard_range(n, rank, world) of it and
        records which global rows those are (reference: Milvus spreads an insert over its `num_shards`,
        indexing.py:234-239, :264-437).  Same return value as ShardSet.add.  Needs `local_ids`."""
        if not self.local_ids:
            raise RuntimeError("this shard set was attached with pre-built contiguous shards; collective ingest needs local_ids=True")
        with self._lock:
            if dense is not None and hasattr(dense, "detach"):
                dense = dense.detach().float().cpu().numpy()
            nrows = dense.shape[0] if dense is not None else (len(sparse_csr[0]) - 1 if sparse_csr is not None else int(n or 0))
            dim = 0 if dense is None else int(dense.shape[1])
            if sparse_csr is not None:      # positions from 0 and exactly nnz entries: the receivers size their buffers by it
                ptr = np.asarray(sparse_csr[0], dtype=np.int64)
                sparse_csr = (ptr - ptr[0], np.asarray(sparse_csr[1])[ptr[0]:ptr[-1]], np.asarray(sparse_csr[2])[ptr[0]:ptr[-1]])
            nnz = 0 if sparse_csr is None else int(sparse_csr[0][-1])
            self._control(self.OP_ADD, (nrows, dim, 1 if dense is not None else 0, 1 if sparse_csr is not None else 0, nnz))
            return self._add_collective(nrows, dim, dense is not None, sparse_csr is not None, nnz, dense, sparse_csr)

    def _add_collective(self, nrows, dim, has_dense, has_sparse, nnz, dense=None, sparse_csr=None):
        from .engine import shard_range
        t = self.torch

        def bcast_array(arr, count, dtype):
            buf = (t.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(self.dev) if self.rank == 0
                   else t.empty(count, dtype=getattr(t, np.dtype(dtype).name), device=self.dev))
            return self._bcast(buf.reshape(-1)).cpu().numpy()

        err, out = None, (0, 0, None)
        try:
            d = bcast_array(dense, nrows * dim, np.float32).reshape(nrows, dim) if has_dense else None
            csr = None
            if has_sparse:
                ptr = bcast_array(sparse_csr[0] if self.rank == 0 else None, nrows + 1, np.int64)
                idx = bcast_array(sparse_csr[1] if self.rank == 0 else None, nnz, np.int32) if nnz else np.zeros(0, np.int32)
                val = bcast_array(sparse_csr[2] if self.rank == 0 else None, nnz, np.float32) if nnz else np.zeros(0, np.float32)
                csr = (ptr, idx, val)
            lo, hi = shard_range(nrows, self.rank, self.world)
            base = self._n_rows if has_dense or not has_sparse else self._n_sparse
            # the global row numbers are taken on every rank BEFORE the local append: a rank whose append fails leaves a
            # hole (rows no search returns) instead of ranks that disagree about the numbering
            if has_dense or not has_sparse:
                self._n_rows += nrows
            if has_sparse or not has_dense:
                self._n_sparse += nrows
            self._masks.clear()          # row slices of cached filters are stale on every rank
            self._mask_ids.clear()
            self._dev_masks.clear()
            sparse_err = None
            if hi > lo:
                piece_csr = None if csr is None else (csr[0][lo:hi + 1], csr[1], csr[2])
                b0 = self.local._n
                _, _, sparse_err = self.local.add(None if d is None else d[lo:hi], piece_csr, hi - lo)
                self.local.rows_of[0][b0:] = np.arange(base + lo, base + hi, dtype=np.int64)   # global rows of the block
            out = (base, base + nrows, sparse_err)
        except Exception as e:
            err = e
        try:
            self._status(err)
        except Exception as e:
            # the row numbers [base, base + nrows) are taken on every rank whatever failed: the caller keeps its payload
            # columns aligned with them (the failing rank's block stays a hole no search returns)
            base = out[0] if err is None else (self._n_rows if has_dense or not has_sparse else self._n_sparse) - nrows
            raise PartialAppend(base, base + nrows, e) from e
        return out

    def finalize(self):
        """Flush every rank's shard (rank 0 calls it; a rank in serve() flushes when the packet arrives)."""
        if self.rank != 0 or not self.local_ids:
            self.local.finalize()
            return
        with self._lock:
            self._control(self.OP_FLUSH)
            self._flush_collective()

    def _flush_collective(self):
        err = None
        try:
            self.local.finalize()
        except Exception as e:
            err = e
        self._status(err)

    def save(self, path_of_shard) -> None:
        """Every rank writes its shard to path_of_shard(rank) (rank 0 calls it; the paths travel in the control packet)."""
        with self._lock:
            paths = "\n".join(path_of_shard(r) for r in range(self.world)).encode("utf-8")
            self._control(self.OP_SAVE, (), paths)
            self._save_collective(paths)

    def _save_collective(self, paths: bytes):
        err = None
        try:
            mine = paths.decode("utf-8").split("\n")[self.rank]
            self.local.save(lambda s: mine)
        except Exception as e:
            err = e
        self._status(err)

    def row_maps(self):
        """The global rows every rank's shard holds, in rank order (rank 0 calls it: one gather)."""
        with self._lock:
            self._control(self.OP_ROWMAPS)
            return self._rowmaps_collective()

    def _rowmaps_collective(self):
        rows = self._global_rows()
        parts = [None] * self.world if self.rank == 0 else None
        self.n_collectives += 1
        self.dist.gather_object(rows, parts, dst=0, group=self.group)
        return [np.asarray(r, dtype=np.int64) for r in parts] if self.rank == 0 else None

    # ------------------------------------------------------------------ protocol
    def _bcast(self, t):
        self.n_collectives += 1
        self.dist.broadcast(t, src=0, group=self.group)
        return t

    def _pack_round(self, dense_q, sparse_queries, k: int, drop: float, keep):
        """rank 0: the whole round as one byte packet (numpy); raises BEFORE any collective if something is wrong."""
        Bd = 0 if dense_q is None else int(dense_q.shape[0])
        Bs = 0 if sparse_queries is None else len(sparse_queries)
        parts, dim = [], 0
        if Bd:
            dense_q = np.ascontiguousarray(dense_q, dtype=np.float32)
            dim = dense_q.shape[1]
            parts.append(dense_q.view(np.uint8).reshape(-1))
        nnz = 0
        if Bs:
            ptr = np.zeros(Bs + 1, dtype=np.int64)
            for b, (qi, _) in enumerate(sparse_queries):
                ptr[b + 1] = ptr[b] + len(qi)
            nnz = int(ptr[-1])
            idx = np.concatenate([np.asarray(qi, np.int32) for qi, _ in sparse_queries]) if nnz else np.zeros(0, np.int32)
            val = np.concatenate([np.asarray(qv, np.float32) for _, qv in sparse_queries]) if nnz else np.zeros(0, np.float32)
            if idx.shape != val.shape:
                raise ValueError("sparse query indices/values length mismatch")
            parts += [ptr.view(np.uint8), idx.view(np.uint8), val.view(np.uint8)]
        mask_id, mask_new, mask_bytes = self._mask_for(keep)
        hdr = np.zeros(self.HEADER, dtype=np.int64)
        hdr[:10] = [self.OP_ROUND, Bd, Bs, k, dim, nnz, int(np.array([drop], dtype=np.float64).view(np.int64)[0]), mask_id,
                    mask_new, 0 if mask_bytes is None else mask_bytes.size]
        body = np.concatenate([hdr.view(np.uint8)] + parts) if parts else hdr.view(np.uint8)
        if body.size > self.PACKET_BYTES:
            raise ValueError(f"a round of {Bd} dense + {Bs} sparse queries needs {body.size} bytes; the packet holds {self.PACKET_BYTES}")
        return body, mask_bytes

    def _send_packet(self, body: Optional[np.ndarray], whole: bool = False):
        """ONE broadcast of the fixed-size packet; returns its bytes as numpy on every rank."""
        t = self.torch
        if self.rank == 0:
            host = np.zeros(self.PACKET_BYTES, dtype=np.uint8)
            host[: body.size] = body
            self._packet.copy_(t.from_numpy(host))
        self._bcast(self._packet)
        if self.rank == 0:
            return host
        hdr = self._packet[: self.HEADER * 8].cpu().numpy().view(np.int64)   # then only the bytes the round uses
        if int(hdr[0]) != self.OP_ROUND:                                      # a control packet: header + its blob
            return self._packet[: self.HEADER * 8 + int(hdr[self.HEADER - 1])].cpu().numpy()
        need = self.HEADER * 8 + int(hdr[1]) * int(hdr[4]) * 4 + (int(hdr[2]) + 1) * 8 * (1 if hdr[2] else 0) + int(hdr[5]) * 8
        return self._packet[: min(need, self.PACKET_BYTES)].cpu().numpy()

    def _mask_slice(self, hdr, mask_bytes_rank0):
        """This rank's boolean slice of the round's filter (None = no filter); receives a new mask if the round carries one."""
        t = self.torch
        mask_id, mask_new, mask_len = int(hdr[7]), int(hdr[8]), int(hdr[9])
        if mask_id == 0:
            return None
        if mask_new:
            buf = t.from_numpy(mask_bytes_rank0).to(self.dev) if self.rank == 0 else t.empty(mask_len, dtype=t.uint8, device=self.dev)
            bits = np.unpackbits(self._bcast(buf).cpu().numpy(), bitorder="little").astype(bool)
            rows = self._global_rows()
            if rows.size and bits.size <= int(rows.max()):
                raise ValueError(f"filter mask covers {bits.size} rows, this rank holds rows up to {int(rows.max()) + 1}")
            if len(self._masks) >= self.MAX_MASKS:
                self._masks.pop(next(iter(self._masks)))
            self._masks[mask_id] = bits[rows]
        return self._masks[mask_id]

    def _local_round(self, pkt: np.ndarray, hdr, keep_local):
        """Run this rank's part of a round -> int64 [2 + n_mod * B * k * 2]: status, then ids and score bits per modality."""
        Bd, Bs, k, dim, nnz = (int(x) for x in hdr[1:6])
        drop = float(np.array([hdr[6]], dtype=np.int64).view(np.float64)[0])
        off = self.HEADER * 8
        out = []
        dense_job = None
        if Bd:
            q = pkt[off: off + Bd * dim * 4].view(np.float32).reshape(Bd, dim)
            off += Bd * dim * 4
            # the slice object itself when it fits: the local ShardSet keeps a filter's packed (device) mask by identity
            kd = None if keep_local is None else (keep_local if keep_local.size == self.local.num_rows else keep_local[: self.local.num_rows])
            if Bs:   # both modalities in the round: the dense search runs beside the sparse one (ctypes releases the GIL)
                if self._side is None:
                    self._side = ThreadPoolExecutor(max_workers=1, thread_name_prefix="round-dense-")
                dense_job = self._side.submit(self.local.search_dense, q, k, kd)
                out.append(None)
            else:
                out.append(self.local.search_dense(q, k, kd))
        if Bs:
            ptr = pkt[off: off + (Bs + 1) * 8].view(np.int64)
            off += (Bs + 1) * 8
            idx = pkt[off: off + nnz * 4].view(np.int32)
            off += nnz * 4
            val = pkt[off: off + nnz * 4].view(np.float32)
            queries = [(idx[ptr[b]:ptr[b + 1]], val[ptr[b]:ptr[b + 1]]) for b in range(Bs)]
            ks = None if keep_local is None else (keep_local if keep_local.size == self.local.num_sparse_rows
                                                  else keep_local[: self.local.num_sparse_rows])
            try:
                out.append(self.local.search_sparse(queries, k, drop, ks))
            finally:
                if dense_job is not None:   # never leave the side thread running into the next round
                    dense_err = dense_job.exception()
            if dense_job is not None:
                if dense_err is not None:
                    raise dense_err
                out[0] = dense_job.result()
        if self.local_ids:
            rows = self.local.rows_of[0]
            out = [(np.where(li >= 0, rows[np.maximum(li, 0)] if rows.size else -1, -1).astype(np.int64), sc) for li, sc in out]
        return out

    def _round(self, body, mask_bytes):
        """Both sides of a round after rank 0 has packed it.  Returns the merged lists on rank 0."""
        t = self.torch
        pkt = self._send_packet(body)
        hdr = pkt[: self.HEADER * 8].view(np.int64)
        if int(hdr[0]) == self.OP_STOP:
            return False
        if int(hdr[0]) == self.OP_HYBRID:
            return self._hybrid_round(pkt, hdr, mask_bytes)
        if int(hdr[0]) != self.OP_ROUND:     # a worker picked up a control operation of rank 0
            op = int(hdr[0])
            if op == self.OP_ADD:
                self._add_collective(int(hdr[1]), int(hdr[2]), bool(hdr[3]), bool(hdr[4]), int(hdr[5]))
            elif op == self.OP_FLUSH:
                self._flush_collective()
            elif op == self.OP_SAVE:
                self._save_collective(pkt[self.HEADER * 8: self.HEADER * 8 + int(hdr[self.HEADER - 1])].tobytes())
            elif op == self.OP_ROWMAPS:
                self._rowmaps_collective()
            return True
        Bd, Bs, k = int(hdr[1]), int(hdr[2]), int(hdr[3])
        n_vals = (Bd + Bs) * k
        mine = np.zeros(2 + 2 * n_vals, dtype=np.int64)
        mine[2: 2 + n_vals] = -1
        err = None
        try:
            lists = self._local_round(pkt, hdr, self._mask_slice(hdr, mask_bytes))
            ids = np.concatenate([li.reshape(-1) for li, _ in lists]).astype(np.int64)
            sc = np.concatenate([s.reshape(-1) for _, s in lists]).astype(np.float32)
            mine[2: 2 + n_vals] = ids
            mine[2 + n_vals:] = sc.view(np.int32).astype(np.int64)
        except Exception as e:   # still reach the gather: nobody may be left waiting in a collective
            err = e
            mine[0] = 1
        mine_t = t.from_numpy(mine).to(self.dev)
        parts = [t.empty_like(mine_t) for _ in range(self.world)] if self.rank == 0 else None
        self.n_collectives += 1
        self.dist.gather(mine_t, parts, dst=0, group=self.group)
        if err is not None and self.rank != 0:
            import logging
            logging.getLogger(__name__).error("rank %d failed its part of a search round: %s", self.rank, err)
        if self.rank != 0:
            return True
        if err is not None:
            raise err
        parts = [p.cpu().numpy() for p in parts]
        bad = [r for r, p in enumerate(parts) if p[0] != 0]
        if bad:
            raise RuntimeError(f"search round failed on rank(s) {bad}")
        out, o = [], 2
        for B in (Bd, Bs):
            if not B:
                out.append(None)
                continue
            ids = [p[o: o + B * k].reshape(B, k) for p in parts]
            scs = [p[2 + n_vals + (o - 2): 2 + n_vals + (o - 2) + B * k].astype(np.int32).view(np.float32).reshape(B, k) for p in parts]
            out.append(merge_lists(ids, scs, k))
            o += B * k
        return out

    # ------------------------------------------------------------------ hybrid rounds on the device
    @property
    def supports_hybrid_round(self) -> bool:
        """The device form of a hybrid round needs real shards whose handles return GLOBAL rows (pre-built contiguous
        shards: the engine merges the ranks' lists by the ids the scans wrote) and the sparse modality."""
        h = self.local.first
        return (self.hybrid_on_device and not self.local_ids and getattr(h, "_h", None) is not None
                and getattr(h, "sparse_dim", 0) > 0)

    def round_hybrid(self, dense_q: np.ndarray, sparse_queries, top_k: int, drop_ratio: float, rrf_k: int,
                     weights: np.ndarray, keep: Optional[np.ndarray] = None):
        """rank 0: the dense search (2 x top_k), the sparse search (2 x top_k) and their rank fusion for B requests as ONE
        collective round on the device: the packet (queries, CSR, per-request weights) is broadcast, every rank runs
        hr_search_hybrid_dev on its shard with pointers INTO the packet, the per-rank lists travel in the engine's one
        all-gather and every rank merges + fuses them in one launch (engine.HybridSearchEngine with the process group: the
        path bench.py times) — under nccl nothing but the 128-byte header and rank 0's answer crosses the host.
        -> dict of numpy arrays: fused_ids / fused_scores / fused_methods [B, top_k], fused_n [B], list_ids /
        list_scores [2, B, k'], proven [B] (False: some rank could not prove a list — the caller redoes that request
        through the host forms)."""
        from .engine import pack_sparse_queries
        dense_q = np.ascontiguousarray(np.atleast_2d(dense_q), dtype=np.float32)
        B, dim = dense_q.shape
        h = self.local.first
        if dim != h.dim:
            raise ValueError(f"query dim {dim} != shard dim {h.dim}")
        ptr, idx, val, max_nnz = pack_sparse_queries(list(sparse_queries), float(drop_ratio), h.sparse_dim)
        if len(ptr) != B + 1 or not idx.size:
            raise ValueError("a hybrid round needs one non-empty sparse query per dense query")
        w = np.zeros((B, 3), dtype=np.float64)
        w[:, :2] = np.asarray(weights, dtype=np.float64).reshape(B, 2)
        with self._lock:
            mask_id, mask_new, mask_bytes = self._mask_for(keep)
            # 8-byte items first: every view of the packet is aligned for its type
            blob = np.concatenate([ptr.astype(np.int64).view(np.uint8), w.view(np.uint8).reshape(-1), dense_q.view(np.uint8).reshape(-1),
                                   idx.astype(np.int32).view(np.uint8), val.astype(np.float32).view(np.uint8)])
            hdr = np.zeros(self.HEADER, dtype=np.int64)
            hdr[:11] = [self.OP_HYBRID, B, int(top_k), dim, int(idx.size), int(rrf_k), 0, mask_id, mask_new,
                        0 if mask_bytes is None else mask_bytes.size, int(max_nnz)]
            hdr[self.HEADER - 1] = blob.size
            body = np.concatenate([hdr.view(np.uint8), blob])
            if body.size > self.PACKET_BYTES:
                raise ValueError(f"a hybrid round of {B} requests needs {body.size} bytes; the packet holds {self.PACKET_BYTES}")
            return self._round(body, mask_bytes)

    def _hybrid_round(self, pkt: np.ndarray, hdr, mask_bytes):
        """Every rank's part of a hybrid round (collective: the engine's all-gather is inside)."""
        from .engine import EngineConfig, HybridSearchEngine
        t = self.torch
        B, top_k, dim, nnz, rrf_k = (int(x) for x in hdr[1:6])
        max_nnz = int(hdr[10])
        keep_local = self._mask_slice(hdr, mask_bytes)
        dev = t.device("cuda", self.local.first.device)
        eng = self._hyb_engines.get((top_k, rrf_k))
        if eng is None:
            if len(self._hyb_engines) >= 8:
                self._hyb_engines.clear()
            eng = self._hyb_engines[(top_k, rrf_k)] = HybridSearchEngine(
                self.local.first, EngineConfig(top_k=top_k, rrf_k=rrf_k, enable_reranking=False), process_group=self.group, device=str(dev))
        # the operands are views of the packet where it lives on the device (nccl), uploads of its host copy otherwise
        src = self._packet if self._packet.is_cuda else t.from_numpy(pkt[: self.HEADER * 8 + int(hdr[self.HEADER - 1])]).to(dev)
        o = self.HEADER * 8

        def view(n_bytes, dtype):
            nonlocal o
            v = src[o: o + n_bytes].view(dtype)
            o += n_bytes
            return v
        ptr = view((B + 1) * 8, t.int64)
        wq = view(B * 24, t.float64).view(B, 3)
        q = view(B * dim * 4, t.float32).view(B, dim)
        idx = view(nnz * 4, t.int32)
        val = view(nnz * 4, t.float32)
        mask = None
        if keep_local is not None:
            mid = int(hdr[7])
            mask = self._dev_masks.get(mid)
            if mask is None:
                if len(self._dev_masks) >= self.MAX_MASKS:
                    self._dev_masks.pop(next(iter(self._dev_masks)))
                mask = self._dev_masks[mid] = t.from_numpy(np.packbits(keep_local, bitorder="little")).to(dev)
        with t.cuda.device(dev):
            b = eng.search(q, (ptr, idx, val, max_nnz), rowmask=mask, weights=wq)
            t.cuda.current_stream(dev).synchronize()      # the packet may be rewritten by the next round
        if self.rank != 0:
            return True
        out = {k: b[k].cpu().numpy() for k in ("fused_ids", "fused_scores", "fused_methods", "fused_n", "list_ids", "list_scores")}
        out["proven"] = b["agg_flags"].cpu().numpy().min(axis=0) == 1
        return out

    def _mask_for(self, keep):
        """rank 0: (mask id, 1 if the mask must travel with this round, its packed bytes or None) of a filter array."""
        if keep is None:
            return 0, 0, None
        ent = self._mask_ids.get(id(keep))
        if ent is not None and ent[0] in self._masks:
            return ent[0], 0, None
        mask_id = self._next_mask_id
        self._next_mask_id += 1
        self._mask_ids = {key: v for key, v in self._mask_ids.items() if v[0] in self._masks}
        self._mask_ids[id(keep)] = (mask_id, keep)
        return mask_id, 1, np.packbits(np.asarray(keep, dtype=bool), bitorder="little")

    # ------------------------------------------------------------------ rank 0
    def round(self, dense_q: Optional[np.ndarray], sparse_queries, k: int, drop_ratio: float = 0.0, keep: Optional[np.ndarray] = None):
        """One round = one broadcast + one gather: dense_q [Bd, dim] and / or Bs sparse queries, all with the same k and
        filter -> ((dense ids, scores) | None, (sparse ids, scores) | None), global rows, merged over the ranks."""
        if dense_q is not None:
            dense_q = np.ascontiguousarray(np.atleast_2d(dense_q), dtype=np.float32)
        with self._lock:
            body, mask_bytes = self._pack_round(dense_q, sparse_queries, int(k), float(drop_ratio), keep)   # may raise: no collective yet
            return self._round(body, mask_bytes)

    def search_dense(self, q: np.ndarray, k: int, keep: Optional[np.ndarray] = None):
        return self.round(q, None, k, 0.0, keep)[0]

    def search_sparse(self, queries, k: int, drop_ratio: float = 0.0, keep: Optional[np.ndarray] = None):
        return self.round(None, list(queries), k, drop_ratio, keep)[1]

    def stop_workers(self):
        with self._lock:
            hdr = np.zeros(self.HEADER, dtype=np.int64)
            hdr[0] = self.OP_STOP
            self._send_packet(hdr.view(np.uint8))
        self._close_side()

    def _close_side(self):
        if self._side is not None:
            self._side.shutdown(wait=False)
            self._side = None

    # ------------------------------------------------------------------ ranks > 0
    def serve(self):
        """Answer rank 0's rounds until it sends OP_STOP."""
        while self._round(None, None):
            pass
        self._close_side()
