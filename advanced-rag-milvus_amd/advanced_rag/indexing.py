"""In-HBM shard store behind the reference's index-manager protocol.

`MilvusIndexManager` here is a drop-in for the reference class of the same name
(reference src/advanced_rag/indexing.py:80-713) with the Milvus server replaced
by libhbmrag: the three collections ("semantic_index" dense/COSINE,
"sparse_index" sparse/IP, "domain_index" dense/COSINE; indexing.py:143-180)
live in HBM as a tiled fp16/fp32 matrix plus doc-range postings — on one GPU, or
spread over several (`devices=[0, 1, ...]`: one shard handle per device, searched in
parallel and merged, the counterpart of the reference's server-side `num_shards`,
indexing.py:234-239) — and `search` runs the HIP kernels instead of an RPC
(indexing.py:503-525).
Host and port are accepted and ignored.  Payload columns (content, ids, scalar
fields of the schema, indexing.py:191-225) stay on the host, keyed by row.

Protocol consumed by HybridRetriever (retrieval.py:348-354, :634-648):
  async _generate_semantic_embedding(text) -> float32[dim]
  async _generate_sparse_embedding(text)   -> {"indices","values"} | ndarray
  async _generate_domain_embedding(text, domain)
  async search(query_embedding, collection_name, top_k, filters, search_params)
        -> [{id, content, score, metadata{doc_id, chunk_index, entropy,
             redundancy, domain_density, timestamp}}]   (best first)
plus index_chunks / get_collection_stats / delete_by_filter / close.
"""
from __future__ import annotations

import asyncio
import threading
import logging
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import filters as _filters
from .columns import FLOAT_COLUMNS, PayloadColumns
from .constants import IndexingConstants
from .embedding_cache import get_semantic_cache
from ._native import HR_MAX_TOPK   # importing the binding module does not load the library
from .shards import PartialAppend, ShardSet

logger = logging.getLogger(__name__)

# FLOAT fields of the collection schema (reference indexing.py:200-202) hold float32; a search hands them back as
# Python floats of the float32-rounded value, which is what the payload columns store
_FLOAT_FIELDS = ("entropy", "redundancy", "domain_density")


def _f32(v) -> float:
    return float(np.float32(v))


class IndexType(Enum):
    SEMANTIC = "semantic"
    SPARSE = "sparse"
    DOMAIN = "domain"
    HYBRID = "hybrid"


@dataclass
class IndexConfig:
    collection_name: str
    dimension: int
    index_type: str = "HNSW"   # accepted for compatibility; the store is always exact FLAT
    metric_type: str = "L2"
    index_params: Optional[Dict] = None
    enable_dynamic_field: bool = True

    def __post_init__(self):
        if self.index_params is None:
            self.index_params = {"M": 16, "efConstruction": 200}


class ShardCollection:
    """What `manager.collections[name]` holds: a named view of one ShardSet (the
    collection's shard handles, one per device) with the handful of Collection methods the reference calls
    (num_entities / flush / load / release / delete, indexing.py:259, :430, :683-701)."""

    def __init__(self, manager: "MilvusIndexManager", name: str, kind: str, handle, dim: int, metric: str):
        self.manager, self.name, self.kind, self.handle, self.dim, self.metric = manager, name, kind, handle, dim, metric

    @property
    def num_entities(self) -> int:
        return self.handle.num_sparse_rows if self.kind == "sparse" else self.handle.num_rows

    @property
    def schema(self) -> str:
        return (f"ShardCollection({self.name}, kind={self.kind}, dim={self.dim}, metric={self.metric}, "
                "fields=[id, chunk_id, doc_id, content, chunk_index, token_count, entropy, redundancy, "
                "domain_density, embedding, metadata_json, timestamp])")

    @property
    def indexes(self) -> List[str]:
        return [f"FLAT/{self.metric} (exact scan, gfx950)"]

    def flush(self):
        self.handle.finalize()

    load = flush

    def release(self):
        return None

    def delete(self, expr: str):
        self.manager._tombstone(expr)


class MilvusIndexManager:
    def __init__(self, host: str = "localhost", port: int = 19530, enable_sharding: bool = True, num_shards: int = 4,
                 semantic_dim: int = 1536, sparse_dim: int = 10000, domain_dim: int = 768, connect: bool = True,
                 *, dtype: str = "float16", device: int = 0, devices: Optional[Sequence[int]] = None,
                 enable_domain: bool = True, device_embedding_cache: int = 0, coalesce: bool = True):
        self.host, self.port = host, port
        self.enable_sharding, self.num_shards = enable_sharding, num_shards
        self.semantic_dim, self.sparse_dim, self.domain_dim = semantic_dim, sparse_dim, domain_dim
        # devices = GPU of every shard (a device may be named more than once); default: one shard on `device`
        self.devices = [int(d) for d in devices] if devices else [int(device)]
        self.dtype, self.device, self.enable_domain = dtype, self.devices[0], enable_domain
        self.collections: Dict[str, ShardCollection] = {}
        self.embedding_generator = None
        self.embedding_executor = ThreadPoolExecutor(max_workers=IndexingConstants.THREAD_POOL_WORKERS,
                                                     thread_name_prefix="embedding-")
        self._main = None      # ShardSet: semantic + sparse
        self._domain = None    # ShardSet: domain
        # payload columns keyed by global row: append-only numeric arrays and offset-encoded strings (columns.py); the
        # filterable ones are mirrored into HBM on first use (device_filters.py)
        self._cols = PayloadColumns()
        self._dev_filters = None
        self._deleted: Optional[np.ndarray] = None
        self._mask_cache: Dict[Any, Optional[np.ndarray]] = {}   # (expr, rows, delete epoch) -> boolean row filter
        self._delete_epoch = 0
        self._synthetic_rows = 0   # rows whose payload is derived from the row number (bulk/benchmark ingest)
        # "embedding cache -> device-resident tensor": with device_embedding_cache=N > 0 the semantic QUERY
        # embeddings are kept in one [N, dim] HBM tensor; a hit hands the search kernel a device pointer
        # (no host hop), a miss encodes once (on the device when the generator offers encode_to_device).
        self._dev_cache_slots = int(device_embedding_cache)
        self._dev_cache = None
        # concurrent search() calls are packed into batched device searches (batching.SearchCoalescer): the counterpart
        # of the Milvus server's request queue (reference indexing.py:503-506 fires one RPC per call and the server
        # batches).  coalesce=False keeps the round-2 behaviour: every call its own scan of the shard.
        self.coalesce = bool(coalesce)
        self._front = None
        self._dev_masks: Dict[Any, Any] = {}   # (expr, rows, delete epoch, kind) -> packed row mask in HBM
        self._mask_lock = threading.RLock()
        if connect:
            self._connect()
            self._initialize_collections()

    # ------------------------------------------------------------------ lifecycle
    def _connect(self):
        from . import _native
        self._native = _native
        _native.load_library()

    def _initialize_collections(self):
        nat = self._native
        store = nat.HR_F16 if self.dtype in ("float16", "fp16", "f16") else nat.HR_F32
        sparse_on = os.getenv("ENABLE_SPARSE", "1") == "1"
        self._main = ShardSet([nat.ShardHandle(self.semantic_dim, store, nat.HR_METRIC_COSINE,
                                               self.sparse_dim if sparse_on else 0, d) for d in self.devices])
        self.collections["semantic_index"] = ShardCollection(self, "semantic_index", "dense", self._main,
                                                             self.semantic_dim, "COSINE")
        if sparse_on:
            self.collections["sparse_index"] = ShardCollection(self, "sparse_index", "sparse", self._main,
                                                               self.sparse_dim, "IP")
        if self.enable_domain:
            self._domain = ShardSet([nat.ShardHandle(self.domain_dim, store, nat.HR_METRIC_COSINE, 0, d)
                                     for d in self.devices])
            self.collections["domain_index"] = ShardCollection(self, "domain_index", "dense", self._domain,
                                                               self.domain_dim, "COSINE")

    def attach_shards(self, handles, rows_of=None, synthetic_rows: int = 0, process_group=None, first_row: int = 0,
                      local_ids: bool = False):
        """Adopt already-built shard handles as the semantic (+ sparse) collection — e.g. the shard a benchmark has
        just filled.  rows_of[s] = global row of every local row of shard s (default: shard s follows shard s-1).

        With `process_group` (torch.distributed, one process per GPU) the collection spans the group's ranks: this
        rank's handle holds global rows [first_row, first_row + its rows) and must carry that offset
        (ShardHandle.set_row_offset); rank 0 answers `search`/`retrieve`, the other ranks call `serve()`.  With
        `local_ids=True` the handle numbers its rows from 0 (rows_of[0] = their global rows; empty at the start) and
        the collection is FILLED through rank 0: index_chunks / add_rows / add_rows_synthetic, finalize and
        save_snapshot on rank 0 become collective operations the serving ranks take part in (shards.py)."""
        if not hasattr(self, "_native"):
            self._connect()
        handles = list(handles)
        if rows_of is None:
            rows_of, base = [], 0
            for h in handles:
                rows_of.append(np.arange(base, base + h.num_rows, dtype=np.int64))
                base += h.num_rows
        local = ShardSet(handles)
        local.rows_of = [np.asarray(r, dtype=np.int64) for r in rows_of]
        local._n = int(sum(len(r) for r in rows_of))
        if process_group is not None:
            import torch.distributed as dist
            from .shards import CollectiveShardSet
            if len(handles) != 1:
                raise ValueError("the torchrun form takes one shard handle per process")
            group = None if process_group is True else process_group
            self._main = CollectiveShardSet(local, first_row, dist, group, local_ids=local_ids)
        else:
            self._main = local
        self.devices = [h.device for h in handles]
        self.collections["semantic_index"] = ShardCollection(self, "semantic_index", "dense", self._main,
                                                             self.semantic_dim, "COSINE")
        if handles[0].sparse_dim:
            self.collections["sparse_index"] = ShardCollection(self, "sparse_index", "sparse", self._main,
                                                               self.sparse_dim, "IP")
        self._synthetic_rows = int(synthetic_rows)
        self._dev_filters = None
        self._dev_masks.clear()
        self._mask_cache.clear()

    def serve(self):
        """Ranks > 0 of the torchrun form: answer rank 0's searches until it calls stop_workers()."""
        self._main.serve()

    def stop_workers(self):
        self._main.stop_workers()

    # ------------------------------------------------------------------ host columns
    def _columns(self) -> Dict[str, np.ndarray]:
        """numpy view of the filterable columns for the HOST restatement of the filter semantics (filters.evaluate): CPU
        tests and collections without a device; string fields are materialised as unicode arrays here."""
        if hasattr(self._cols, "filter_columns"):
            return self._cols.filter_columns()
        return {k: np.asarray(v) for k, v in self._cols.items()}

    @property
    def num_rows(self) -> int:
        return self._synthetic_rows or len(self._cols["id"])

    @staticmethod
    def synthetic_id(row: int) -> str:
        """String id of a bulk-ingested row: doc{row//10}::{row%10}::{row:08x}."""
        return f"doc{row // 10}::{row % 10}::{row:08x}"

    def _filters_on_device(self):
        """The device evaluator of this collection's filter expressions, or None (no GPU shard: CPU-only unit tests)."""
        main = getattr(self, "_main", None)
        if main is None or not hasattr(main, "first") or not hasattr(main.first, "_h"):
            return None
        if self._dev_filters is None:
            from .device_filters import DeviceFilters
            self._dev_filters = DeviceFilters(None if self._synthetic_rows else self._cols, main.first.device)
        return self._dev_filters

    def _global_device_mask(self, expr: Optional[str]):
        """Packed mask over GLOBAL rows of `expr` + tombstones as a CUDA tensor (None = all rows), evaluated on the
        device and kept per (expression, rows, tombstone epoch)."""
        n = self.num_rows
        dead = self._deleted is not None and bool(self._deleted[:n].any())
        if not expr and not dead:
            return None
        key = ("global", expr, n, self._delete_epoch)
        with self._mask_lock:   # searches of one request run in different threads and ask for the same new mask at once
            hit = self._dev_masks.get(key)
            if hit is None:
                if len(self._dev_masks) >= 32:
                    self._dev_masks.pop(next(iter(self._dev_masks)))
                hit = self._dev_masks[key] = self._filters_on_device().evaluate(expr, n, self._deleted if dead else None)[0]
        return hit

    def _row_mask(self, expr: Optional[str]) -> Optional[np.ndarray]:
        """Boolean filter over global rows for a filter expression plus the tombstones, or None for "all rows" — the
        host-side view of the mask (sharded / collective searches cut it up per shard).  With a GPU shard the
        predicate is evaluated on the device and read back once; without one (CPU-only tests) filters.evaluate runs on
        the host columns.  Kept per (expression, row count, tombstone epoch)."""
        n = self.num_rows
        if self._synthetic_rows and not expr and not (self._deleted is not None and self._deleted[:n].any()):
            return None
        key = (expr, n, self._delete_epoch)
        if key in self._mask_cache:
            return self._mask_cache[key]
        keep = None
        if self._filters_on_device() is not None:
            m = self._global_device_mask(expr)
            if m is not None:
                keep = np.unpackbits(m.cpu().numpy(), bitorder="little")[:n].astype(bool)
        else:
            if expr:
                keep = self._host_predicate(expr, n)
            if self._deleted is not None and self._deleted[:n].any():
                alive = np.ones(n, dtype=bool)
                alive[:self._deleted.shape[0]] = ~self._deleted[:n]
                keep = alive if keep is None else (keep & alive)
        if len(self._mask_cache) >= 64:
            self._mask_cache.pop(next(iter(self._mask_cache)))
        self._mask_cache[key] = keep
        return keep

    def _host_predicate(self, expr: str, n: int) -> np.ndarray:
        """filters.evaluate over the host columns (collections without a device: CPU tests, oracle-backed shards)."""
        if self._synthetic_rows:
            # bulk-ingested rows carry no payload columns; what their synthetic id encodes can still be filtered on:
            # chunk_index = row % 10 (synthetic_id), derived on the fly
            fields = {f for f, _, _ in _filters.parse(expr)}
            if fields - {"chunk_index"}:
                raise ValueError("this shard was bulk-ingested without payload columns: only chunk_index (= row % 10) "
                                 f"can be filtered on, not {sorted(fields - {'chunk_index'})}")
            return _filters.evaluate(expr, {"chunk_index": np.arange(n, dtype=np.int64) % 10}, n)
        return _filters.evaluate(expr, self._columns(), n)

    def _tombstone(self, expr: str):
        n = self.num_rows
        dev = self._filters_on_device()
        if dev is not None:
            hit = np.unpackbits(dev.evaluate(expr, n)[0].cpu().numpy(), bitorder="little")[:n].astype(bool)
        else:
            hit = self._host_predicate(expr, n)
        if self._deleted is None or self._deleted.shape[0] < n:
            grown = np.zeros(n, dtype=bool)
            if self._deleted is not None:
                grown[:self._deleted.shape[0]] = self._deleted
            self._deleted = grown
        self._deleted[:n] |= hit
        self._delete_epoch += 1

    # ------------------------------------------------------------------ ingest
    async def index_chunks(self, chunks: List["Chunk"], domain: Optional[str] = None) -> Dict[str, Any]:
        """Embed and append chunks to every collection (reference indexing.py:264-437);
        same summary keys, same best-effort handling of sparse/domain failures.

        When the embedding generator offers `encode_to_device` (encoders.SentenceEncoder) the dense rows never leave
        the GPU: the encoder's output tensor is cast to the shard's storage type on the device and re-tiled into the
        shard by hr_add_dense_raw_dev on the encoder's stream; the host semantic cache is not
        consulted or filled on that path (it holds host arrays).  Sparse payloads of all chunks of a call travel as ONE
        CSR, and every collection is flushed once per call.  `timing_ms` in the summary is an addition to the
        reference's keys: encoder / append / flush wall time of this call."""
        import time as _time
        summary = {"total_chunks": len(chunks), "indexed_semantic": 0, "indexed_sparse": 0, "indexed_domain": 0,
                   "errors": []}
        timing = {"encode": 0.0, "append": 0.0, "flush": 0.0}
        use_sparse = "sparse_index" in self.collections and os.getenv("ENABLE_SPARSE", "1") == "1"
        gen = self.embedding_generator
        on_device = gen is not None and hasattr(gen, "encode_to_device") and self._main is not None and \
            hasattr(self._main, "handles")
        t0 = _time.perf_counter()
        texts = [c.text for c in chunks]
        dense_dev = dom_dev = None
        if on_device and chunks:
            dense_dev = await self._run_encoder(gen.encode_to_device, texts)
            if dense_dev.dim() != 2 or dense_dev.shape[1] != self.semantic_dim:
                raise ValueError(f"semantic embedding has dim {tuple(dense_dev.shape)[1:]}, expected {self.semantic_dim}")
            if "domain_index" in self.collections and hasattr(gen, "encode_domain_to_device"):
                dom_dev = await self._run_encoder(gen.encode_domain_to_device, texts, domain)
                if dom_dev.shape[1] != self.domain_dim:
                    raise ValueError(f"domain embedding has dim {dom_dev.shape[1]}, expected {self.domain_dim}")
            dense_vecs = None
        else:
            dense_vecs = await self._generate_semantic_embeddings_batch(texts)
        # sparse payloads of the whole call from ONE hook call when the generator offers the batch form (SentenceEncoder:
        # hr_bm25_encode_dev on its GPU; the reference calls encode_sparse once per chunk, indexing.py:379-404); a batch the
        # hook or the checks below refuse goes chunk by chunk, where a bad payload costs only its own row
        sparse_batch = None
        if use_sparse and chunks and gen is not None and hasattr(gen, "encode_sparse_csr"):
            try:
                sparse_batch = self._checked_sparse_csr(await self._run_encoder(gen.encode_sparse_csr, texts), len(chunks))
            except Exception as e:
                logger.warning("batched sparse payloads failed (%s); encoding chunk by chunk", e)
        rows_dense, rows_domain, sp_ptr, sp_idx, sp_val, kept, kept_idx = [], [], [0], [], [], [], []
        n_sparse_ok = 0
        for i, chunk in enumerate(chunks):
            try:
                dense = None
                if dense_vecs is not None:
                    dense = np.asarray(dense_vecs[i], dtype=np.float32).reshape(-1)
                    if dense.shape[0] != self.semantic_dim:
                        raise ValueError(f"semantic embedding has dim {dense.shape[0]}, expected {self.semantic_dim}")
                sp = None
                if sparse_batch is not None:
                    bp, bi, bv = sparse_batch
                    sp = (bi[bp[i]:bp[i + 1]], bv[bp[i]:bp[i + 1]])
                    n_sparse_ok += 1
                elif use_sparse:
                    try:
                        sp = self._clean_sparse_payload(await self._generate_sparse_embedding(chunk.text, role="document"))
                        n_sparse_ok += 1
                    except Exception as e:
                        summary["errors"].append({"chunk_id": chunk.metadata.chunk_id,
                                                  "error": f"sparse_embedding_failed: {e}"})
                dom = None
                if "domain_index" in self.collections and dom_dev is None:
                    dom = np.asarray(await self._generate_domain_embedding(chunk.text, domain),
                                     dtype=np.float32).reshape(-1)
                    if dom.shape[0] != self.domain_dim:
                        raise ValueError(f"domain embedding has dim {dom.shape[0]}, expected {self.domain_dim}")
            except Exception as e:
                summary["errors"].append({"chunk_id": chunk.metadata.chunk_id, "error": str(e)})
                continue
            if dense is not None:
                rows_dense.append(dense)
            if dom is not None:
                rows_domain.append(dom)
            if use_sparse:  # a row per chunk keeps row numbers aligned across collections
                si, sv = sp if sp is not None else (np.zeros(0, np.int32), np.zeros(0, np.float32))
                sp_idx.append(si)
                sp_val.append(sv)
                sp_ptr.append(sp_ptr[-1] + len(si))
            kept.append(chunk)
            kept_idx.append(i)
        timing["encode"] = (_time.perf_counter() - t0) * 1e3
        summary["timing_ms"] = timing
        if not kept:
            return summary

        def device_rows(t, store_half: bool):
            """The kept rows of an encoder output, in the shard's storage type, still on the device."""
            import torch
            if len(kept_idx) != t.shape[0]:
                t = t.index_select(0, torch.as_tensor(kept_idx, device=t.device))
            return t.to(torch.float16 if store_half else torch.float32).contiguous()

        try:
            if "semantic_index" not in self.collections:
                raise KeyError("semantic_index")
            sparse_csr = None
            if use_sparse:
                sparse_csr = (np.asarray(sp_ptr, np.int64), np.concatenate(sp_idx) if sp_idx else np.zeros(0, np.int32),
                              np.concatenate(sp_val) if sp_val else np.zeros(0, np.float32))
            store_half = self.dtype in ("float16", "fp16", "f16")
            t1 = _time.perf_counter()
            # dense rows, sparse rows and payload columns of a chunk share one row number: the three are appended
            # together, and whatever fails afterwards is padded rather than left short
            dense_rows = device_rows(dense_dev, store_half) if dense_dev is not None else np.stack(rows_dense)
            try:
                _, _, sparse_err = await asyncio.to_thread(self._main.add, dense_rows, sparse_csr)
            except PartialAppend as pa:
                # some shards took their piece before another refused: the payload of exactly those rows goes in, so
                # that later batches keep the row numbers the shards gave them
                self._append_payload(kept[: pa.end - pa.base])
                summary["indexed_semantic"] = pa.end - pa.base
                raise pa.cause
            self._append_payload(kept)
            summary["indexed_semantic"] = len(kept)
            if use_sparse:
                if sparse_err is None:
                    summary["indexed_sparse"] = n_sparse_ok
                else:
                    logger.warning("Sparse insert failed; continuing without sparse index: %s", sparse_err)
                    summary["errors"].append({"insert_sparse_error": str(sparse_err)})
            if "domain_index" in self.collections:
                before = self._domain.num_rows
                try:
                    dom_rows = device_rows(dom_dev, store_half) if dom_dev is not None else np.stack(rows_domain)
                    await asyncio.to_thread(self._domain.add, dom_rows)
                    summary["indexed_domain"] = len(kept)
                except Exception as e:  # zero rows never match (cosine 0): the domain collection stays row-aligned
                    missing = len(kept) - (self._domain.num_rows - before)   # pad only what did not go in
                    if missing > 0:
                        await asyncio.to_thread(self._domain.add, np.zeros((missing, self.domain_dim), np.float32))
                    summary["errors"].append({"insert_domain_error": str(e)})
            timing["append"] = (_time.perf_counter() - t1) * 1e3
            t2 = _time.perf_counter()
            for coll in {id(c.handle): c for c in self.collections.values()}.values():
                await asyncio.to_thread(coll.flush)
            timing["flush"] = (_time.perf_counter() - t2) * 1e3
        except Exception as e:
            summary["errors"].append({"insert_error": str(e)})
        return summary

    def _append_payload(self, chunks):
        c = self._cols
        metas = [ch.metadata for ch in chunks]
        c["id"].extend(m.chunk_id for m in metas)
        c["doc_id"].extend(str(m.doc_id) for m in metas)
        c["content"].extend(ch.text[:65535] for ch in chunks)
        c["chunk_index"].extend([int(m.chunk_index) for m in metas])
        c["token_count"].extend([int(m.token_count) for m in metas])
        for name in FLOAT_COLUMNS:   # FLOAT fields hold float32 (reference schema, indexing.py:200-202)
            c[name].extend([getattr(m, name) for m in metas])
        c["timestamp"].extend(str(m.timestamp) for m in metas)
        c["metadata_json"].extend(str(m.to_dict())[:10000] for m in metas)
        self._mask_cache.clear()
        self._dev_masks.clear()

    def add_rows(self, dense: np.ndarray, sparse_csr=None, ids: Optional[Sequence[str]] = None,
                 contents: Optional[Sequence[str]] = None, **scalar_columns):
        """Bulk ingest of pre-computed vectors (benchmarks, snapshots): dense [n, dim]
        float16/float32, optional CSR triple (indptr, indices, values)."""
        if self._synthetic_rows:
            raise ValueError("shard is in synthetic-payload mode; use add_rows_synthetic")
        n = dense.shape[0]
        base = self.num_rows
        failed = None
        try:
            _, _, sparse_err = self._main.add(dense, sparse_csr if "sparse_index" in self.collections else None)
        except PartialAppend as pa:    # the shards numbered pa.end - pa.base rows: their payload goes in, then the caller hears
            n, failed, sparse_err = pa.end - pa.base, pa.cause, None
        c = self._cols
        defaults = {"doc_id": lambda r: f"doc{r // 10}", "chunk_index": lambda r: r % 10, "token_count": lambda r: 0,
                    "entropy": lambda r: 0.0, "redundancy": lambda r: 0.0, "domain_density": lambda r: 0.0,
                    "timestamp": lambda r: "", "metadata_json": lambda r: ""}
        step = 1 << 18   # bounded temporaries: the columns hold bytes and numbers, not Python objects
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            c["id"].extend(ids[lo:hi] if ids is not None else (self.synthetic_id(base + r) for r in range(lo, hi)))
            c["content"].extend(contents[lo:hi] if contents is not None else ("" for _ in range(lo, hi)))
            for name, fn in defaults.items():
                given = scalar_columns.get(name)
                c[name].extend(given[lo:hi] if given is not None else [fn(base + r) for r in range(lo, hi)])
        self._mask_cache.clear()
        self._dev_masks.clear()
        if failed is not None:
            raise failed
        if sparse_err is not None:  # the rows are in (with empty sparse rows); the caller still hears about it
            raise sparse_err

    def add_rows_synthetic(self, dense: np.ndarray, sparse_csr=None):
        """Bulk ingest without host payload columns: ids/metadata are derived from the row
        number on demand (10M-row benchmarks would otherwise hold GBs of Python strings)."""
        if len(self._cols["id"]):
            raise ValueError("shard already holds payload columns")
        n, failed = dense.shape[0], None
        try:
            _, _, sparse_err = self._main.add(dense, sparse_csr if "sparse_index" in self.collections else None)
        except PartialAppend as pa:
            n, failed, sparse_err = pa.end - pa.base, pa.cause, None
        self._synthetic_rows += n
        self._dev_filters = None
        self._dev_masks.clear()
        self._mask_cache.clear()
        if failed is not None:
            raise failed
        if sparse_err is not None:
            raise sparse_err

    def finalize(self):
        for coll in {id(c.handle): c for c in self.collections.values()}.values():
            coll.flush()

    # ------------------------------------------------------------------ snapshot
    def save_snapshot(self, directory: str) -> None:
        """Write the shard(s) and the host payload columns to `directory` (resume without re-embedding)."""
        os.makedirs(directory, exist_ok=True)
        self.finalize()
        self._main.save(lambda s: os.path.join(directory, f"main.{s}.hbmrag"))
        if self._domain is not None:
            self._domain.save(lambda s: os.path.join(directory, f"domain.{s}.hbmrag"))
        cols = {k: (v.as_str_array() if hasattr(v, "as_str_array") else v.array()) for k, v in self._cols.items()}
        deleted = self._deleted if self._deleted is not None else np.zeros(0, dtype=bool)
        maps = {f"rows_main_{s}": r for s, r in enumerate(self._main.row_maps())}
        if self._domain is not None:
            maps.update({f"rows_domain_{s}": r for s, r in enumerate(self._domain.row_maps())})
        np.savez(os.path.join(directory, "payload.npz"), synthetic_rows=np.int64(self._synthetic_rows), deleted=deleted,
                 n_shards=np.int64(self._main.n_shards), **maps, **{f"col_{k}": v for k, v in cols.items()})

    def load_snapshot(self, directory: str) -> None:
        """Replace this manager's (empty) collections with the ones saved by `save_snapshot`; the manager must have
        been created with as many devices as the snapshot has shards."""
        nat = self._native
        store = nat.HR_F16 if self.dtype in ("float16", "fp16", "f16") else nat.HR_F32
        sparse_on = "sparse_index" in self.collections
        with np.load(os.path.join(directory, "payload.npz"), allow_pickle=False) as z:
            n_shards = int(z["n_shards"])
            if n_shards != len(self.devices):
                raise ValueError(f"snapshot has {n_shards} shards, this manager was created with {len(self.devices)} devices")
            main = [nat.ShardHandle.load(os.path.join(directory, f"main.{s}.hbmrag"), self.semantic_dim, store,
                                         nat.HR_METRIC_COSINE, self.sparse_dim if sparse_on else 0, d)
                    for s, d in enumerate(self.devices)]
            self._main.adopt(main, [z[f"rows_main_{s}"] for s in range(n_shards)])
            if self._domain is not None and os.path.exists(os.path.join(directory, "domain.0.hbmrag")):
                dom = [nat.ShardHandle.load(os.path.join(directory, f"domain.{s}.hbmrag"), self.domain_dim, store,
                                            nat.HR_METRIC_COSINE, 0, d) for s, d in enumerate(self.devices)]
                self._domain.adopt(dom, [z[f"rows_domain_{s}"] for s in range(n_shards)])
            self._synthetic_rows = int(z["synthetic_rows"])
            self._deleted = z["deleted"].copy() if z["deleted"].size else None
            self._cols = PayloadColumns()
            for k in self._cols:
                self._cols[k].extend(z[f"col_{k}"].tolist())
        self._dev_filters = None
        self._mask_cache.clear()
        self._dev_masks.clear()

    def load_snapshot_rank(self, directory: str, process_group=True, device: Optional[int] = None) -> None:
        """Resume the torchrun form from a snapshot rank 0 wrote with save_snapshot(): EVERY rank calls this — it loads
        its own shard file main.<rank>.hbmrag and its row map, rank 0 also the payload columns — then ranks > 0 serve()."""
        import torch.distributed as dist
        if not hasattr(self, "_native"):
            self._connect()
        nat = self._native
        group = None if process_group is True else process_group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        store = nat.HR_F16 if self.dtype in ("float16", "fp16", "f16") else nat.HR_F32
        with np.load(os.path.join(directory, "payload.npz"), allow_pickle=False) as z:
            if int(z["n_shards"]) != world:
                raise ValueError(f"snapshot has {int(z['n_shards'])} shards, the process group has {world} ranks")
            rows = z[f"rows_main_{rank}"].astype(np.int64)
            sparse_on = os.getenv("ENABLE_SPARSE", "1") == "1"
            dev = self.devices[0] if device is None else device
            h = nat.ShardHandle.load(os.path.join(directory, f"main.{rank}.hbmrag"), self.semantic_dim, store,
                                     nat.HR_METRIC_COSINE, self.sparse_dim if sparse_on else 0, dev)
            synthetic = int(z["synthetic_rows"])
            if rank == 0:
                self._deleted = z["deleted"].copy() if z["deleted"].size else None
                self._cols = PayloadColumns()
                for k in self._cols:
                    self._cols[k].extend(z[f"col_{k}"].tolist())
        if self._main is not None:
            self._main.close()
        self.attach_shards([h], rows_of=[rows], synthetic_rows=synthetic, process_group=process_group, local_ids=True)

    # ------------------------------------------------------------------ search
    @staticmethod
    def _as_sparse_payload(emb):
        if isinstance(emb, dict):
            idx = np.asarray(emb.get("indices", []), dtype=np.int32)
            val = np.asarray(emb.get("values", []), dtype=np.float32)
        elif hasattr(emb, "tocsr"):  # scipy sparse matrix (1, dim)
            m = emb.tocsr()
            idx, val = m.indices.astype(np.int32), m.data.astype(np.float32)
        else:
            raise ValueError("Sparse query embedding must be dict with indices/values or a scipy.sparse matrix")
        if idx.shape != val.shape:
            raise ValueError("sparse indices/values length mismatch")
        order = np.argsort(idx, kind="stable")
        return idx[order], val[order]

    def _checked_sparse_csr(self, csr, n: int):
        """A batch hook's CSR as (indptr int64, indices int32, values float32) if it is what the shard accepts as it stands —
        n rows, indices inside [0, sparse_dim) and strictly ascending within a row, finite values, rows of at most 60 000
        entries — else None (the per-chunk path then cleans or rejects payload by payload)."""
        ptr, idx, val = (np.asarray(csr[0], dtype=np.int64), np.asarray(csr[1], dtype=np.int32), np.asarray(csr[2], dtype=np.float32))
        if ptr.shape != (n + 1,) or ptr[0] != 0 or ptr[-1] != idx.shape[0] or idx.shape != val.shape or (np.diff(ptr) < 0).any():
            return None
        if idx.size:
            if idx.min() < 0 or idx.max() >= self.sparse_dim or not np.isfinite(val).all() or np.diff(ptr).max() > 60000:
                return None
            asc = np.ones(idx.size, dtype=bool)
            asc[1:] = idx[1:] > idx[:-1]
            asc[ptr[:-1][ptr[:-1] < idx.size]] = True          # the first entry of a row has no predecessor in its row
            if not asc.all():
                return None
        return ptr, idx, val

    def _clean_sparse_payload(self, emb):
        """A DOCUMENT's sparse payload, made acceptable to the shard or rejected here, chunk by chunk — a whole batch
        must not fail later because of one row: indices in range, values finite and within the fp16 posting range,
        duplicate indices merged by summing (what scipy's CSR arithmetic does with the reference's payload)."""
        idx, val = self._as_sparse_payload(emb)
        if idx.size:
            if idx[0] < 0 or idx[-1] >= self.sparse_dim:
                raise ValueError(f"sparse index out of range [0, {self.sparse_dim})")
            if not np.isfinite(val).all():
                raise ValueError("non-finite sparse value")
            if (idx[1:] == idx[:-1]).any():
                uniq, start = np.unique(idx, return_index=True)
                val = np.add.reduceat(val.astype(np.float64), start).astype(np.float32)
                idx = uniq.astype(np.int32)
            if np.abs(val).max() > 60000.0:
                raise ValueError("sparse weight exceeds the fp16 posting range (|w| <= 60000)")
        return idx, val

    def _format_hits(self, ids: np.ndarray, scores: np.ndarray) -> List[Dict[str, Any]]:
        c = self._cols
        out = []
        for row, score in zip(ids.tolist(), scores.tolist()):
            if row < 0:
                break
            if self._synthetic_rows:
                out.append({"id": self.synthetic_id(row), "content": "", "score": float(score),
                            "metadata": {"doc_id": f"doc{row // 10}", "chunk_index": row % 10, "entropy": 0.0,
                                         "redundancy": 0.0, "domain_density": 0.0, "timestamp": ""},
                            "_row": row})
                continue
            out.append({"id": c["id"][row], "content": c["content"][row], "score": float(score),
                        "metadata": {"doc_id": c["doc_id"][row], "chunk_index": c["chunk_index"][row],
                                     "entropy": c["entropy"][row], "redundancy": c["redundancy"][row],
                                     "domain_density": c["domain_density"][row], "timestamp": c["timestamp"][row]},
                        "_row": row})
        return out

    def _search_params(self, coll, search_params: Optional[Dict]) -> Dict:
        params = search_params or ({"metric_type": "IP"} if coll.kind == "sparse"
                                   else {"metric_type": "COSINE", "params": {"ef": 64}})
        metric = params.get("metric_type", coll.metric)
        if metric != coll.metric:
            raise ValueError(f"metric_type {metric} does not match collection {coll.name} ({coll.metric})")
        return params

    def _search_lists_blocking(self, query, collection_name: str, top_k: int, filters: Optional[str], params: Dict):
        """(row ids [k], scores [k]) of ONE query through the host forms (which escalate until the list is proven)."""
        coll = self.collections[collection_name]
        drop = float((params.get("params") or params).get("drop_ratio_search", 0.0))
        if getattr(coll.handle, "n_shards", 1) == 1 and hasattr(coll.handle, "handles") and self._filters_on_device() is not None:
            # one local GPU shard: the mask never leaves the device
            h = coll.handle.first
            d_mask = self._device_row_mask(filters, coll.kind)
            ptr = d_mask.data_ptr() if d_mask is not None else 0
            if coll.kind == "sparse":
                ids, sc = h.search_sparse([query], top_k, drop, None, ptr)
            elif hasattr(query, "is_cuda") and query.is_cuda and d_mask is None:
                ids, sc = self._search_dense_device(h, query, top_k)
            else:
                if hasattr(query, "detach"):
                    query = query.detach().cpu().numpy()
                ids, sc = h.search_dense(np.asarray(query, dtype=np.float32).reshape(1, -1), top_k, None, ptr)
            return ids[0], sc[0]
        mask = self._row_mask(filters)
        if coll.kind == "sparse":
            ids, sc = coll.handle.search_sparse([query], top_k, drop, mask)
        else:
            if hasattr(query, "detach"):
                query = query.detach().cpu().numpy()
            ids, sc = coll.handle.search_dense(np.asarray(query, dtype=np.float32).reshape(1, -1), top_k, mask)
        return ids[0], sc[0]

    def _search_blocking(self, query_embedding, collection_name: str, top_k: int, filters: Optional[str],
                         search_params: Optional[Dict]) -> List[Dict[str, Any]]:
        coll = self.collections[collection_name]
        params = self._search_params(coll, search_params)
        query = self._as_sparse_payload(query_embedding) if coll.kind == "sparse" else query_embedding
        return self._format_hits(*self._search_lists_blocking(query, collection_name, top_k, filters, params))

    @staticmethod
    def _search_dense_device(handle, q_dev, top_k: int):
        """Query already in HBM (device-resident embedding cache): device form, no upload; an unproven
        list (ties at the candidate cut) is redone through the host form."""
        import torch
        q = q_dev.reshape(1, -1).to(torch.float32).contiguous()
        ids = torch.empty((1, top_k), dtype=torch.int64, device=q.device)
        sc = torch.empty((1, top_k), dtype=torch.float32, device=q.device)
        flag = torch.zeros((1,), dtype=torch.int32, device=q.device)
        stream = torch.cuda.current_stream(q.device)
        handle.search_dense_dev(q.data_ptr(), 1, top_k, ids.data_ptr(), sc.data_ptr(), flag.data_ptr(), 0, stream.cuda_stream)
        stream.synchronize()
        if int(flag.item()) != 1:
            return handle.search_dense(q.cpu().numpy(), top_k)
        return ids.cpu().numpy(), sc.cpu().numpy()

    # ---- coalescing front (batching.py) -------------------------------------------------------------------------
    def _coalescer(self, coll):
        """The batching front, or None when this collection cannot use it (spread over several shard handles: those
        searches fan out per shard and merge on the host)."""
        if not self.coalesce:
            return None
        collective = hasattr(coll.handle, "round")          # torchrun form: rounds of one broadcast + one gather
        if not collective and (getattr(coll.handle, "n_shards", 1) != 1 or not hasattr(coll.handle, "handles")):
            return None
        if self._front is None:
            from .batching import SearchCoalescer
            self._front = SearchCoalescer(self)
        return self._front

    def _device_row_mask(self, expr: Optional[str], kind: str):
        """Packed row mask of a filter expression (+ tombstones) for a single-shard collection, as a uint8 CUDA tensor
        evaluated on the device (device_filters.py) and kept per (expression, rows, tombstone epoch): a batch of filtered
        searches uploads nothing.  None = all rows."""
        g = self._global_device_mask(expr)
        if g is None:
            return None
        n_local = self._main.first.num_sparse_rows if kind == "sparse" else self._main.first.num_rows
        if g.numel() * 8 < n_local:
            raise ValueError(f"row mask covers {g.numel() * 8} rows, the collection holds {n_local}")
        return g

    @staticmethod
    def _params_key(params: Dict) -> tuple:
        return tuple(sorted((k, v) for k, v in (params.get("params") or {}).items() if isinstance(v, (int, float, str, bool))))

    async def search(self, query_embedding, collection_name: str, top_k: int = 20, filters: Optional[str] = None,
                     search_params: Optional[Dict] = None) -> List[Dict[str, Any]]:
        if collection_name not in self.collections:
            raise ValueError(f"Collection {collection_name} not found")
        coll = self.collections[collection_name]
        front = self._coalescer(coll)
        try:
            if front is not None:
                params = self._search_params(coll, search_params)
                query = self._as_sparse_payload(query_embedding) if coll.kind == "sparse" else query_embedding
                fut = front.submit_async("sparse" if coll.kind == "sparse" else "dense",
                                         (collection_name, int(top_k), filters, self._params_key(params)), query)
                ids, sc = await asyncio.wait_for(fut, timeout=IndexingConstants.MILVUS_TIMEOUT_SECONDS)
                return self._format_hits(ids, sc)
            return await asyncio.wait_for(
                asyncio.to_thread(self._search_blocking, query_embedding, collection_name, top_k, filters,
                                  search_params),
                timeout=IndexingConstants.MILVUS_TIMEOUT_SECONDS)
        except asyncio.TimeoutError:
            logging.error("shard search timeout for collection %s", collection_name)
            raise Exception(f"Search timeout for collection {collection_name}")

    async def hybrid_search(self, dense_embedding, sparse_embedding, top_k: int, filters: Optional[str],
                            weights: Sequence[float], rrf_k: int = 60, semantic_params: Optional[Dict] = None,
                            sparse_params: Optional[Dict] = None):
        """The semantic search (2 x top_k), the sparse search (2 x top_k) and their rank fusion for ONE request in ONE
        round of the batching front: what `search` + `search` + `fuse_rank_lists_async` return, cut to the fused top_k.

        -> [(hit dict as `search` formats it, with "score" = the score in the list the payload comes from (the semantic
        list if the row is in it, else the sparse one), float64 fused score, method bit mask)] in fused order, or None
        when this manager cannot answer that way (no front, sharded collection, a list the device form could not prove
        exact, bad parameters ...) — the caller then takes the general path, which also owns the error behaviour."""
        sem, spa = self.collections.get("semantic_index"), self.collections.get("sparse_index")
        if sem is None or spa is None or sem.handle is not spa.handle or 2 * int(top_k) > HR_MAX_TOPK:
            return None
        front = self._coalescer(sem)
        if front is None or (front.collective and not getattr(self._main, "supports_hybrid_round", False)):
            return None        # (the torchrun form answers in one round when its shards allow it: shards.round_hybrid)
        try:
            self._search_params(sem, semantic_params)
            sp = self._search_params(spa, sparse_params)
            drop = float((sp.get("params") or sp).get("drop_ratio_search", 0.0))
            payload = self._as_sparse_payload(sparse_embedding)
        except Exception:
            return None
        if not len(payload[0]):
            return None
        w = list(weights) + [0.0] * (2 - len(weights))
        fut = front.submit_async("hybrid", (int(top_k), filters, drop, int(rrf_k)), (dense_embedding, payload, float(w[0]), float(w[1])))
        try:
            # no timer of its own: the caller (HybridRetriever.retrieve) already bounds the whole request with
            # RetrievalConstants.TIMEOUT_SECONDS, and a wait_for here is a task + a timer handle per request on the event
            # loop that serves every in-flight retrieve()
            res = await fut
        except Exception:
            return None
        if res is None:
            return None
        rows, fused_scores, methods, orig = res
        hits = self._format_hits(rows, orig)
        return list(zip(hits, fused_scores.tolist(), methods.tolist()))

    @staticmethod
    def _fuse_inputs(row_lists, id_lists, weights, rrf_k):
        row_to_id = {}
        for rows, ids in zip(row_lists, id_lists):
            for r, i in zip(rows, ids):
                row_to_id.setdefault(int(r), i)
        lists = tuple([np.asarray(r, dtype=np.int64) for r in row_lists] + [np.zeros(0, np.int64)] * (3 - len(row_lists)))
        w = list(weights) + [0.0] * (3 - len(weights))
        return row_to_id, lists, (float(w[0]), float(w[1]), float(w[2]), int(rrf_k))

    @staticmethod
    def _fuse_output(row_to_id, rows, scores, methods):
        return [(row_to_id[int(r)], float(s), [b for b in range(3) if (int(m) >> b) & 1])
                for r, s, m in zip(rows, scores, methods)]

    def _fuse_rows_blocking(self, lists, key):
        wa, wb, wc, rrf_k = key
        return self._main.first.fuse_rrf(lists[0], lists[1], lists[2], wa, wb, wc, rrf_k)

    def fuse_rank_lists(self, row_lists: Sequence[Sequence[int]], id_lists: Sequence[Sequence[Any]],
                        weights: Sequence[float], rrf_k: int = 60):
        """RRF on the device (csrc/fuse.h) over row numbers; returns
        [(id, float64 score, [list indices])] in fused order."""
        if self._main is None:
            raise RuntimeError("no shard handle")
        row_to_id, lists, key = self._fuse_inputs(row_lists, id_lists, weights, rrf_k)
        return self._fuse_output(row_to_id, *self._fuse_rows_blocking(lists, key))

    async def fuse_rank_lists_async(self, row_lists, id_lists, weights, rrf_k: int = 60):
        """The same fusion for a caller on the event loop: concurrent callers share one hr_fuse_rrf_dev launch."""
        if self._main is None:
            raise RuntimeError("no shard handle")
        row_to_id, lists, key = self._fuse_inputs(row_lists, id_lists, weights, rrf_k)
        front = self._coalescer(self.collections["semantic_index"]) if "semantic_index" in self.collections else None
        if front is None or front.collective or max(len(x) for x in lists) > 256:
            return self._fuse_output(row_to_id, *await asyncio.to_thread(self._fuse_rows_blocking, lists, key))
        rows, scores, methods = await asyncio.wait_for(front.submit_async("fuse", key, lists),
                                                       timeout=IndexingConstants.MILVUS_TIMEOUT_SECONDS)
        return self._fuse_output(row_to_id, rows, scores, methods)

    # ------------------------------------------------------------------ embeddings
    async def _run_encoder(self, fn, *args):
        if asyncio.iscoroutinefunction(fn):
            return await fn(*args)
        if getattr(self.embedding_generator, "run_inline", False):
            # the generator declares its synchronous hooks cheap (a lookup, a hash): no thread-pool hop per request (the
            # reference offloads every sync hook, indexing.py:610-620 — right for a model forward, 60 us of pure overhead
            # for a table)
            return fn(*args)
        return await asyncio.get_event_loop().run_in_executor(self.embedding_executor, lambda: fn(*args))

    async def _generate_semantic_embeddings_batch(self, texts: List[str]) -> List[np.ndarray]:
        cache = get_semantic_cache()
        out: List[Optional[np.ndarray]] = []
        missing: List[int] = []
        for i, t in enumerate(texts):
            hit = await cache.get(t)
            out.append(hit)
            if hit is None:
                missing.append(i)
        if missing and self.embedding_generator:
            gen = self.embedding_generator
            batch_fn = getattr(gen, "encode_semantic_batch", None)
            miss_texts = [texts[i] for i in missing]
            if batch_fn is not None:   # one batched forward instead of the reference's per-text loop (:584-587)
                vecs = await self._run_encoder(batch_fn, miss_texts)
            elif asyncio.iscoroutinefunction(gen.encode_semantic):
                vecs = [await gen.encode_semantic(t) for t in miss_texts]
            else:
                vecs = await asyncio.get_event_loop().run_in_executor(
                    self.embedding_executor, lambda: [gen.encode_semantic(t) for t in miss_texts])
            for i, v in zip(missing, vecs):
                await cache.put(texts[i], v)
                out[i] = v
        else:
            for i in missing:  # placeholder, uncached in the batch path as in the reference (:593-597)
                out[i] = np.random.randn(self.semantic_dim).astype(np.float32)
        return out  # type: ignore[return-value]

    def _device_query_embedding(self, text: str):
        """float32 [dim] CUDA tensor for `text` from the device-resident table (filled on a miss)."""
        from .embedding_cache import DeviceEmbeddingTable, EmbeddingCache
        if self._dev_cache is None:
            self._dev_cache = DeviceEmbeddingTable(self._dev_cache_slots, self.semantic_dim, f"cuda:{self.device}")
            self.device_cache_stats = {"hits": 0, "misses": 0}
        key = EmbeddingCache._materialize_key(text)
        vec = self._dev_cache.lookup(key)
        if vec is not None:
            self.device_cache_stats["hits"] += 1
            return vec
        self.device_cache_stats["misses"] += 1
        gen = self.embedding_generator
        if gen is not None and hasattr(gen, "encode_to_device"):
            fresh = gen.encode_to_device([text])[0]
        elif gen is not None:
            fresh = np.asarray(gen.encode_semantic(text), dtype=np.float32)
        else:
            fresh = np.random.randn(self.semantic_dim).astype(np.float32)
        return self._dev_cache.store(key, fresh)

    async def _generate_semantic_embedding(self, text: str):
        gen = self.embedding_generator
        if self._dev_cache_slots > 0 and self._main is not None and not (
                gen is not None and asyncio.iscoroutinefunction(gen.encode_semantic)):
            from .embedding_cache import DeviceEmbeddingTable, EmbeddingCache
            if self._dev_cache is None:
                self._dev_cache = DeviceEmbeddingTable(self._dev_cache_slots, self.semantic_dim, f"cuda:{self.device}")
                self.device_cache_stats = {"hits": 0, "misses": 0}
            key = EmbeddingCache._materialize_key(text)
            hit = self._dev_cache.lookup(key)
            if hit is not None:          # a hit is a dictionary lookup: no executor hop, no task
                self.device_cache_stats["hits"] += 1
                return hit
            front = self._encode_front(gen)
            if front is not None:        # the misses of a round share ONE encoder forward (batching.py::_enqueue_encode)
                self.device_cache_stats["misses"] += 1
                return await front.submit_async("encode", ("encode",), (key, text))
            return await asyncio.get_event_loop().run_in_executor(self.embedding_executor, self._device_query_embedding, text)
        front = self._encode_front(gen)
        if front is not None:
            from .embedding_cache import EmbeddingCache
            key = EmbeddingCache._materialize_key(text)

            async def compute_batched() -> np.ndarray:   # the host cache holds host arrays (reference indexing.py:601-627)
                row = await front.submit_async("encode", ("encode",), (key, text))
                return await asyncio.get_event_loop().run_in_executor(self.embedding_executor, lambda: row.cpu().numpy())
            return await get_semantic_cache().get_or_compute(text, compute_batched)

        async def compute() -> np.ndarray:
            if self.embedding_generator:
                return await self._run_encoder(self.embedding_generator.encode_semantic, text)
            return np.random.randn(self.semantic_dim).astype(np.float32)
        return await get_semantic_cache().get_or_compute(text, compute)

    def _encode_front(self, gen):
        """The batching front when it can encode queries for this manager (a generator with `encode_to_device` on a
        single-shard, single-process manager), else None."""
        if gen is None or not hasattr(gen, "encode_to_device") or not self.coalesce or "semantic_index" not in self.collections:
            return None
        front = self._coalescer(self.collections["semantic_index"])
        return None if front is None or front.collective else front

    async def _generate_sparse_embedding(self, text: str, role: str = "query"):
        gen = self.embedding_generator
        if gen:
            fn = gen.encode_sparse
            if role == "query" and hasattr(gen, "encode_sparse_query"):
                fn = gen.encode_sparse_query
            return await self._run_encoder(fn, text)
        nnz = min(100, self.sparse_dim)
        picks = np.random.choice(self.sparse_dim, size=nnz, replace=False)
        vals = np.abs(np.random.randn(nnz).astype(np.float32))
        order = np.argsort(picks)
        return {"indices": picks[order].tolist(), "values": vals[order].astype(float).tolist()}

    async def _generate_domain_embedding(self, text: str, domain: Optional[str] = None) -> np.ndarray:
        gen = self.embedding_generator
        if gen:
            if asyncio.iscoroutinefunction(gen.encode_domain):
                return await gen.encode_domain(text, domain or "")
            return await asyncio.get_event_loop().run_in_executor(self.embedding_executor,
                                                                  lambda: gen.encode_domain(text, domain))
        return np.random.randn(self.domain_dim).astype(np.float32)

    # ------------------------------------------------------------------ misc
    def get_collection_stats(self, collection_name: str) -> Dict[str, Any]:
        coll = self.collections.get(collection_name)
        if coll is None:
            return {}
        return {"name": collection_name, "num_entities": coll.num_entities, "schema": coll.schema,
                "indexes": coll.indexes}

    async def delete_by_filter(self, collection_name: str, expr: str):
        if collection_name in self.collections:
            self.collections[collection_name].delete(expr)

    async def close(self):
        if self._front is not None:
            await asyncio.to_thread(self._front.close)
            self._front = None
        self._dev_masks.clear()
        for h in (self._main, self._domain):
            if h is not None:
                try:
                    h.close()
                except Exception:
                    pass
        self._main = self._domain = None
        self.collections.clear()
        if hasattr(self, "embedding_executor"):
            self.embedding_executor.shutdown(wait=True)


HbmIndexManager = MilvusIndexManager
