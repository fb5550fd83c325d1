"""Coalescing of concurrent `MilvusIndexManager.search` calls into batched device searches.

The reference serves up to 64 in-flight `retrieve()` calls on one event loop (service.py:136,149); each of them
gathers its two or three `index_manager.search` calls (retrieval.py:293-306), and every such call is one RPC that the
Milvus server is free to batch with the others (indexing.py:503-506 runs it in a worker thread).  Here a search is a
scan of the whole shard, so the counterpart of the server's request queue is this front: calls that arrive while a
round of scans is in flight — or within a short window of the first call of a round — are packed per
(collection, top_k, filter expression, search params) into ONE `hr_search_dense_dev` / `hr_search_sparse_dev` batch
(and the rank fusions of the same callers into one `hr_fuse_rrf_dev`), run on the front's own stream with one
synchronisation per round, and the per-query lists are handed back to the awaiting coroutines.  Lists the device form
could not prove exact are redone through the host form, which widens the candidate set by itself.

A caller that has BOTH embeddings of a request in hand (HybridRetriever._retrieve_inner via
MilvusIndexManager.hybrid_search) submits one "hybrid" request instead of two searches and a fusion: the batch runs as
one `hr_search_hybrid_dev` (both scans, the fused finishing kernel) + one `hr_post_lists_dev` (RRF) — the engine's own
step — and the caller gets the fused top-k with the per-modality scores in ONE round instead of two.

One worker thread per manager; the event loop never blocks on the GPU.
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .engine import pack_sparse_queries


def _deliver(fut: Future, value) -> None:
    """Hand a result to a caller that may be gone: a request whose awaiting coroutine timed out or was cancelled has a
    cancelled Future (asyncio.wrap_future propagates it), and set_result on it raises InvalidStateError — which must
    never reach the other requests of the batch."""
    try:
        if not fut.done():
            fut.set_result(value)
    except Exception:   # InvalidStateError: cancelled between the check and the call
        pass


def _fail(fut: Future, exc: BaseException) -> None:
    try:
        if not fut.done():
            fut.set_exception(exc)
    except Exception:
        pass


def _set_many(items) -> None:
    """On the event loop: the answers of one round, in one callback."""
    for fut, ok, value in items:
        if not fut.done():          # a caller that timed out or was cancelled is simply skipped
            if ok:
                fut.set_result(value)
            else:
                fut.set_exception(value)


class _LoopFuture:
    """What a coroutine's request carries instead of a concurrent Future: the asyncio future it awaits.  Answers are
    collected in the front's outbox and handed to the loop with ONE call_soon_threadsafe per round (asyncio.wrap_future
    costs a concurrent future, a chained callback and a loop wake-up per request)."""
    __slots__ = ("loop", "afut", "outbox", "_done")

    def __init__(self, loop, afut, outbox):
        self.loop, self.afut, self.outbox, self._done = loop, afut, outbox, False

    def done(self) -> bool:
        return self._done or self.afut.cancelled()     # a benign cross-thread read: worst case one unread answer

    def set_running_or_notify_cancel(self) -> bool:
        return not self.afut.cancelled()

    def set_result(self, value) -> None:
        self._done = True
        self.outbox.setdefault(self.loop, []).append((self.afut, True, value))

    def set_exception(self, exc) -> None:
        self._done = True
        self.outbox.setdefault(self.loop, []).append((self.afut, False, exc))


@dataclass
class _Request:
    kind: str                 # "dense" | "sparse" | "fuse"
    key: Tuple                # requests with equal (kind, key) share a launch
    payload: Any
    future: Future = field(default_factory=Future)


class SearchCoalescer:
    """Batches requests of one MilvusIndexManager.  `max_batch` bounds the queries per launch (256 = the largest single
    dense pass); `window_s` is how long a round keeps waiting for the next request after the last one
    arrived (the sibling searches of one retrieve() are microseconds apart; under load the queue is never empty and the
    window never runs)."""

    def __init__(self, manager, max_batch: int = 256, window_s: float = 40e-6):
        self.mgr = manager
        self.max_batch = int(max_batch)
        self.window_s = float(window_s)
        self._q: "queue.SimpleQueue[Optional[_Request]]" = queue.SimpleQueue()
        self._thread: Optional[threading.Thread] = None
        self._lock = threading.Lock()
        self._closed = False
        # torchrun form (shards.CollectiveShardSet): a round of the front = one round of the shard set, i.e. ONE
        # broadcast + ONE gather for all the dense and sparse searches that share (top_k, filter, drop ratio)
        self.collective = hasattr(getattr(manager, "_main", None), "round")
        if self.collective:
            # a round of the torchrun form costs a broadcast + a gather over every rank whatever it carries: waiting a few
            # hundred microseconds for the other search of the same retrieve() (and for other callers) halves the collectives
            self.window_s = max(self.window_s, 250e-6)
        self.stats = {"rounds": 0, "requests": 0, "dense_launches": 0, "sparse_launches": 0, "fuse_launches": 0,
                      "hybrid_launches": 0, "encode_launches": 0, "encoded_texts": 0, "max_batch_seen": 0, "redone_unproven": 0,
                      "busy_s": 0.0}
        self._engines: Dict[Tuple, Any] = {}   # hybrid engines per (top_k, rrf_k)
        self._inflight: List[_Request] = []    # the requests of the round in progress (failed as a whole if the worker dies)
        self._outbox: Dict[Any, list] = {}     # event loop -> [(asyncio future, ok, value)] of the round in progress

    # ------------------------------------------------------------------ front
    def submit(self, kind: str, key: Tuple, payload: Any) -> Future:
        req = _Request(kind, key, payload)
        with self._lock:
            if self._closed:
                raise RuntimeError("search front is closed")
            if self._thread is None or not self._thread.is_alive():   # first request, or the worker died: start one
                self._thread = threading.Thread(target=self._guarded, name="search-coalescer", daemon=True)
                self._thread.start()
        self._q.put(req)
        return req.future

    def _guarded(self):
        """The worker loop with a last line of defence: whatever escapes a round (a HIP error from a stream
        synchronisation, a bug) fails every request that is still waiting — nobody hangs — and ends this thread; the
        next submit() starts a fresh one."""
        self._inflight: List[_Request] = []
        try:
            (self._run_collective if self.collective else self._run)()
        except BaseException as e:   # noqa: BLE001 - the callers must hear about it, whatever it was
            self.stats["worker_failures"] = self.stats.get("worker_failures", 0) + 1
            err = RuntimeError(f"search front worker failed: {type(e).__name__}: {e}")
            for r in self._inflight:
                _fail(r.future, err)
            while True:
                try:
                    r = self._q.get_nowait()
                except queue.Empty:
                    break
                if r is not None:
                    _fail(r.future, err)
            self._flush()
            with self._lock:
                if self._thread is threading.current_thread():
                    self._thread = None

    def submit_async(self, kind: str, key: Tuple, payload: Any):
        """submit() for a caller on an event loop: returns an asyncio future of that loop (await it directly)."""
        import asyncio
        loop = asyncio.get_running_loop()
        afut = loop.create_future()
        req = _Request(kind, key, payload, _LoopFuture(loop, afut, self._outbox))
        with self._lock:
            if self._closed:
                raise RuntimeError("search front is closed")
            if self._thread is None or not self._thread.is_alive():
                self._thread = threading.Thread(target=self._guarded, name="search-coalescer", daemon=True)
                self._thread.start()
        self._q.put(req)
        return afut

    def _flush(self):
        """Hand the collected answers to their event loops: one wake-up per loop and round."""
        if not self._outbox:
            return
        out = list(self._outbox.items())
        self._outbox.clear()
        for loop, items in out:
            try:
                loop.call_soon_threadsafe(_set_many, items)
            except RuntimeError:      # the loop is closed: nobody is waiting any more
                pass

    def close(self):
        with self._lock:
            self._closed = True
            t = self._thread
        if t is not None:
            self._q.put(None)
            t.join(timeout=30)

    # ------------------------------------------------------------------ worker
    def _collect(self) -> Optional[List[_Request]]:
        """The requests of one round: everything already queued (what piled up while the previous round ran), plus what
        arrives within `window_s` of the LAST arrival, at most 4 windows after the first request.  The wait is a blocking
        get with a timeout — it releases the GIL, so the event loop can go on submitting, and returns the moment a request
        arrives; a lone retrieve() pays the expiry of one window per round (search round, fusion round)."""
        first = self._q.get()
        if first is None:
            return None
        reqs = [first]
        t_first = time.perf_counter()
        while len(reqs) < 4 * self.max_batch:
            try:
                r = self._q.get_nowait()
            except queue.Empty:
                if time.perf_counter() - t_first >= 4 * self.window_s:
                    break
                try:
                    r = self._q.get(timeout=self.window_s)
                except queue.Empty:
                    break
            if r is None:
                self._q.put(None)  # close() arrived behind real work: finish this round first
                break
            reqs.append(r)
        # claim the futures: a request cancelled while it waited in the queue is dropped here, and one that is claimed
        # can no longer be cancelled under the worker's feet (Future.cancel() fails on a running future)
        live = [r for r in reqs if r.future.set_running_or_notify_cancel()]
        self.stats["cancelled_before_launch"] = self.stats.get("cancelled_before_launch", 0) + len(reqs) - len(live)
        self._inflight = live
        return live

    def _run(self):
        import torch
        dev = torch.device("cuda", self.mgr.device)
        # one stream per kind of work: the dense and the sparse searches of a round run side by side (a lone retrieve()
        # overlaps its two scans, as its two worker threads did before the front existed)
        streams = {k: torch.cuda.Stream(dev) for k in ("dense", "sparse", "fuse", "hybrid", "encode")}
        while True:
            reqs = self._collect()
            if reqs is None:
                return
            if not reqs:
                continue
            t0 = time.perf_counter()
            groups: Dict[Tuple, List[_Request]] = {}
            for r in reqs:
                groups.setdefault((r.kind, r.key), []).append(r)
            self.stats["rounds"] += 1
            self.stats["requests"] += len(reqs)
            launched, used = [], set()
            for (kind, key), rs in groups.items():
                stream = streams[kind]
                used.add(kind)
                with torch.cuda.stream(stream):
                    step = min(self.max_batch, 128) if kind == "hybrid" else self.max_batch
                    for c0 in range(0, len(rs), step):
                        chunk = rs[c0:c0 + step]
                        self.stats["max_batch_seen"] = max(self.stats["max_batch_seen"], len(chunk))
                        try:
                            launched.append((kind, key, chunk, getattr(self, "_enqueue_" + kind)(torch, dev, stream, key, chunk)))
                        except Exception as e:   # a bad batch must not take the other groups down: one by one
                            launched.append((kind, key, chunk, e))
            for kind in used:
                streams[kind].synchronize()
            for kind, key, chunk, state in launched:
                if isinstance(state, Exception):
                    self._one_by_one(kind, key, chunk)
                    continue
                try:
                    getattr(self, "_scatter_" + kind)(key, chunk, state)
                except Exception as e:  # never leave a caller waiting
                    for r in chunk:
                        _fail(r.future, e)
            self._inflight = []
            self._flush()
            self.stats["busy_s"] += time.perf_counter() - t0

    def _run_collective(self):
        """Rounds of the torchrun form: the dense and the sparse searches of the callers that share (top_k, filter
        expression, drop ratio) travel in one packet; other combinations take a round of their own."""
        cs = self.mgr._main
        if getattr(cs.dev, "type", "cpu") == "cuda":   # the current device is a per-thread setting: the collectives of this
            import torch                                # thread must run on the rank's GPU
            torch.cuda.set_device(cs.dev)
        while True:
            reqs = self._collect()
            if reqs is None:
                return
            if not reqs:
                continue
            t0 = time.perf_counter()
            self.stats["rounds"] += 1
            self.stats["requests"] += len(reqs)
            groups: Dict[Tuple, Dict[str, List[_Request]]] = {}
            hybrid: Dict[Tuple, List[_Request]] = {}
            for r in reqs:
                if r.kind == "hybrid":
                    hybrid.setdefault(r.key, []).append(r)
                    continue
                if r.kind == "fuse":
                    try:
                        _deliver(r.future, self.mgr._fuse_rows_blocking(r.payload, r.key))
                    except Exception as e:
                        _fail(r.future, e)
                    continue
                coll_name, top_k, expr, params_key = r.key
                if self.mgr.collections[coll_name].handle is not cs:
                    # a collection that is NOT spread over the ranks (the local domain shard): its searches have nothing
                    # to do with the shard set's rounds — the blocking single-shard path answers them
                    try:
                        _deliver(r.future, self.mgr._search_lists_blocking(r.payload, coll_name, top_k, expr, dict(params_key)))
                    except Exception as e:
                        _fail(r.future, e)
                    continue
                drop = float(dict(params_key).get("drop_ratio_search", 0.0)) if r.kind == "sparse" else None
                groups.setdefault((top_k, expr), {}).setdefault((r.kind, drop), []).append(r)
            for (top_k, expr, drop, rrf_k), rs in hybrid.items():
                # both searches and the fusion of these requests as ONE collective round on the device (shards.round_hybrid)
                for c0 in range(0, len(rs), 64):
                    chunk = rs[c0:c0 + 64]
                    try:
                        keep = self.mgr._row_mask(expr)
                        q = np.stack([np.asarray(r.payload[0].detach().cpu().numpy() if hasattr(r.payload[0], "detach") else r.payload[0],
                                                 dtype=np.float32).reshape(-1) for r in chunk])
                        res = cs.round_hybrid(q, [r.payload[1] for r in chunk], top_k, drop, rrf_k,
                                              np.array([[r.payload[2], r.payload[3]] for r in chunk], dtype=np.float64), keep)
                        self.stats["hybrid_launches"] += 1
                        self.stats["max_batch_seen"] = max(self.stats["max_batch_seen"], len(chunk))
                        self._scatter_hybrid_lists(chunk, res["fused_ids"], res["fused_scores"], res["fused_methods"], res["fused_n"],
                                                   res["list_ids"], res["list_scores"], res["proven"])
                    except Exception as e:
                        for r in chunk:
                            _fail(r.future, e)
            for (top_k, expr), by_kind in groups.items():
                dense = next((v for (kind, _), v in by_kind.items() if kind == "dense"), [])
                sparse_sets = [(d, v) for (kind, d), v in by_kind.items() if kind == "sparse"] or [(0.0, [])]
                for n_round, (drop, sparse) in enumerate(sparse_sets):
                    dn = dense if n_round == 0 else []
                    for c0 in range(0, max(len(dn), len(sparse), 1), 64):
                        d_chunk, s_chunk = dn[c0:c0 + 64], sparse[c0:c0 + 64]
                        if not d_chunk and not s_chunk:
                            continue
                        try:
                            keep = self.mgr._row_mask(expr)
                            q = np.stack([np.asarray(r.payload.detach().cpu().numpy() if hasattr(r.payload, "detach") else r.payload,
                                                     dtype=np.float32).reshape(-1) for r in d_chunk]) if d_chunk else None
                            res = cs.round(q, [r.payload for r in s_chunk] if s_chunk else None, top_k, drop or 0.0, keep)
                            self.stats["dense_launches"] += 1 if d_chunk else 0
                            self.stats["sparse_launches"] += 1 if s_chunk else 0
                            self.stats["max_batch_seen"] = max(self.stats["max_batch_seen"], len(d_chunk), len(s_chunk))
                            for chunk, lists in ((d_chunk, res[0]), (s_chunk, res[1])):
                                for i, r in enumerate(chunk):
                                    _deliver(r.future, (lists[0][i], lists[1][i]))
                        except Exception as e:
                            for r in d_chunk + s_chunk:
                                _fail(r.future, e)
            self._inflight = []
            self._flush()
            self.stats["busy_s"] += time.perf_counter() - t0

    def _one_by_one(self, kind: str, key: Tuple, chunk: List[_Request]):
        """Fallback when a batched launch was refused: each request through the blocking single-query path, so that
        only the request that is actually at fault fails."""
        for r in chunk:
            try:
                if kind == "fuse":
                    _deliver(r.future, self.mgr._fuse_rows_blocking(r.payload, key))
                elif kind == "hybrid":
                    _deliver(r.future, None)   # the caller falls back to two searches + a fusion
                elif kind == "encode":
                    _deliver(r.future, self.mgr.embedding_generator.encode_to_device([r.payload[1]])[0])
                else:
                    coll_name, top_k, expr, params_key = key
                    _deliver(r.future, self.mgr._search_lists_blocking(r.payload, coll_name, top_k, expr, dict(params_key)))
            except Exception as e:
                _fail(r.future, e)

    # ------------------------------------------------------------------ dense
    def _enqueue_dense(self, torch, dev, stream, key, chunk):
        coll_name, top_k, expr, _ = key
        handle = self.mgr.collections[coll_name].handle.first
        B = len(chunk)
        on_dev = [hasattr(r.payload, "is_cuda") and r.payload.is_cuda for r in chunk]
        if all(on_dev):
            q = torch.stack([r.payload.reshape(-1).to(torch.float32) for r in chunk]).contiguous()
        else:
            host = np.stack([np.asarray(r.payload.detach().cpu().numpy() if hasattr(r.payload, "detach") else r.payload,
                                        dtype=np.float32).reshape(-1) for r in chunk])
            if host.shape[1] != handle.dim:
                raise ValueError(f"query dim {host.shape[1]} != shard dim {handle.dim}")
            q = torch.from_numpy(host).to(dev)
        ids = torch.empty((B, top_k), dtype=torch.int64, device=dev)
        sc = torch.empty((B, top_k), dtype=torch.float32, device=dev)
        fl = torch.zeros((B,), dtype=torch.int32, device=dev)
        mask = self.mgr._device_row_mask(expr, "dense")
        handle.search_dense_dev(q.data_ptr(), B, top_k, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(),
                                mask.data_ptr() if mask is not None else 0, stream.cuda_stream)
        self.stats["dense_launches"] += 1
        return {"ids": ids, "sc": sc, "fl": fl, "keep": (q, mask)}

    def _scatter_dense(self, key, chunk, st):
        self._scatter_lists(key, chunk, st)

    # ------------------------------------------------------------------ sparse
    def _enqueue_sparse(self, torch, dev, stream, key, chunk):
        coll_name, top_k, expr, params_key = key
        handle = self.mgr.collections[coll_name].handle.first
        drop = float(dict(params_key).get("drop_ratio_search", 0.0))
        ptr, idx, val, max_nnz = pack_sparse_queries([r.payload for r in chunk], drop, handle.sparse_dim)
        B = len(chunk)
        d_ptr, d_idx, d_val = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev)
        ids = torch.empty((B, top_k), dtype=torch.int64, device=dev)
        sc = torch.empty((B, top_k), dtype=torch.float32, device=dev)
        fl = torch.zeros((B,), dtype=torch.int32, device=dev)
        mask = self.mgr._device_row_mask(expr, "sparse")
        handle.search_sparse_dev(d_ptr.data_ptr(), d_idx.data_ptr() if idx.size else 0, d_val.data_ptr() if idx.size else 0, B,
                                 int(idx.shape[0]), int(max_nnz), top_k, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(),
                                 mask.data_ptr() if mask is not None else 0, stream.cuda_stream)
        self.stats["sparse_launches"] += 1
        return {"ids": ids, "sc": sc, "fl": fl, "keep": (d_ptr, d_idx, d_val, mask)}

    def _scatter_sparse(self, key, chunk, st):
        self._scatter_lists(key, chunk, st)

    def _scatter_lists(self, key, chunk, st):
        coll_name, top_k, expr, params_key = key
        ids, sc, fl = st["ids"].cpu().numpy(), st["sc"].cpu().numpy(), st["fl"].cpu().numpy()
        for i, r in enumerate(chunk):
            try:
                if fl[i] != 1:  # ties at the candidate cut: the host form widens the candidate set until the proof holds
                    self.stats["redone_unproven"] += 1
                    _deliver(r.future, self.mgr._search_lists_blocking(r.payload, coll_name, top_k, expr, dict(params_key)))
                else:
                    _deliver(r.future, (ids[i], sc[i]))
            except Exception as e:
                _fail(r.future, e)

    # ------------------------------------------------------------------ hybrid (both searches + RRF of a request)
    def _enqueue_hybrid(self, torch, dev, stream, key, chunk):
        from .engine import EngineConfig, HybridSearchEngine
        top_k, expr, drop, rrf_k = key
        handle = self.mgr.collections["semantic_index"].handle.first
        ekey = (top_k, rrf_k)
        eng = self._engines.get(ekey)
        if eng is None:
            if len(self._engines) >= 16:
                self._engines.clear()
            eng = self._engines[ekey] = HybridSearchEngine(
                handle, EngineConfig(top_k=top_k, rrf_k=rrf_k, enable_reranking=False), device=str(dev))
        B = len(chunk)
        dense = [r.payload[0] for r in chunk]
        # the fusion weights are per REQUEST (a weight_adapter may pick them per query, reference retrieval.py:251-262):
        # they travel as a [B, 3] operand of the post kernel, so requests with different weights still share the round
        wq = torch.from_numpy(np.array([[r.payload[2], r.payload[3], 0.0] for r in chunk], dtype=np.float64)).to(dev)
        if all(hasattr(q, "is_cuda") and q.is_cuda for q in dense):
            q = torch.stack([x.reshape(-1).to(torch.float32) for x in dense]).contiguous()
        else:
            host = np.stack([np.asarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32).reshape(-1)
                             for x in dense])
            q = torch.from_numpy(host).to(dev)
        if q.shape[1] != handle.dim:
            raise ValueError(f"query dim {q.shape[1]} != shard dim {handle.dim}")
        ptr, idx, val, max_nnz = pack_sparse_queries([r.payload[1] for r in chunk], drop, handle.sparse_dim)
        if not idx.size:
            raise ValueError("no sparse terms in the batch")   # -> one by one through the general path
        d_sparse = (torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), int(max_nnz))
        mask = self.mgr._device_row_mask(expr, "dense")
        b = eng.search(q, d_sparse, rowmask=mask, weights=wq)
        self.stats["hybrid_launches"] += 1
        self.stats["dense_launches"] += 1
        self.stats["sparse_launches"] += 1
        # the engine reuses ONE buffer set per batch size: another group (or the next chunk of this one) with the same B
        # is enqueued before this round is read back, so the results are copied out here, in stream order
        out = {k: b[k].clone() for k in ("fused_ids", "fused_scores", "fused_methods", "fused_n", "ids", "scores", "flags")}
        return {"b": out, "keep": (q, d_sparse, mask, wq)}

    def _scatter_hybrid(self, key, chunk, st):
        b = st["b"]
        fi, fs = b["fused_ids"].cpu().numpy(), b["fused_scores"].cpu().numpy()
        fm, fn = b["fused_methods"].cpu().numpy(), b["fused_n"].cpu().numpy()
        ids, sc, fl = b["ids"].cpu().numpy(), b["scores"].cpu().numpy(), b["flags"].cpu().numpy()
        self._scatter_hybrid_lists(chunk, fi, fs, fm, fn, ids, sc, fl.min(axis=0) == 1)

    def _scatter_hybrid_lists(self, chunk, fi, fs, fm, fn, ids, sc, proven):
        # score of a fused row in the list its payload comes from: the dense list if the row is in it, else the sparse one
        # (HybridRetriever._assemble_fused); rows are unique within a list
        in_d = fi[:, :, None] == ids[0][:, None, :]
        in_s = fi[:, :, None] == ids[1][:, None, :]
        orig = np.where(in_d.any(axis=2), np.take_along_axis(sc[0], in_d.argmax(axis=2), axis=1),
                        np.take_along_axis(sc[1], in_s.argmax(axis=2), axis=1))
        for i, r in enumerate(chunk):
            if not proven[i]:   # ties at a candidate cut: the general path redoes the searches through the host forms
                self.stats["redone_unproven"] += 1
                _deliver(r.future, None)
                continue
            n = int(fn[i])
            _deliver(r.future, (fi[i, :n].copy(), fs[i, :n].copy(), fm[i, :n].copy(), orig[i, :n].copy()))

    # ------------------------------------------------------------------ query encoder
    def _enqueue_encode(self, torch, dev, stream, key, chunk):
        """The query texts of a round's cache misses in ONE forward of the sentence encoder (the reference embeds each
        request alone, indexing.py:601-627; its batch form loops per text, :580-587): payload = (cache key, text).  The rows
        go into the manager's device-resident embedding table (when it has one) and are handed to the callers as device
        tensors — the search round that follows reads them in place, no host hop."""
        gen = self.mgr.embedding_generator
        texts: Dict[str, str] = {}
        for r in chunk:
            texts.setdefault(r.payload[0], r.payload[1])
        vecs = gen.encode_to_device(list(texts.values()), batch_size=len(texts))
        table = getattr(self.mgr, "_dev_cache", None)
        rows = {k: (table.store(k, vecs[i]) if table is not None else vecs[i]) for i, k in enumerate(texts)}
        self.stats["encode_launches"] += 1
        self.stats["encoded_texts"] += len(texts)
        return {"rows": rows, "keep": vecs}

    def _scatter_encode(self, key, chunk, st):
        for r in chunk:
            _deliver(r.future, st["rows"][r.payload[0]])

    # ------------------------------------------------------------------ fuse
    def _enqueue_fuse(self, torch, dev, stream, key, chunk):
        from . import _native as nat
        wa, wb, wc, rrf_k = key
        B = len(chunk)
        ka = max(1, max(len(r.payload[0]) for r in chunk))
        kb = max(len(r.payload[1]) for r in chunk)
        kc = max(len(r.payload[2]) for r in chunk)
        host = np.full((B, ka + kb + kc), -1, dtype=np.int64)
        for i, r in enumerate(chunk):
            a, b, c = r.payload
            host[i, :len(a)] = a
            host[i, ka:ka + len(b)] = b
            host[i, ka + kb:ka + kb + len(c)] = c
        d = torch.from_numpy(host).to(dev)
        la = d[:, :ka].contiguous()
        lb = d[:, ka:ka + kb].contiguous() if kb else None
        lc = d[:, ka + kb:].contiguous() if kc else None
        total = ka + kb + kc
        oi = torch.empty((B, total), dtype=torch.int64, device=dev)
        os_ = torch.empty((B, total), dtype=torch.float64, device=dev)
        om = torch.empty((B, total), dtype=torch.int32, device=dev)
        on = torch.empty((B,), dtype=torch.int32, device=dev)
        nat.fuse_rrf_dev(la.data_ptr(), ka, lb.data_ptr() if kb else 0, kb, lc.data_ptr() if kc else 0, kc, B, wa, wb, wc,
                         rrf_k, total, oi.data_ptr(), os_.data_ptr(), om.data_ptr(), on.data_ptr(), stream.cuda_stream)
        self.stats["fuse_launches"] += 1
        return {"oi": oi, "os": os_, "om": om, "on": on, "keep": (la, lb, lc)}

    def _scatter_fuse(self, key, chunk, st):
        oi, os_, om, on = st["oi"].cpu().numpy(), st["os"].cpu().numpy(), st["om"].cpu().numpy(), st["on"].cpu().numpy()
        for i, r in enumerate(chunk):
            n = int(on[i])
            _deliver(r.future, (oi[i, :n].copy(), os_[i, :n].copy(), om[i, :n].copy()))
