"""Batched, device-resident hybrid search over a row-sharded corpus.

This is the throughput path behind `retrieve()` for B concurrent queries (the
reference serves up to 64 in-flight retrieve() calls, service.py:137,149): one
pass of  dense top-k' + sparse top-k' -> [all-gather over RCCL/xGMI + merge] ->
RRF -> learned-ranker rerank,  every step a HIP kernel of libhbmrag enqueued on
the caller's stream; PyTorch only provides the device buffers, the stream and
`torch.distributed` (backend "nccl" = RCCL).

Sharding: rank r of W owns rows [r*ceil(N/W), ...) — its shard
handle was created with that row offset, so local lists already carry global
row ids.  Per batch each rank runs its local searches, then ONE all-gather of a
packed buffer (both modalities' ids + scores: 2*B*k'*12 bytes per rank — a
latency-bound exchange, far below per-link xGMI bandwidth), then every rank
merges the W lists with the same (score desc, id asc) rule, so all ranks hold
identical fused results without a second collective.

Semantics mirror HybridRetriever._retrieve_inner/_fuse_results/rerank
(reference retrieval.py:249-339, :421-491, :518-563) with profile "default":
k' = 2*top_k per modality, weights dense/sparse, RRF k = 60, fused[:top_k],
then rerank to rerank_top_k with LearnedRanker's linear score.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat


def shard_range(n_rows: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous row range of `rank`: ceil(n/world) rows each, rounded up to `align`."""
    per = -(-n_rows // world)
    per = -(-per // align) * align
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def pack_sparse_queries(queries: Sequence[Tuple[Sequence[int], Sequence[float]]], drop_ratio: float = 0.0,
                        sparse_dim: Optional[int] = None):
    """Host prep of a sparse query batch for the device form: apply
    drop_ratio_search (smallest |value| first; among equals the later entry),
    sort by index, build CSR.  Returns (indptr int64, idx int32, val float32, max_nnz).
    Indices must be distinct, non-negative and (when sparse_dim is given) below sparse_dim — what
    hr_search_sparse checks for the host form."""
    ptr = [0]
    idx_parts, val_parts = [], []
    max_nnz = 0
    for qi, qv in queries:
        qi = np.asarray(qi, dtype=np.int32)
        qv = np.asarray(qv, dtype=np.float32)
        if qi.shape != qv.shape:
            raise ValueError("sparse query indices/values length mismatch")
        if qi.size and (qi.min() < 0 or (sparse_dim is not None and qi.max() >= sparse_dim)):
            raise ValueError(f"sparse query index out of range [0, {sparse_dim})")
        if np.unique(qi).size != qi.size:
            raise ValueError("duplicate sparse query index")
        n_drop = int(np.floor(drop_ratio * len(qi)))
        if n_drop:
            order = np.lexsort((-np.arange(len(qi)), np.abs(qv)))  # |v| asc, later entry first
            keep = np.sort(order[n_drop:])
            qi, qv = qi[keep], qv[keep]
        order = np.argsort(qi, kind="stable")
        idx_parts.append(qi[order])
        val_parts.append(qv[order])
        ptr.append(ptr[-1] + len(qi))
        max_nnz = max(max_nnz, len(qi))
    idx = np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.int32)
    val = np.concatenate(val_parts) if val_parts else np.zeros(0, np.float32)
    return np.asarray(ptr, np.int64), idx.astype(np.int32), val.astype(np.float32), max_nnz


@dataclass(frozen=True)
class ListPack:
    """Byte layout of one rank's slot in the all-gather buffer: the ids of every
    modality ([n_mod][B][kp] int64), their scores ([n_mod][B][kp] fp32), then the per-list
    "proven exact" flags ([n_mod][B] int32) — so that after the ONE exchange every rank also knows
    which lists some rank could not prove, and all ranks take the same repair decision.
    Backend-agnostic (torch uint8 tensors on any device) so the gloo CPU tests
    exercise exactly the offsets and strides the HIP merge kernel is given."""
    n_mod: int
    B: int
    kp: int

    @property
    def id_bytes(self) -> int:
        return self.n_mod * self.B * self.kp * 8

    @property
    def score_bytes(self) -> int:
        return self.n_mod * self.B * self.kp * 4

    @property
    def nbytes(self) -> int:
        raw = self.id_bytes + self.score_bytes + self.n_mod * self.B * 4
        return -(-raw // 8) * 8  # slots stay 8-byte aligned inside the gathered buffer

    def views(self, pack):
        """(ids int64 [n_mod,B,kp], scores float32 [n_mod,B,kp]) views of a 1-D uint8 tensor."""
        import torch
        ids = pack[: self.id_bytes].view(torch.int64).view(self.n_mod, self.B, self.kp)
        scores = pack[self.id_bytes: self.id_bytes + self.score_bytes].view(torch.float32).view(self.n_mod, self.B, self.kp)
        return ids, scores

    def flags_view(self, pack):
        """int32 [n_mod, B] view of the flags of one slot (1-D uint8 tensor) or of every slot ([world, nbytes])."""
        import torch
        lo = self.id_bytes + self.score_bytes
        hi = lo + self.n_mod * self.B * 4
        if pack.dim() == 1:
            return pack[lo:hi].view(torch.int32).view(self.n_mod, self.B)
        return pack[:, lo:hi].contiguous().view(torch.int32).view(pack.shape[0], self.n_mod, self.B)

    def merge_args(self, modality: int):
        """(score byte offset, id byte offset, score stride [floats], id stride [int64s]) of one
        modality inside a gathered [world, nbytes] buffer."""
        return (self.id_bytes + modality * self.B * self.kp * 4, modality * self.B * self.kp * 8,
                self.nbytes // 4, self.nbytes // 8)


def exchange_lists(pack, world: int, dist=None, group=None, out=None):
    """All-gather every rank's packed lists: returns a [world, nbytes] uint8 tensor
    (one collective per query batch; RCCL on GPUs, gloo in the CPU tests)."""
    import torch
    if out is None:
        out = torch.empty((world, pack.numel()), dtype=torch.uint8, device=pack.device)
    if pack.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU): gloo has no device all-gather, stage through the host
        host = torch.empty((world, pack.numel()), dtype=torch.uint8)
        dist.all_gather_into_tensor(host.view(-1), pack.cpu(), group=group)
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out.view(-1), pack, group=group)
    return out


@dataclass
class EngineConfig:
    top_k: int = 20
    rerank_top_k: int = 5
    dense_weight: float = 0.7
    sparse_weight: float = 0.3
    rrf_k: int = 60
    enable_reranking: bool = True
    base_weight: float = 1.0      # LearnedRankerConfig defaults (ranker.py:27-30)
    method_bonus: float = 0.1
    recency_weight: float = 0.0
    use_sparse: bool = True
    domain_weight: float = 0.2    # weight of the optional domain list (reference retrieval.py:455-468)


class HybridSearchEngine:
    def __init__(self, handle: "nat.ShardHandle", config: Optional[EngineConfig] = None, process_group=None,
                 device: Optional[str] = None, stream=None, domain_handle: "Optional[nat.ShardHandle]" = None,
                 simulate_ranks: int = 0):
        import torch
        self.torch = torch
        self.h = handle
        # optional third modality (reference _search_domain, retrieval.py:397-419): a second dense shard over the same
        # rows (sharded the same way when the corpus is), searched with k = top_k (not 2k) and fused with weight 0.2
        self.hd = domain_handle
        self.cfg = config or EngineConfig()
        self.group = process_group
        self.dist = None
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.dist = torch.distributed
            self.world = self.dist.get_world_size(process_group)
            self.rank = self.dist.get_rank(process_group)
        # simulate_ranks = W > 1 on ONE process: the post-exchange work of a W-rank step (merge of W lists per
        # modality) is put on the finishing stream without a collective — a projection aid for benchmarks
        self.simulate_ranks = int(simulate_ranks) if self.world == 1 and simulate_ranks and simulate_ranks > 1 else 0
        self.n_lists = self.simulate_ranks or self.world
        self.device = torch.device(device or f"cuda:{handle.device}")
        # Optional private stream: several engines on different streams keep several query batches
        # in flight, so the latency-bound tail of one batch (select / refine / exchange / fuse)
        # overlaps the bandwidth-bound scans of the next.
        self.stream = stream
        self._bufs: Dict[int, dict] = {}

    # ------------------------------------------------------------------ buffers
    def _buffers(self, B: int) -> dict:
        """Per-batch-size buffers.  Lists live in ONE packed tensor (the rank's slot of the all-gather): modality 0 =
        dense, 1 = sparse (if used), last = domain (if the engine has a domain shard; its lists hold top_k entries in
        kp-wide rows, -1 padded)."""
        b = self._bufs.get(B)
        if b is not None:
            return b
        t, dev, cfg = self.torch, self.device, self.cfg
        kp = 2 * cfg.top_k
        n_main = 2 if cfg.use_sparse else 1
        n_mod = n_main + (1 if self.hd is not None else 0)
        layout = ListPack(n_mod, B, kp)
        pack = t.zeros(layout.nbytes, dtype=t.uint8, device=dev)
        ids_v, scores_v = layout.views(pack)
        ids_v.fill_(-1)
        b = {
            "kp": kp, "n_mod": n_mod, "n_main": n_main, "pack": pack, "layout": layout,
            "ids": ids_v, "scores": scores_v, "flags": layout.flags_view(pack),
            "fused_ids": t.empty((B, cfg.top_k), dtype=t.int64, device=dev),
            "fused_scores": t.empty((B, cfg.top_k), dtype=t.float64, device=dev),
            "fused_methods": t.empty((B, cfg.top_k), dtype=t.int32, device=dev),
            "fused_n": t.empty((B,), dtype=t.int32, device=dev),
            "rr_ids": t.empty((B, cfg.rerank_top_k), dtype=t.int64, device=dev),
            "rr_scores": t.empty((B, cfg.rerank_top_k), dtype=t.float64, device=dev),
            "rr_orig": t.empty((B, cfg.rerank_top_k), dtype=t.float64, device=dev),
            "use_domain": False,
        }
        if self.hd is not None:  # compact [B, top_k] lists: what the search writes and what the fusion reads
            b["dom_ids"] = t.full((B, cfg.top_k), -1, dtype=t.int64, device=dev)
            b["dom_scores"] = t.zeros((B, cfg.top_k), dtype=t.float32, device=dev)
            b["dom_flags"] = t.ones((B,), dtype=t.int32, device=dev)
        if self.n_lists > 1:
            b["gathered"] = t.empty((self.n_lists, layout.nbytes), dtype=t.uint8, device=dev)
            b["agg_flags_buf"] = t.ones((n_mod, B), dtype=t.int32, device=dev)
            if self.simulate_ranks:  # copy r of the local lists carries ids shifted by r * 2^40
                b["sim_shift"] = (t.arange(self.n_lists, dtype=t.int64, device=dev) << 40).view(-1, 1)
            b["m_ids"] = t.empty((n_main, B, kp), dtype=t.int64, device=dev)
            b["m_scores"] = t.empty((n_main, B, kp), dtype=t.float32, device=dev)
            if self.hd is not None:
                b["m_dom_ids"] = t.empty((B, cfg.top_k), dtype=t.int64, device=dev)
                b["m_dom_scores"] = t.empty((B, cfg.top_k), dtype=t.float32, device=dev)
        self._bufs[B] = b
        return b

    def _search_domain(self, b: dict, domain_q, B: int, stream: int):
        """Third list: top_k of the domain shard into the compact buffers, mirrored into the pack for the exchange."""
        cfg = self.cfg
        b["use_domain"] = domain_q is not None
        if self.hd is None:
            return
        m = b["n_mod"] - 1
        if domain_q is None:  # nothing to exchange for this batch: an empty, proven list
            b["ids"][m].fill_(-1)
            b["flags"][m].fill_(1)
            return
        self.hd.search_dense_dev(domain_q.data_ptr(), B, cfg.top_k, b["dom_ids"].data_ptr(), b["dom_scores"].data_ptr(),
                                 b["dom_flags"].data_ptr(), 0, stream)
        b["flags"][m].copy_(b["dom_flags"])
        if self.n_lists > 1:
            b["ids"][m, :, :cfg.top_k].copy_(b["dom_ids"])
            b["scores"][m, :, :cfg.top_k].copy_(b["dom_scores"])

    # ------------------------------------------------------------------ one batch
    def search(self, q, sparse=None, domain_q=None, rowmask=None, weights=None) -> dict:
        """q: float32 [B, dim] device tensor.  sparse: (indptr int64[B+1], idx int32, val float32, max_nnz)
        device tensors from `upload_sparse`.  domain_q: float32 [B, domain_dim] device tensor for the optional
        domain shard (engine built with domain_handle).  rowmask: packed uint8 device tensor (a filter expression's
        rows, device_filters.py) applied to the dense and the sparse search.  Asynchronous on the current stream;
        returns the buffer dict (fused_* and rr_* tensors are the results; they are reused by the next call)."""
        if self.stream is not None and self.torch.cuda.current_stream(self.device) != self.stream:
            with self.torch.cuda.stream(self.stream):
                return self.search(q, sparse, domain_q, rowmask, weights)
        if domain_q is not None and self.hd is None:
            raise ValueError("domain queries need an engine built with domain_handle")
        t, cfg = self.torch, self.cfg
        B = q.shape[0]
        b = self._buffers(B)
        kp = b["kp"]
        stream = t.cuda.current_stream(self.device).cuda_stream
        d_mask = rowmask.data_ptr() if rowmask is not None else 0
        if cfg.use_sparse:
            indptr, idx, val, max_nnz = sparse
            # one call: dense scan alone on this stream, sparse chain overlapping the dense tail
            self.h.search_hybrid_dev(q.data_ptr(), indptr.data_ptr(), idx.data_ptr(), val.data_ptr(), B,
                                     int(idx.shape[0]), int(max_nnz), kp, b["ids"].data_ptr(), b["scores"].data_ptr(),
                                     b["flags"].data_ptr(), d_mask, stream)
        else:
            self.h.search_dense_dev(q.data_ptr(), B, kp, b["ids"][0].data_ptr(), b["scores"][0].data_ptr(),
                                    b["flags"][0].data_ptr(), d_mask, stream)
        self._search_domain(b, domain_q, B, stream)
        if weights is not None and (weights.dtype != t.float64 or tuple(weights.shape) != (B, 3) or not weights.is_contiguous()):
            raise ValueError("weights must be a contiguous float64 [B, 3] device tensor")
        b["w_query"] = weights          # kept alive until the next call reuses the buffer set
        return self._post_lists(b, B, stream)

    def _post_args(self, b: dict, B: int, n_lists: int, gathered) -> "nat.PostArgs":
        """hr_post_args of one buffer set: built once per (buffer set, use of the domain list) and reused."""
        key = ("post_args", n_lists, b["use_domain"])
        a = b.get(key)
        if a is not None:
            return a
        cfg, kp, lay = self.cfg, b["kp"], b["layout"]
        a = nat.PostArgs()
        mods = [(0, 0, kp)]                                   # (fusion slot, modality in the pack, entries fused)
        if cfg.use_sparse:
            mods.append((1, 1, kp))
        if b["use_domain"]:
            mods.append((2, b["n_mod"] - 1, cfg.top_k))
        for slot, m, k_fuse in mods:
            a.k_fuse[slot] = k_fuse
            if n_lists > 1:
                sc_off, id_off, sc_stride, id_stride = lay.merge_args(m)
                a.ids[slot], a.scores[slot], a.k_in[slot] = gathered.data_ptr() + id_off, gathered.data_ptr() + sc_off, kp
                a.id_stride, a.score_stride = id_stride, sc_stride
                mi, ms = (b["m_dom_ids"], b["m_dom_scores"]) if slot == 2 else (b["m_ids"][m], b["m_scores"][m])
                a.merged_ids[slot], a.merged_scores[slot] = mi.data_ptr(), ms.data_ptr()
            elif slot == 2:
                a.ids[slot], a.k_in[slot] = b["dom_ids"].data_ptr(), cfg.top_k
            else:
                a.ids[slot], a.k_in[slot] = b["ids"][m].data_ptr(), kp
        a.n_lists, a.rrf_k, a.top_k = n_lists, cfg.rrf_k, cfg.top_k
        a.w[0], a.w[1], a.w[2] = cfg.dense_weight, cfg.sparse_weight, cfg.domain_weight
        a.fused_ids, a.fused_scores = b["fused_ids"].data_ptr(), b["fused_scores"].data_ptr()
        a.fused_methods, a.fused_n = b["fused_methods"].data_ptr(), b["fused_n"].data_ptr()
        a.rerank = 1 if cfg.enable_reranking else 0
        a.base_w, a.method_bonus, a.recency_w, a.k_out = cfg.base_weight, cfg.method_bonus, cfg.recency_weight, cfg.rerank_top_k
        a.rr_ids, a.rr_scores, a.rr_orig = b["rr_ids"].data_ptr(), b["rr_scores"].data_ptr(), b["rr_orig"].data_ptr()
        if n_lists > 1:  # the flags travelled with the lists: their minimum over the ranks comes out of the same launch
            a.flags = gathered.data_ptr() + lay.id_bytes + lay.score_bytes
            a.flag_stride, a.n_flag_rows = lay.nbytes // 4, lay.n_mod * lay.B
            a.agg_flags = b["agg_flags_buf"].data_ptr()
        b[key] = a
        return a

    def _exchange(self, b: dict):
        """The per-rank lists of this batch from every rank: one all-gather (or, for `simulate_ranks`, a device copy
        that stands in for it: the local lists replicated with their ids shifted per copy, so that the merge does the
        work of a real W-rank step on one GPU — timing only, the merged lists mean nothing)."""
        if self.simulate_ranks:
            g = b["gathered"]
            g.copy_(b["pack"].unsqueeze(0).expand(self.n_lists, -1))
            ids_all = g[:, : b["layout"].id_bytes].view(self.torch.int64).view(self.n_lists, -1)
            ids_all += b["sim_shift"] * (ids_all >= 0)
            return g
        return exchange_lists(b["pack"], self.world, self.dist, self.group, out=b["gathered"])

    def _post_lists(self, b: dict, B: int, stream: int) -> dict:
        """Everything after the per-shard lists exist: [exchange] -> ONE launch for merge + RRF + rerank."""
        ids, scores = b["ids"], b["scores"]
        g = None
        if self.n_lists > 1:
            g = self._exchange(b)
            ids, scores = b["m_ids"], b["m_scores"]
            # a list is proven iff every rank proved its part: identical on all ranks, so they all take the same
            # repair decision (resolve_inexact); the minimum over the ranks is taken by the post kernel
            b["agg_flags"] = b["agg_flags_buf"]
        else:
            b["agg_flags"] = b["flags"]
        a = self._post_args(b, B, self.n_lists, g)
        wq = b.get("w_query")
        a.w_query = wq.data_ptr() if wq is not None else None
        nat.post_lists_dev(a, B, stream)
        b["list_ids"], b["list_scores"] = ids, scores
        return b

    def resolve_inexact(self, b: dict, q_host: np.ndarray, sparse_host=None, drop_ratio: float = 0.0,
                        domain_q_host: Optional[np.ndarray] = None) -> int:
        """The device forms report, per (modality, query), whether the list is PROVEN exact.  For the rare ones
        that are not (ties at the candidate cut), redo those queries through the host forms — which widen the
        candidate set until the proof holds — patch the lists and redo exchange / fusion / rerank.  `sparse_host`
        holds the queries as given to pack_sparse_queries (before the drop).  Call after synchronising the batch.

        On a sharded corpus EVERY rank calls this with the same arguments: the exchanged flags (minimum over ranks,
        the same on every rank) decide which queries are redone, each rank repairs the lists ITS shard could not
        prove, and the exchange is repeated — collectively, and only if some list was unproven somewhere.
        Returns the number of (modality, query) lists this rank redid."""
        t = self.torch
        agg = b["agg_flags"].cpu().numpy()
        if agg.min() == 1:
            return 0
        own = b["flags"].cpu().numpy()
        redone = 0
        B = own.shape[1]
        for m in range(b["n_mod"]):
            bad = np.nonzero((agg[m] == 0) & (own[m] == 0))[0]
            if not len(bad):
                continue
            if m == 0:
                ids, sc = self.h.search_dense(np.ascontiguousarray(q_host[bad]), b["kp"])
            elif m < b["n_main"]:
                ids, sc = self.h.search_sparse([sparse_host[i] for i in bad], b["kp"], drop_ratio)
            else:
                ids, sc = self.hd.search_dense(np.ascontiguousarray(domain_q_host[bad]), self.cfg.top_k)
                sel = t.from_numpy(bad).to(self.device)
                b["dom_ids"].index_copy_(0, sel, t.from_numpy(ids).to(self.device))
                b["dom_scores"].index_copy_(0, sel, t.from_numpy(sc).to(self.device))
                pad = b["kp"] - self.cfg.top_k
                ids = np.pad(ids, ((0, 0), (0, pad)), constant_values=-1)
                sc = np.pad(sc, ((0, 0), (0, pad)))
            sel = t.from_numpy(bad).to(self.device)
            b["ids"][m].index_copy_(0, sel, t.from_numpy(ids).to(self.device))
            b["scores"][m].index_copy_(0, sel, t.from_numpy(sc).to(self.device))
            b["flags"][m].index_fill_(0, sel, 1)
            redone += len(bad)
        self._post_lists(b, B, t.cuda.current_stream(self.device).cuda_stream)
        return redone

    def upload_sparse(self, packed):
        t = self.torch
        indptr, idx, val, max_nnz = packed
        return (t.from_numpy(indptr).to(self.device), t.from_numpy(idx).to(self.device),
                t.from_numpy(val).to(self.device), max_nnz)


class PipelinedSearchEngine(HybridSearchEngine):
    """Keeps `depth` query batches in flight on two HIP streams.

    The bandwidth-bound scans of consecutive batches run back to back on the HEAVY stream
    (hr_hybrid_scan_dev, one workspace slot per batch in flight); everything that follows a batch's
    scans — candidate select, canonical refine, top-k, the RCCL exchange, merge, RRF, rerank — runs
    on the LIGHT stream and so hides behind the next batch's scans.  Scans never overlap each
    other, which keeps the event-timed scan (roofline) meaningful.  Results of `submit` are valid
    once the light stream has passed the returned `done` event (or after `synchronize()`).
    Requires the sparse modality (the two-phase C entry points are hybrid).  The optional domain
    list (engine built with domain_handle) is searched on the heavy stream right behind the batch's scans.
    """

    def __init__(self, handle, config: Optional[EngineConfig] = None, process_group=None, device: Optional[str] = None,
                 depth: int = 2, domain_handle=None, simulate_ranks: int = 0, light_cus: int = 0,
                 prep_stream: bool = False):
        super().__init__(handle, config, process_group, device, domain_handle=domain_handle,
                         simulate_ranks=simulate_ranks)
        if not self.cfg.use_sparse:
            raise ValueError("PipelinedSearchEngine needs the sparse modality")
        if not 1 <= depth <= 4:
            raise ValueError("depth must be 1..4 (HR_MAX_SLOTS)")
        t = self.torch
        self.depth = depth
        # The light stream is created with HIGH priority.  HIP multiplexes streams onto a few hardware queues
        # (round-robin in creation order), and two streams that land on the same queue execute in enqueue order:
        # batch i's finish work would then sit between scan(i) and scan(i+1) instead of beside scan(i+1)
        # (seen as 4.8-4.95 ms/step instead of 4.45-4.65 whenever other streams had been created first, e.g. by the
        # host-form searches of bench.py's parity gate).  Queues are per priority level, so a high-priority light
        # stream can never share one with the normal-priority heavy stream.
        # (Swapping the priorities — scans high, finishing work normal — measures the same: 0.611 against 0.618 ms per
        # step on a rank-sized shard, 3.83 against 3.77 ms at 10M rows.)
        #
        # prep_stream=True gives the query preparation of every batch (fragment-order queries, |q|^2, the sparse queries'
        # fixed-point scale: a handful of short, nearly empty launches) a third stream, so that the heavy stream carries
        # the scans only.  Measured on a rank-sized shard it buys nothing (0.605 against 0.613 ms per step with the
        # five-launch chain, 0.60 - 0.69 against 0.59 with the fused finishing kernel: a third queue contending for the
        # same compute units costs what the ~25 us of prep launches on the heavy stream cost), so it is off by default.
        #
        # light_cus > 0 confines the finishing + prep streams to the first `light_cus` compute units and the scans to
        # the rest (hr_stream_create CU masks; the scans' persistent grids are sized to their share).
        self._raw_streams = []
        self.light_cus = int(light_cus)
        if self.light_cus > 0:
            n_cu = t.cuda.get_device_properties(self.device).multi_processor_count
            if not 0 < self.light_cus < n_cu:
                raise ValueError(f"light_cus must be in (0, {n_cu})")
            di = self.device.index or 0
            low, high = nat.cu_mask_words(n_cu, 0, self.light_cus), nat.cu_mask_words(n_cu, self.light_cus, n_cu)
            self._raw_streams = [nat.stream_create(di, 0, high), nat.stream_create(di, 0, low), nat.stream_create(di, 0, low)]
            self.heavy, self.light, self.prep = (t.cuda.ExternalStream(r, device=self.device) for r in self._raw_streams)
            for hh in (self.h, self.hd):
                if hh is not None:
                    hh.set_scan_cus(n_cu - self.light_cus)
        else:
            self.heavy = t.cuda.Stream(self.device)
            self.light = t.cuda.Stream(self.device, priority=-1)
            self.prep = t.cuda.Stream(self.device, priority=-1) if prep_stream else None
        if not prep_stream:
            self.prep = None
        self._slot_bufs = [dict() for _ in range(depth)]
        self._n = 0
        # optional callable(buffers) run on the light stream after fusion/rerank of every batch, before the
        # batch is marked done (e.g. a cross-encoder forward over the fused candidates)
        self.post_hook = None
        # post_hook_exclusive: the hook is compute-bound work that gains nothing from sharing the chip with the HBM-bound scans
        # (a cross-encoder forward beside the scans of the next batches: both slower, the step no shorter — DESIGN section 5).
        # The hook of batch i is then enqueued when batch i + 1 is submitted: it waits for scan(i + 1), and scan(i + 2) waits
        # for it, so a step is  forward + max(scans, finishing)  instead of everything sharing the chip.
        self.post_hook_exclusive = False
        self._pending_hook = None       # the batch whose hook has not been enqueued yet
        self._hook_ev = None            # recorded behind the hook enqueued last

    def _run_hook(self, b):
        """Enqueue the post hook of batch b on the light stream (the caller has made the stream wait for what it must)."""
        t = self.torch
        with t.cuda.stream(self.light):
            self.post_hook(b)
            b["done"].record(self.light)
            if self._hook_ev is None:
                self._hook_ev = t.cuda.Event()
            self._hook_ev.record(self.light)

    def flush_hook(self):
        """Enqueue the hook of the last submitted batch (exclusive mode defers it to the next submit)."""
        if self._pending_hook is not None:
            pb, self._pending_hook = self._pending_hook, None
            self._run_hook(pb)

    def _slot(self, slot: int, B: int) -> dict:
        b = self._slot_bufs[slot].get(B)
        if b is None:
            self._bufs.pop(B, None)  # fresh buffer set for this slot
            b = self._buffers(B)
            self._bufs.pop(B, None)
            t = self.torch
            b["scan_done"] = t.cuda.Event()
            b["prep_done"] = t.cuda.Event()
            b["done"] = t.cuda.Event()
            b["done"].record(t.cuda.current_stream(self.device))
            self._slot_bufs[slot][B] = b
        return b

    def submit(self, q, sparse, domain_q=None) -> dict:
        t, cfg = self.torch, self.cfg
        if domain_q is not None and self.hd is None:
            raise ValueError("domain queries need an engine built with domain_handle")
        B = q.shape[0]
        slot = self._n % self.depth
        self._n += 1
        b = self._slot(slot, B)
        kp = b["kp"]
        indptr, idx, val, max_nnz = sparse
        nnz = int(idx.shape[0])
        if self.prep is not None:
            self.prep.wait_event(b["done"])  # the batch that used this slot before has been finished
            self.h.hybrid_prep_dev(q.data_ptr(), indptr.data_ptr(), idx.data_ptr(), val.data_ptr(), B, nnz, int(max_nnz),
                                   kp, slot, self.prep.cuda_stream)
            b["prep_done"].record(self.prep)
            self.heavy.wait_event(b["prep_done"])
        else:
            self.heavy.wait_event(b["done"])
        exclusive = self.post_hook is not None and self.post_hook_exclusive and self.depth >= 2   # (one slot: its buffers are reused at once)
        if exclusive and self._hook_ev is not None:
            self.heavy.wait_event(self._hook_ev)    # the hook enqueued last (batch i - 2) has the chip to itself
        self.h.hybrid_scan_dev(q.data_ptr(), indptr.data_ptr(), idx.data_ptr(), val.data_ptr(), B, nnz, int(max_nnz),
                               kp, slot, self.heavy.cuda_stream)
        with t.cuda.stream(self.heavy):
            self._search_domain(b, domain_q, B, self.heavy.cuda_stream)
        b["scan_done"].record(self.heavy)
        self.light.wait_event(b["scan_done"])
        if not exclusive:
            self.flush_hook()                       # (the mode was switched off with a hook still pending)
        elif self._pending_hook is not None:
            pb, self._pending_hook = self._pending_hook, None
            self._run_hook(pb)                      # batch i - 1's hook: behind this batch's scan, before its finishing work
        with t.cuda.stream(self.light):
            self.h.hybrid_finish_dev(q.data_ptr(), indptr.data_ptr(), idx.data_ptr(), val.data_ptr(), B, int(max_nnz),
                                     kp, slot, b["ids"].data_ptr(), b["scores"].data_ptr(), b["flags"].data_ptr(),
                                     self.light.cuda_stream)
            self._post_lists(b, B, self.light.cuda_stream)
            if exclusive:
                self._pending_hook = b              # "done" is recorded behind the hook, at the next submit or flush_hook()
            else:
                if self.post_hook is not None:
                    self.post_hook(b)
                b["done"].record(self.light)
        return b

    def synchronize(self):
        self.flush_hook()
        if self.prep is not None:
            self.prep.synchronize()
        self.heavy.synchronize()
        self.light.synchronize()

    def close(self):
        """Release the masked streams (if any) and give the scans the whole device back."""
        self.synchronize()
        if self._raw_streams:
            for hh in (self.h, self.hd):
                if hh is not None:
                    hh.set_scan_cus(0)
            raw, self._raw_streams = self._raw_streams, []
            self.heavy = self.light = self.prep = None
            for r in raw:
                nat.stream_destroy(self.device.index or 0, r)

    def all_flags_exact(self) -> bool:
        return all(bool(b["agg_flags"].min().item() == 1) for slot in self._slot_bufs for b in slot.values()
                   if "agg_flags" in b)
