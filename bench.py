#!/usr/bin/env python3
"""Headline benchmark: queries/s (+ p50 retrieve() latency) of the hybrid search
hot path — dense top-k' + sparse top-k' + RRF + learned-ranker rerank — on
BASELINE.json's metric shape: 10M x 768 fp16 corpus, top_k=20 (k'=40), COSINE,
sparse docs of 100 nnz over 10k dims, batch of 128 concurrent queries per step.

    python bench.py --gpus N --steps K --warmup W        (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; the corpus is row-sharded over the N ranks (strong
scaling: 10M rows in total whatever N), each step ends with one RCCL all-gather
of the per-shard top-k' lists.  Rank 0 prints ONE JSON line.

What is inside the timed region: K full steps (all kernels + collective) with
queries already resident in HBM; nothing is cached between steps (each step
uses a different query batch) and nothing is skipped.  By default three batches
are in flight (--in-flight 3; 4.99 / 4.79 / 4.68 / 4.65 ms per step for 1 / 2 / 3 / 4 on one box): the scans of batch i+1 run on a heavy stream
while batch i is finished on a light stream — the way the reference's service
overlaps up to 64 concurrent retrieve() calls (service.py:137,149); every step's
results are complete when the region ends.  The dense-scan kernel is
bracketed with HIP events on its own stream inside the timed region
(hr_set_profiling(1)) to get roofline.achieved.  The CPU baseline (N=1 only) is
the oracle's numpy/scipy restatement of the reference path timed on a 1M-row
subsample and extrapolated per row — labelled as such.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
sys.path.insert(0, ROOT)

SPARSE_DIM = 10000
SPARSE_NNZ = 100
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0  # same guide: what a float4 copy measures ("also report /6290")
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense fp16/bf16 MFMA


def dense_block(block: int, n: int, dim: int, seed: int = 1234) -> np.ndarray:
    """Rows of corpus block `block`: N(0,1) fp32 from default_rng(seed + block)."""
    return np.random.default_rng(seed + block).standard_normal((n, dim), dtype=np.float32)


def sparse_block(block: int, n: int, seed: int = 5678):
    """n sparse docs: 100 distinct ascending indices in [0,10000) (one uniform pick per
    stratum of 100 — the vectorised stand-in for the reference placeholder's
    choice(10000, 100, replace=False), indexing.py:647-654) with |N(0,1)| fp32 values."""
    rng = np.random.default_rng(seed + block)
    idx = (np.arange(SPARSE_NNZ, dtype=np.int32) * (SPARSE_DIM // SPARSE_NNZ))[None, :] + \
        rng.integers(0, SPARSE_DIM // SPARSE_NNZ, size=(n, SPARSE_NNZ), dtype=np.int32)
    val = np.abs(rng.standard_normal((n, SPARSE_NNZ), dtype=np.float32))
    indptr = np.arange(n + 1, dtype=np.int64) * SPARSE_NNZ
    return indptr, idx.reshape(-1), val.reshape(-1)


ZIPF_S = 1.1


def zipf_term_probs(V: int = SPARSE_DIM, mean_nnz: float = SPARSE_NNZ, s: float = ZIPF_S) -> np.ndarray:
    """P(term t occurs in a doc) = min(1, c / (t + 1)^s) with c such that a doc holds `mean_nnz` terms on average:
    the Zipfian shape of real BM25 postings: at V = 10 000 / 100 terms per doc some twenty terms occur in
    every doc, term 32 in half of them, the tail in fewer than 1 in 1 000."""
    w = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** s
    lo, hi = 0.0, float(V)
    for _ in range(60):  # bisection on c
        c = 0.5 * (lo + hi)
        if np.minimum(1.0, c * w).sum() > mean_nnz:
            hi = c
        else:
            lo = c
    return np.minimum(1.0, 0.5 * (lo + hi) * w)


def sparse_block_zipf(block: int, n: int, seed: int = 5678, V: int = SPARSE_DIM, mean_nnz: float = SPARSE_NNZ):
    """n sparse docs whose terms follow zipf_term_probs (independent per term), |N(0,1)| fp32 values, CSR with
    ascending distinct indices per row.  Heavy terms (p >= 1/8) are drawn as a Bernoulli matrix, the tail by a Poisson
    number of inverse-CDF draws per doc (duplicates dropped)."""
    rng = np.random.default_rng(seed + block)
    p = zipf_term_probs(V, mean_nnz)
    n_head = int((p >= 0.125).sum())
    head = rng.random((n, n_head), dtype=np.float32) < p[:n_head].astype(np.float32)
    hd, ht = np.nonzero(head)
    tail_p = p[n_head:]
    lam = float(tail_p.sum())
    k = rng.poisson(lam, size=n)
    td = np.repeat(np.arange(n, dtype=np.int64), k)
    tt = n_head + np.searchsorted(np.cumsum(tail_p) / lam, rng.random(td.shape[0]), side="right")
    tt = np.minimum(tt, V - 1)
    key = np.concatenate([hd.astype(np.int64) * V + ht, td * V + tt])
    key = np.unique(key)  # sorted by (doc, term), duplicates of the tail draws dropped
    doc = key // V
    idx = (key - doc * V).astype(np.int32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(doc, minlength=n), out=indptr[1:])
    val = np.abs(rng.standard_normal(idx.shape[0], dtype=np.float32))
    return indptr, idx, val


def zipf_queries(rng, B: int, n_terms: int = 20, V: int = SPARSE_DIM):
    """B queries of n_terms distinct terms drawn with the corpus' own term distribution (so frequent terms are in
    most queries, as stop-word-free BM25 queries still have them), |N(0,1)| weights."""
    p = zipf_term_probs(V)
    p = p / p.sum()
    out = []
    for _ in range(B):
        qi = np.sort(rng.choice(V, size=n_terms, replace=False, p=p)).astype(np.int32)
        out.append((qi, np.abs(rng.standard_normal(n_terms)).astype(np.float32)))
    return out


def pmc_traffic(rows: int, dim: int, batch: int, n_gpus: int, kernel: str = "dense_scan", dist: str = "uniform"):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*_pmc_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script, FETCH doubled as the gfx950 guide
    prescribes).  Only returned when the profiled workload is the one being run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            w = d.get("workload", {})
            if n_gpus == 1 and (w.get("rows"), w.get("dim"), w.get("batch"), w.get("sparse_dist", "uniform")) == (rows, dim, batch, dist):
                # the NEWEST collection of this workload is the only one consulted: an older round's file lists kernels
                # (the five-launch finishing chain) that a later round's step no longer launches
                if kernel in d["kernels"]:
                    return d["kernels"][kernel]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
                return None, None
        except Exception:
            continue
    return None, None


# search kernels of a step; each is launched once (the selection kernels cover both modalities of a hybrid step in one launch)
STEP_KERNELS = ("dense_scan", "sparse_scan", "refine_dense", "refine_sparse", "select_groups", "bucket_max", "select_topk",
                "finish_fused", "post_lists")


def step_hbm(rows: int, dim: int, batch: int, n_gpus: int, dist: str, use_sparse: bool, ms_per_step: float):
    """HBM bytes ALL search kernels of a step move (the committed PMC passes: FETCH + WRITE per launch, one launch of
    each per step) over the measured step time: how full the memory system is as a whole, beside the per-kernel rooflines."""
    total, src = 0.0, None
    for k in STEP_KERNELS:
        if not use_sparse and k in ("sparse_scan", "refine_sparse"):
            continue
        b, path = pmc_traffic(rows, dim, batch, n_gpus, kernel=k, dist=dist)
        if b is None:
            if k in ("dense_scan", "sparse_scan"):
                return None
            continue
        total += b
        src = path
    gbps = total / (ms_per_step * 1e-3) / 1e9
    return {"bytes_per_step": total, "achieved": gbps, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "frac_of_measured_copy": gbps / HBM_COPY_GBPS, "traffic_source": src}


def scan_kernel_name(B: int, dim: int, mask: int = 0) -> str:
    """Which dense scan serves a batch of B queries (dense_search_enqueue in csrc/hbmrag.hip); mask = --dense-kernel-mask."""
    kt = -(-(-(-dim // 32)) // 4) * 4          # 1 KiB tiles per row, padded to a multiple of 4
    if B > 128 and kt == 24 and (mask & 16) and not (mask & 1):
        return "dense_scan_q64_kernel<KT=24> (256 queries per pass: 4 waves x 64 queries in the unified register file)"
    if B > 128 and kt == 24:
        return "dense_scan_qreg_kernel<KT=24,GW=2,NW=8> (256 queries per pass)"
    if B > 128 and kt >= 8:
        return "dense_scan_gemm_kernel<GQ=16> (256 queries per pass, both operands through LDS)"
    if B > 64:
        return "dense_scan_bigq_kernel<f16,GQ=8> (128 queries per pass)"
    return "dense_scan_kernel<f16> (up to 64 queries per pass)"


def make_queries(n_batches: int, B: int, dim: int, seed: int = 4321, dist: str = "uniform"):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((n_batches, B, dim), dtype=np.float32)
    sq = []
    for _ in range(n_batches):
        if dist == "zipf":
            sq.append(zipf_queries(rng, B))
            continue
        _, idx, val = sparse_block(0, B, seed=int(rng.integers(1 << 30)))
        sq.append([(idx[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ], val[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ]) for i in range(B)])
    return Q, sq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=128,
                    help="concurrent queries per step (the reference's service admits RAG_MAX_CONCURRENCY=64 requests "
                         "per instance by default, service.py:137; 128 = one large-batch pass of the dense scan)")
    ap.add_argument("--top-k", type=int, default=20)
    ap.add_argument("--block-rows", type=int, default=250_000)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline / parity-gate subsample")
    ap.add_argument("--latency-queries", type=int, default=200)
    ap.add_argument("--rerank", choices=("learned", "cross-encoder"), default="learned",
                    help="learned = LearnedRanker's linear score (the reference's deterministic rerank branch, HIP kernel); "
                         "cross-encoder = additionally run a random-init MiniLM-L6 cross-encoder (PyTorch-ROCm) over the "
                         "top_k fused candidates of every query and keep its best rerank_top_k (BASELINE config 4's '20->5')")
    ap.add_argument("--ce-seq-len", type=int, default=128)
    ap.add_argument("--no-sparse", action="store_true")
    ap.add_argument("--sparse-dist", choices=("uniform", "zipf"), default="uniform",
                    help="uniform = the reference's placeholder (100 uniform terms per doc, indexing.py:647-654; ~80-term queries); "
                         "zipf = Zipfian postings (s = 1.1: some twenty terms in every doc, df = 50 %% at term 32) and 20-term "
                         "queries drawn from the same distribution (SURVEY §7 'sparse skew')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-api-concurrent", action="store_true", help="skip the concurrent retrieve() measurement")
    ap.add_argument("--no-config4-full", action="store_true", help="skip the second timed region with the cross-encoder rerank")
    ap.add_argument("--ce-overlap", action="store_true",
                    help="let the cross-encoder forward share the chip with the next batches' scans (round 3's arrangement) "
                         "instead of giving it the chip to itself")
    ap.add_argument("--ingest", action="store_true",
                    help="measure AdvancedRAGPipeline.ingest_documents instead of the search step: synthetic documents of "
                         "~512 tokens through SentenceEncoder (random-init, bge-base shape at --dim 768) + BM25SparseEncoder "
                         "into an fp16 shard; the device-to-device ingest path beside the host-hop one; prints its own JSON line")
    ap.add_argument("--ingest-docs", type=int, default=1024)
    ap.add_argument("--ingest-batch", type=int, default=128, help="documents per ingest_documents call")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel phase, not just the scans")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="query batches kept in flight: 1 = one batch at a time; 2..4 = scans of batch i+1 on a heavy "
                         "stream while batch i is finished (select/refine/exchange/fuse) on a light stream")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="one process, one GPU: put the post-exchange work of a W-rank step (merge of W lists per modality, "
                         "fed with W copies of the local lists) on the finishing stream — with --rows N/W a PROJECTION of "
                         "the per-rank step of an N-row corpus on W GPUs; the collective itself is not simulated")
    ap.add_argument("--light-cus", type=int, default=0,
                    help="confine the finishing + prep streams to this many compute units and the scans to the rest "
                         "(CU-masked HIP streams); 0 = no masks, stream priorities only")
    ap.add_argument("--prep-stream", action="store_true", help="query prep on a third stream instead of the scan stream")
    ap.add_argument("--group-rows", type=int, choices=(0, 16, 64), default=0, help="rows per candidate group (0 = by shard size)")
    ap.add_argument("--no-trim", action="store_true",
                    help="refine all candidate groups (round 3's constant 64 per query) instead of the data-dependent set: A/B")
    ap.add_argument("--dense-kernel-mask", type=int, default=0,
                    help="hr_debug_option(HR_DEBUG_DENSE_KERNELS) bit mask for A/B runs")
    ap.add_argument("--finish-mode", choices=("auto", "chain", "fused"), default="auto",
                    help="finishing path: auto = fused kernel for batches that fill the chip, chain = five launches")
    args = ap.parse_args()
    if args.ingest:
        return bench_ingest(args)

    import torch
    import torch.distributed as dist
    from advanced_rag import _native as nat
    from advanced_rag.engine import (EngineConfig, HybridSearchEngine, PipelinedSearchEngine, pack_sparse_queries,
                                     shard_range)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # Rehearsal on a one-GPU box: BENCH_REHEARSAL=1 puts every rank on cuda:0 and exchanges over gloo
    # (RCCL refuses two ranks on one device).  Never used for reported numbers.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # RCCL's internal stream at high priority: HIP maps streams to hardware queues per priority level, so
            # the all-gather can never queue up behind a scan of the (normal-priority) heavy stream
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=dev, pg_options=opts)
            except Exception:  # older torch without the option: plain group
                if dist.is_initialized():
                    raise
                dist.init_process_group("nccl", device_id=dev)

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    use_sparse = not args.no_sparse
    N, D, B, blk = args.rows, args.dim, args.batch, min(args.block_rows, args.rows)
    lo, hi = shard_range(N, rank, world, align=blk)
    n_local = hi - lo
    nat.debug_option(nat.HR_DEBUG_NO_TRIM, 1 if args.no_trim else 0)
    nat.debug_option(nat.HR_DEBUG_DENSE_KERNELS, args.dense_kernel_mask)
    nat.debug_option(nat.HR_DEBUG_GROUP_ROWS, args.group_rows)
    h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, SPARSE_DIM if use_sparse else 0, local_rank)
    nat.debug_option(nat.HR_DEBUG_GROUP_ROWS, 0)
    h.set_row_offset(lo)
    h.reserve(max(n_local, 64))

    # ---- build this rank's shard (generated in blocks by a thread pool, ingested in order) ----------
    do_cpu = (world == 1 and not args.no_cpu_baseline)
    cpu_rows = min(args.cpu_rows, N)
    sample_dense, sample_sparse = [], []
    t0 = time.time()
    blocks = [(b, min(blk, N - b * blk)) for b in range(lo // blk, -(-hi // blk))] if n_local else []
    workers = max(1, min(12, (os.cpu_count() or 8) // max(1, min(world, 8) if world > 1 else 1)))

    sparse_gen = sparse_block_zipf if args.sparse_dist == "zipf" else sparse_block
    df = np.zeros(SPARSE_DIM, dtype=np.int64)  # postings per term in this rank's shard (for the sparse roofline)

    def gen(bn):
        b, n = bn
        return b, dense_block(b, n, D), (sparse_gen(b, n) if use_sparse else None)

    with ThreadPoolExecutor(max_workers=workers) as pool:
        pending = []
        it = iter(blocks)
        for _ in range(workers):
            nxt = next(it, None)
            if nxt is not None:
                pending.append(pool.submit(gen, nxt))
        while pending:
            b, Xb, Sb = pending.pop(0).result()
            nxt = next(it, None)
            if nxt is not None:
                pending.append(pool.submit(gen, nxt))
            h.add_dense(Xb)  # fp32 -> fp16 (RNE) on the device
            if use_sparse:
                h.add_sparse(*Sb)
                df += np.bincount(Sb[1], minlength=SPARSE_DIM)
            if do_cpu and b * blk < cpu_rows:
                take = min(Xb.shape[0], cpu_rows - b * blk)
                sample_dense.append(Xb[:take].astype(np.float16))
                if use_sparse:
                    sample_sparse.append((Sb[0][:take + 1], Sb[1][:Sb[0][take]], Sb[2][:Sb[0][take]]))
            del Xb, Sb
    h.finalize()
    torch.cuda.synchronize()
    log(f"shard rows [{lo},{hi}) built in {time.time() - t0:.1f}s, {h.device_bytes / 1e9:.2f} GB in HBM")

    # ---- queries (resident in HBM before the timed region) --------------------------------------------
    n_batches = 8
    Q, SQ = make_queries(n_batches, B, D, dist=args.sparse_dist)
    cfg = EngineConfig(top_k=args.top_k, use_sparse=use_sparse)
    n_fly = max(1, args.in_flight) if use_sparse else 1
    nat.debug_option(nat.HR_DEBUG_FINISH_MODE, {"auto": 0, "chain": 1, "fused": 2}[args.finish_mode])
    sim = args.simulate_ranks if world == 1 else 0
    eng = PipelinedSearchEngine(h, cfg, device=str(dev), depth=n_fly, simulate_ranks=sim, light_cus=args.light_cus,
                                prep_stream=args.prep_stream) if n_fly > 1 else \
        HybridSearchEngine(h, cfg, device=str(dev), simulate_ranks=sim)
    dQ = [torch.from_numpy(Q[i]).to(dev) for i in range(n_batches)]
    dS = [eng.upload_sparse(pack_sparse_queries(SQ[i], 0.2)) if use_sparse else None for i in range(n_batches)]
    kp = 2 * args.top_k

    # ---- parity gate on the CPU subsample (N=1): GPU restricted by a row mask vs the canonical oracle -----
    parity = None
    if do_cpu:
        import oracle
        Xs = np.concatenate(sample_dense)
        n_s = Xs.shape[0]
        mask = np.zeros((n_local + 7) // 8, dtype=np.uint8)
        mask[: n_s // 8] = 0xFF
        for r in range(n_s // 8 * 8, n_s):
            mask[r >> 3] |= 1 << (r & 7)
        n_gate = 32
        gids, gsc = h.search_dense(Q[0][:n_gate], kp, mask)
        # (the oracle is single-threaded C; its per-query calls release the GIL, so the gate's queries run side by side)
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 8)) as ex:
            parts = list(ex.map(lambda b: oracle.dense_search(Xs, Q[0][b:b + 1], kp, oracle.COSINE), range(n_gate)))
        oids, osc = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        ok = bool(np.array_equal(gids, oids) and np.array_equal(gsc.view(np.uint32), osc.view(np.uint32)))
        fused_ok = True
        if use_sparse:
            s_idx = np.concatenate([p[1] for p in sample_sparse])
            s_val = np.concatenate([p[2] for p in sample_sparse])
            s_ptr = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p[0]) for p in sample_sparse]))]).astype(np.int64)
            sids, ssc = h.search_sparse(SQ[0][:n_gate], kp, 0.2, mask)
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 8)) as ex:
                parts = list(ex.map(lambda b: oracle.sparse_search(s_ptr, s_idx, s_val, SQ[0][b:b + 1], kp, 0.2), range(n_gate)))
            osids, ossc = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
            ok = ok and bool(np.array_equal(sids, osids) and np.array_equal(ssc.view(np.uint32), ossc.view(np.uint32)))
            for i in range(n_gate):
                fi, fs, _ = h.fuse_rrf(gids[i], sids[i][sids[i] >= 0], (), cfg.dense_weight, cfg.sparse_weight, 0.2, 60)
                oi, os_, _ = oracle.rrf(oids[i], osids[i][osids[i] >= 0], (), cfg.dense_weight, cfg.sparse_weight, 0.2, 60)
                fused_ok = fused_ok and bool(np.array_equal(fi, oi) and np.max(np.abs(fs - os_), initial=0.0) <= 1e-4)
        parity = {"sample_rows": int(n_s), "queries": n_gate, "ids_bit_exact": ok, "fused_within_1e-4": fused_ok}
        log(f"parity gate on {n_s} rows: {parity}")
        if not (ok and fused_ok):
            print(json.dumps({"error": "parity gate failed", "parity": parity}))
            sys.exit(2)

    # ---- timed region -----------------------------------------------------------------------------------
    ce_note = None
    ce_model = {}

    def make_cross_encoder(T):
        """(hook, events, pairs per step) of the cross-encoder leg at T tokens per pair: a random-init MiniLM-L6
        cross-encoder (PyTorch-ROCm) over the top_k fused candidates of every query, keeping its best rerank_top_k."""
        from advanced_rag.encoders import CrossEncoderModel
        if "m" not in ce_model:
            ce_model["m"] = CrossEncoderModel(device=str(dev), max_len=512)
        ce = ce_model["m"]
        vocab = ce.config.vocab_size
        # The rerank is per query, not per shard: with W ranks each rank scores the fused candidates of
        # its ceil(B / W) queries and one small all-gather (5 ids + scores per query) puts the result on every rank.
        ce_nq = -(-B // world)
        ce_q0 = min(B, rank * ce_nq)
        ce_q1 = min(B, ce_q0 + ce_nq)
        ce_pairs = (ce_q1 - ce_q0) * args.top_k
        pos = torch.arange(T, device=dev, dtype=torch.int64)[None, None, :]
        types = torch.zeros((ce_pairs, T), dtype=torch.long, device=dev)
        types[:, T // 4:] = 1
        mask = torch.ones((ce_pairs, T), dtype=torch.bool, device=dev)
        qslot = torch.arange(ce_q0, ce_q1, device=dev)[:, None, None]
        events = []

        def cross_encode(b):
            # synthetic token ids derived from (query slot, fused doc id, position): there is no text behind the
            # random corpus; the forward pass (the cost being measured) does not depend on what the tokens are
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(dev))
            _cross_encode(b)
            ev1.record(torch.cuda.current_stream(dev))
            events.append((ev0, ev1))

        def _cross_encode(b):
            fused = b["fused_ids"][ce_q0:ce_q1]
            ids = torch.full((ce_nq, cfg.rerank_top_k), -1, dtype=torch.int64, device=dev)
            sc = torch.full((ce_nq, cfg.rerank_top_k), float("-inf"), dtype=torch.float32, device=dev)
            if ce_pairs:
                doc = fused.clamp_min(0)[:, :, None]
                toks = (1000 + (doc * 7919 + pos * 104729 + qslot * 31) % (vocab - 1000))
                toks = toks.view(ce_pairs, T)
                toks[:, 0] = 101
                with torch.inference_mode():
                    scores = ce.module(toks, types, mask).view(ce_q1 - ce_q0, args.top_k)
                scores = scores.float().masked_fill(fused < 0, float("-inf"))
                top = torch.topk(scores, cfg.rerank_top_k, dim=1)
                ids[: ce_q1 - ce_q0] = torch.gather(fused, 1, top.indices)
                sc[: ce_q1 - ce_q0] = top.values
            if world > 1:
                all_ids = torch.empty((world * ce_nq, cfg.rerank_top_k), dtype=torch.int64, device="cpu" if rehearsal else dev)
                all_sc = torch.empty((world * ce_nq, cfg.rerank_top_k), dtype=torch.float32, device="cpu" if rehearsal else dev)
                dist.all_gather_into_tensor(all_ids, ids.cpu() if rehearsal else ids)
                dist.all_gather_into_tensor(all_sc, sc.cpu() if rehearsal else sc)
                ids, sc = all_ids[:B].to(dev), all_sc[:B].to(dev)
            b["ce_ids"] = ids[:B]
            b["ce_scores"] = sc[:B]

        def forward_alone(reps=5):
            """ms per forward of the same pairs on an otherwise idle chip (the figure in the pipeline shares the chip with the
            scans of the next batches)."""
            if not ce_pairs:
                return 0.0
            toks = (1000 + (pos * 104729 + qslot * 31 + torch.arange(args.top_k, device=dev)[None, :, None] * 7919) % (vocab - 1000)).view(ce_pairs, T)
            toks[:, 0] = 101
            torch.cuda.synchronize()
            with torch.inference_mode():
                for _ in range(2):
                    ce.module(toks, types, mask)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    ce.module(toks, types, mask)
                e1.record()
                e1.synchronize()
            return e0.elapsed_time(e1) / reps

        cross_encode.alone = forward_alone
        return cross_encode, events, ce_pairs, ce

    if args.rerank == "cross-encoder":
        T = args.ce_seq_len
        cross_encode, ce_events, ce_pairs, ce = make_cross_encoder(T)
        ce_note = (f"+ cross-encoder rerank {args.top_k}->{cfg.rerank_top_k}: random-init MiniLM-L6-H384 (hand-written HIP layer kernels, fp16), "
                   f"{B * args.top_k} pairs x {T} tokens per step" + (f", the queries split over the {world} ranks" if world > 1 else ""))
        if n_fly > 1:
            eng.post_hook = cross_encode
            eng.post_hook_exclusive = not args.ce_overlap

    def step(i):
        if n_fly > 1:
            return eng.submit(dQ[i % n_batches], dS[i % n_batches])
        out = eng.search(dQ[i % n_batches], dS[i % n_batches])
        if args.rerank == "cross-encoder":
            cross_encode(out)
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if args.rerank == "cross-encoder":
        ce_events.clear()
    h.kernel_ms()  # drop warm-up spans
    h.set_profiling(2 if args.profile_all else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    host_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    h.set_profiling(0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    phases = h.kernel_ms()
    flags_exact = eng.all_flags_exact() if n_fly > 1 else bool(out["flags"].min().item() == 1)
    if world > 1:
        ft = torch.tensor([1 if flags_exact else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(ft, op=dist.ReduceOp.MIN)
        flags_exact = bool(ft.item())

    # the finishing kernel and the post-exchange kernel of the last batch once more, ALONE on an idle chip (their
    # in-region times are those of kernels starved by the scans they run beside): what the chain costs by itself
    alone = None
    if n_fly > 1 and use_sparse:
        torch.cuda.synchronize()
        st = torch.cuda.current_stream(dev)
        b, slot = out, (eng._n - 1) % eng.depth
        q_, (ip_, ix_, iv_, mx_) = dQ[(args.warmup + args.steps - 1) % n_batches], dS[(args.warmup + args.steps - 1) % n_batches]
        def timed(fn, reps=30):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            e1.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        fin_us = timed(lambda: h.hybrid_finish_dev(q_.data_ptr(), ip_.data_ptr(), ix_.data_ptr(), iv_.data_ptr(), B, int(mx_), kp,
                                                   slot, b["ids"].data_ptr(), b["scores"].data_ptr(), b["flags"].data_ptr(),
                                                   st.cuda_stream))
        pa = eng._post_args(b, B, eng.n_lists, b.get("gathered"))
        post_us = timed(lambda: nat.post_lists_dev(pa, B, st.cuda_stream))
        alone = {"finish_us": fin_us, "post_lists_us": post_us, "lists_merged_per_modality": eng.n_lists,
                 "note": "back-to-back launches on an idle chip, launch overhead included"}

    # ---- full BASELINE config 4: the same step WITH the cross-encoder rerank 20 -> 5 (reference retrieval.py:518-563,
    # :651-681), a second short timed region on the same shard, at 128 and at 512 tokens per (query, document) pair
    config4_full = None
    if n_fly > 1 and use_sparse and args.rerank == "learned" and not args.no_config4_full and (N, D) == (10_000_000, 768):
        config4_full = {}
        for T in (128, 512):
            hook, evs, pairs, ce_m = make_cross_encoder(T)
            eng.post_hook = hook
            eng.post_hook_exclusive = not args.ce_overlap
            k_full, w_full = (10, 3) if T == 128 else (4, 2)
            for i in range(w_full):
                step(i)
            torch.cuda.synchronize()
            evs.clear()
            h.kernel_ms()
            h.set_profiling(1)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(k_full):
                step(w_full + i)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t1
            h.set_profiling(0)
            if world > 1:
                tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            ph = h.kernel_ms()
            eng.synchronize()
            alone_ms = hook.alone()
            config4_full[f"seq_len_{T}"] = {
                "value": B * k_full / el, "unit": "queries/s", "ms_per_step": el / k_full * 1e3, "steps": k_full, "warmup": w_full,
                "cross_encoder": ce_report(ce_m, evs, pairs, T, alone_ms),
                "forward_placement": "shares the chip with the next batches' scans" if args.ce_overlap else
                                     "exclusive: behind the next batch's scans, ahead of the scans after that (engine.post_hook_exclusive)",
                "dense_scan_ms": round(ph["dense_scan"][0], 4), "sparse_scan_ms": round(ph["sparse_scan"][0], 4)}
            eng.post_hook = None
        eng.synchronize()
        config4_full["note"] = ("hybrid dense+sparse -> RRF -> cross-encoder rerank 20 -> 5: random-init MiniLM-L6-H384, fp16, every full layer "
                                "as three hand-written HIP launches (QKV projection, attention, output projection + LayerNorm + FFN + "
                                "LayerNorm: csrc/encoder_layer.h, attention.h), the last layer's token-0 rows and the head through "
                                "PyTorch-ROCm; synthetic token ids; the forward runs on the finishing stream with the chip to itself "
                                "(compute-bound beside HBM-bound scans gained nothing: both slowed down), the finishing kernels of the "
                                "batch overlap the next batch's scans; `frac_alone` is the same forward timed back to back on an idle chip")

    # every rank merges the same gathered lists, so every rank must hold the same fused answer for the last batch
    ranks_agree = None
    if world > 1:
        w8 = torch.arange(1, out["fused_ids"].numel() + 1, dtype=torch.int64, device=out["fused_ids"].device)
        chk = (out["fused_ids"].reshape(-1).to(torch.int64) * w8).sum().reshape(1)
        chk = chk.cpu() if rehearsal else chk
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        ranks_agree = all(int(c.item()) == int(allc[0].item()) for c in allc)

    # Roofline of the dominant kernel, per STEP: SURVEY §8(d) counts ONE pass over the shard for the B queries of a
    # step, so a batch that takes several scan launches (256 queries at D = 1024: two 128-query passes) is charged
    # all of them.  passes_per_step = 1 on the default workload, where per-step and per-launch figures coincide.
    scan_ms, scan_launches = phases["dense_scan"]
    scan_bytes = h.dense_scan_bytes
    passes = scan_launches / args.steps if args.steps else 1.0
    step_scan_ms = scan_ms * passes
    achieved = scan_bytes / (step_scan_ms * 1e-3) / 1e9 if step_scan_ms > 0 else 0.0
    per_launch = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    dpad = -(-D // 128) * 128
    mfma_tflops = 2.0 * n_local * dpad * B / (step_scan_ms * 1e-3) / 1e12 if step_scan_ms > 0 else 0.0

    # Sparse scan roofline: algorithmic bytes of one launch = every posting of every kept query term
    # (4 B: fp16 weight | u16 slot) + 2 * 4 B of run bounds per (term, range), averaged over the query batches timed.
    sparse_roof = None
    if use_sparse:
        sp_ms, sp_launches = phases["sparse_scan"]
        n_ranges = -(-n_local // 16384)
        per_batch = []
        for sq in SQ:
            _, kept_idx, _, _ = pack_sparse_queries(sq, 0.2)
            per_batch.append(4 * int(df[kept_idx].sum()) + 8 * n_ranges * int(kept_idx.shape[0]))
        sp_bytes = float(np.mean(per_batch))
        sp_achieved = sp_bytes / (sp_ms * 1e-3) / 1e9 if sp_ms > 0 else 0.0
        sp_traffic, sp_src = pmc_traffic(N, D, B, world, kernel="sparse_scan", dist=args.sparse_dist)
        sparse_roof = {"bound": "hbm", "kernel": "sparse_scan_kernel", "achieved": sp_achieved, "peak": HBM_PEAK_GBPS,
                       "unit": "GB/s", "frac": sp_achieved / HBM_PEAK_GBPS, "frac_of_measured_copy": sp_achieved / HBM_COPY_GBPS,
                       "traffic": sp_traffic, "traffic_source": sp_src,
                       "algorithmic_bytes_per_launch": sp_bytes, "avg_launch_ms": sp_ms, "launches": sp_launches,
                       "postings_per_query": float(np.mean(per_batch)) / 4 / B, "distribution": args.sparse_dist}
        if args.sparse_dist == "zipf":
            # the algorithmic figure prices every posting at 4 bytes; the runs of terms that at least half of a range's docs
            # have are STORED at 2 bytes per doc of the range (csrc/sparse.h), so on this corpus the scan fetches less than
            # it is credited with and `frac` can pass what the memory system can deliver: read it as a speed in
            # postings, not as a bandwidth (`traffic` is what it really moves: profiles/r3_pmc_traffic_zipf.json)
            sparse_roof["note"] = "4 B per posting credited; frequent terms' runs are stored at 2 B per doc (dense form)"

    # ---- p50 latency of single-query retrieve() through the Python API ---------------------------------------------
    # N = 1: the manager holds the one shard.  N > 1: the collection spans the ranks (torchrun form of the sharded
    # manager): rank 0 calls retrieve(), every rank searches its shard, one gather per modality brings the lists back.
    latency = None
    if not args.no_latency and args.latency_queries > 0:
        latency = measure_latency(h, Q, SQ, args, use_sparse, world, rank, lo, N)
        if world > 1:
            dist.barrier()

    # ---- CPU baseline (N=1): the reference path with Milvus replaced by numpy/scipy, on the subsample -----------
    cpu = None
    if do_cpu:
        import scipy.sparse as sp
        import oracle
        X32 = Xs.astype(np.float32)
        X32 /= np.maximum(np.linalg.norm(X32, axis=1, keepdims=True), 1e-30)
        csr = q_csr = None
        if use_sparse:
            csr = sp.csr_matrix((s_val, s_idx, s_ptr), shape=(n_s, SPARSE_DIM), dtype=np.float32)
            p, i_, v_, _ = pack_sparse_queries(SQ[1], 0.2)
            q_csr = sp.csr_matrix((v_, i_, p), shape=(B, SPARSE_DIM), dtype=np.float32)
        oracle.cpu_hybrid(X32, csr, Q[1][:4], q_csr[:4] if use_sparse else None, args.top_k) if use_sparse else \
            oracle.cpu_dense_topk(X32, Q[1][:4], kp)  # warm-up
        reps, t_cpu = 0, 0.0
        while t_cpu < 10.0 and reps < 20:
            t1 = time.perf_counter()
            if use_sparse:
                oracle.cpu_hybrid(X32, csr, Q[1], q_csr, args.top_k)
            else:
                oracle.cpu_dense_topk(X32, Q[1], kp)
            t_cpu += time.perf_counter() - t1
            reps += 1
        per_batch = t_cpu / reps * (N / n_s)  # per-row extrapolation to the full corpus
        cpu = {"value": B / per_batch, "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
               "sample": f"oracle.cpu_hybrid (numpy fp32 BLAS matmul + argpartition, scipy CSR*q, Python RRF) on the "
                         f"first {n_s} rows x {B} queries, {reps} reps, time extrapolated x{N / n_s:.1f} to {N} rows"}

    if rank == 0:
        traffic, traffic_src = pmc_traffic(N, D, B, world)
        total_q = B * args.steps
        res = {
            "metric": "queries_per_sec_hybrid_retrieve", "value": total_q / elapsed, "unit": "queries/s",
            "n_gpus": world, **({"rehearsal_single_gpu_gloo": True} if rehearsal else {}), "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{N}x{D} fp16 COSINE corpus" + (f" + sparse {SPARSE_NNZ}nnz/{SPARSE_DIM}d" + (" zipf(1.1)" if args.sparse_dist == "zipf" else "") if use_sparse else "")
                       + (f", hybrid dense+sparse k'={kp} -> RRF(k=60, 0.7/0.3)" if use_sparse else f", dense only k'={kp} -> RRF(k=60) of the one list")
                       + f" top_k={args.top_k} -> learned-ranker rerank {args.top_k}->{cfg.rerank_top_k}, batch {B} queries/step "
                       + (ce_note if ce_note else ("(BASELINE config 4 without the cross-encoder forward)" if (N, D) == (10_000_000, 768) and use_sparse
                                                   else "(BASELINE config 5 on one GPU)" if (N, D, B, use_sparse) == (50_000_000, 1024, 256, False) else "")),
                       "rows": N, "dim": D, "batch": B, "top_k": args.top_k, "k_prime": kp,
                       "batches_in_flight": n_fly,
                       "streams": ("heavy (scans) + light (finish, exchange, post)" + (" + prep (query preparation)" if args.prep_stream else "")
                                   + (f"; CU masks: {args.light_cus} CUs for light + prep, the rest for the scans" if args.light_cus else "; priorities only")) if n_fly > 1 else "one stream",
                       "finish": args.finish_mode, "group_rows": args.group_rows or "by shard size",
                       **({"dense_kernel_mask": args.dense_kernel_mask} if args.dense_kernel_mask else {}),
                       **({"simulate_ranks": sim, "projection": f"per-rank step of a {N * sim}-row corpus on {sim} GPUs: the merge of {sim} lists per modality "
                           "runs on the finishing stream, the all-gather itself is NOT included"} if sim else {}),
                       "parallelism": f"row-sharded x{world}, one RCCL all-gather of per-shard top-k' per step" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": scan_kernel_name(B, D, args.dense_kernel_mask), "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "frac_of_measured_copy": achieved / HBM_COPY_GBPS,
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": scan_bytes, "avg_launch_ms": scan_ms, "launches": scan_launches,
                         "passes_per_step": passes, "per_launch_frac": per_launch / HBM_PEAK_GBPS,
                         # the batched distance is a dense contraction once B >= 16: its share of the fp16 MFMA peak
                         "mfma": {"achieved": mfma_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": mfma_tflops / MFMA_F16_PEAK_TFLOPS}},
            "roofline_sparse": sparse_roof,
            "step_hbm": step_hbm(N, D, B, world, args.sparse_dist, use_sparse, elapsed / args.steps * 1e3),
            **({"cross_encoder": ce_report(ce, ce_events, ce_pairs, args.ce_seq_len)} if args.rerank == "cross-encoder" else {}),
            "cpu_baseline": cpu,
            "kernel_ms": {k: round(v[0], 4) for k, v in phases.items() if v[1]},
            **({"finishing_alone": alone} if alone else {}),
            **({"config4_full": config4_full} if config4_full else {}),
            "all_lists_proven_exact": flags_exact, **({"ranks_agree": ranks_agree} if ranks_agree is not None else {}),
            "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
        }
        if latency:
            res.update(latency)
        if parity:
            res["parity_gate"] = parity
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def bench_ingest(args):
    """docs/s and chunks/s of AdvancedRAGPipeline.ingest_documents (reference pipeline.py:120-215 -> indexing.py:264-437)
    on one GPU: host work (diagnostics, chunking, enrichment, BM25) + encoder forward + append + flush, for the
    device-to-device dense path (encoder output -> hr_add_dense_raw_dev) and for the host-hop path
    (.cpu().numpy() -> np.stack -> hr_add_dense) side by side."""
    import asyncio
    import contextlib
    import torch
    from advanced_rag import AdvancedRAGPipeline, BM25SparseEncoder, PipelineConfig
    from advanced_rag.encoders import EncoderConfig, SentenceEncoder
    from advanced_rag.embedding_cache import initialize_caches

    dev = "cuda:0"
    rng = np.random.default_rng(99)
    vocab = [f"w{i}" for i in range(20000)]
    zipf = 1.0 / np.arange(1, len(vocab) + 1) ** 1.05
    zipf /= zipf.sum()

    def make_doc(i):
        words = rng.choice(len(vocab), size=512, p=zipf)
        sents = [" ".join(vocab[w] for w in words[j:j + 16]).capitalize() + "." for j in range(0, 512, 16)]
        return {"id": f"doc{i}", "text": " ".join(sents), "metadata": {"source": "bench"}}

    docs = [make_doc(i) for i in range(args.ingest_docs)]
    cfg = EncoderConfig(hidden=args.dim, layers=12, intermediate=4 * args.dim, heads=args.dim // 64) if args.dim >= 768 else \
        EncoderConfig(hidden=args.dim)   # bge-base / bge-large shapes: heads of 64 (12 x 64 = 768, 16 x 64 = 1024)
    bm25 = BM25SparseEncoder(sparse_dim=SPARSE_DIM).fit(d["text"] for d in docs)
    enc = SentenceEncoder(cfg, device=dev, sparse_encoder=bm25, max_len=256, batch_size=128)

    class HostHop:  # the same encoder without the device entry points: index_chunks takes the host path
        def __init__(self, e):
            self.e = e
        encode_semantic = lambda self, t: self.e.encode_semantic(t)
        encode_semantic_batch = lambda self, ts: self.e.encode_semantic_batch(ts)
        encode_domain = lambda self, t, d=None: self.e.encode_domain(t, d)
        encode_sparse = lambda self, t: self.e.encode_sparse(t)
        encode_sparse_query = lambda self, t: self.e.encode_sparse_query(t)

    out = {}
    for name, gen in (("device_to_device", enc), ("host_hop", HostHop(enc))):
        initialize_caches()
        pipe = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False), semantic_dim=args.dim,
                                   sparse_dim=SPARSE_DIM, enable_domain=False, dtype="float16")
        pipe.index_manager.embedding_generator = gen
        tm = {"encode": 0.0, "append": 0.0, "flush": 0.0}
        chunks = 0

        async def run():
            nonlocal chunks
            for b0 in range(0, len(docs), args.ingest_batch):
                rep_ = await pipe.ingest_documents(docs[b0:b0 + args.ingest_batch])
                summ = rep_["indexing_summary"]
                assert not summ["errors"], summ["errors"][:2]
                chunks += summ["indexed_semantic"]
                for k in tm:
                    tm[k] += summ["timing_ms"][k]

        with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
            asyncio.run(pipe.ingest_documents(docs[:16]))  # warm-up (kernels, workspaces); these rows stay in the shard
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            asyncio.run(run())
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        rows = pipe.index_manager.get_collection_stats("semantic_index")["num_entities"]
        out[name] = {"docs_per_s": len(docs) / wall, "chunks_per_s": chunks / wall, "wall_s": wall, "chunks": chunks,
                     "rows_in_shard": rows, "encoder_ms": tm["encode"], "append_ms": tm["append"], "flush_ms": tm["flush"],
                     "host_pipeline_ms": wall * 1e3 - tm["encode"] - tm["append"] - tm["flush"]}
        asyncio.run(pipe.close())
    # the sentence encoder's forward by itself (reference hook: indexing.py:610-620 encode_semantic / :580-587 the batch form):
    # 1 024 sequences x 512 tokens of synthetic ids, HIP events around the forwards, arithmetic = per token and layer
    # 24 H^2 (projections + FFN) + 4 T H (attention products)
    T_enc, n_enc = 512, 1024
    big = SentenceEncoder(cfg, device=dev, max_len=512, batch_size=128)
    ids = torch.randint(1000, 30000, (n_enc, T_enc), device=dev)
    ids[:, 0] = 101
    types = torch.zeros((n_enc, T_enc), dtype=torch.long, device=dev)
    mask = torch.ones((n_enc, T_enc), dtype=torch.bool, device=dev)
    with torch.inference_mode():
        for _ in range(2):
            big.module(ids, types, mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            big.module(ids, types, mask)
        e1.record()
        e1.synchronize()
    enc_ms = e0.elapsed_time(e1) / 5
    Hd, Ld = cfg.hidden, cfg.layers
    enc_flop = n_enc * Ld * (2 * T_enc * (4 * Hd * Hd + 2 * Hd * cfg.intermediate) + 4 * T_enc * T_enc * Hd)
    enc_tf = enc_flop / (enc_ms * 1e-3) / 1e12
    sentence_encoder = {"model": f"random-init BERT hidden {Hd} x {Ld} layers, {cfg.heads} heads x {Hd // cfg.heads}, fp16",
                        "sequences": n_enc, "seq_len": T_enc, "forward_ms": enc_ms, "flop_per_forward": enc_flop,
                        "achieved": enc_tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": enc_tf / MFMA_F16_PEAK_TFLOPS,
                        "tokens_per_s": n_enc * T_enc / (enc_ms * 1e-3),
                        "kernels": ("hand-written layer kernels (encoder_layer.h) + HIP attention" if all(
                            getattr(l, "_fused_ok")(torch.empty((1, T_enc, Hd), dtype=torch.float16, device=dev), torch.ones(1, dtype=torch.int32, device=dev), Hd // cfg.heads, T_enc)
                            for l in big.module.encoder.layers) else
                            "hipBLASLt GEMMs (bias + GELU epilogue) + HIP attention (head dim 64: attention.h) + HIP add + LayerNorm")}
    print(json.dumps({"metric": "ingest_docs_per_sec", "value": out["device_to_device"]["docs_per_s"], "unit": "docs/s",
                      "n_gpus": 1, "data": "synthetic", "dtype": "f16",
                      "config": {"workload": f"{len(docs)} documents x ~512 tokens -> AdvancedRAGPipeline.ingest_documents, "
                                             f"SentenceEncoder random-init hidden {args.dim} (PyTorch-ROCm fp16) + BM25, "
                                             f"{args.ingest_batch} documents per call, fp16 shard"},
                      "paths": out, "sentence_encoder": sentence_encoder,
                      "note": "encoder_ms includes the host-side sparse (BM25) payloads of the call; host_pipeline_ms = "
                              "diagnostics + chunking + enrichment in Python"}))


def ce_report(ce, events, pairs: int, T: int, alone_ms: float = 0.0):
    """Forward pass of the cross-encoder leg (BASELINE config 4's rerank 20 -> 5; north_star assigns it to PyTorch-ROCm):
    device ms per step from events around the forward, its arithmetic (per token and layer 24 H^2 for the projections and
    the FFN + 4 T H for the attention products) and the share of the fp16 MFMA peak."""
    cfg = ce.config
    H, L = cfg.hidden, cfg.layers
    flops = pairs * ce.flops_per_pair(T)   # the arithmetic the forward EXECUTES: last layer = keys / values + token 0 (encoders.py)
    ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else 0.0
    tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    alone = {"forward_alone_ms": alone_ms, "frac_alone": flops / (alone_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS} if alone_ms > 0 else {}
    return {"model": f"random-init MiniLM-L{L}-H{H} (hand-written HIP layer kernels, fp16)", "pairs_per_step": pairs, "seq_len": T,
            "forward_ms_per_step": ms, "flop_per_step": flops, **alone,
            "flop_per_step_if_every_layer_ran_over_every_token": pairs * ce.flops_per_pair(T, executed=False),
            "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS, "forwards_timed": len(events)}


def measure_latency(h, Q, SQ, args, use_sparse, world=1, rank=0, first_row=0, n_total=None):
    """p50/p95 wall time of `await HybridRetriever.retrieve(query, profile_hint="default")` for single
    queries: host buffers in, Python dicts out (string ids), both searches + device RRF.  With world > 1 the
    searches fan out over the ranks (ranks > 0 serve) and the times are rank 0's."""
    import asyncio
    from advanced_rag.constants import RetrievalConstants
    from advanced_rag.indexing import MilvusIndexManager
    from advanced_rag.retrieval import HybridRetriever, RetrievalConfig

    mgr = MilvusIndexManager(semantic_dim=args.dim, sparse_dim=SPARSE_DIM, connect=False)
    if world > 1:
        mgr.attach_shards([h], rows_of=[np.arange(h.num_rows)], synthetic_rows=n_total, process_group=True,
                          first_row=first_row)
        if rank != 0:
            mgr.serve()
            return None
    else:
        mgr.attach_shards([h], synthetic_rows=h.num_rows)
    flatQ = Q.reshape(-1, args.dim)
    flatS = [s for batch in SQ for s in batch]

    class Gen:
        run_inline = True   # table lookups: the manager calls them on the event loop instead of hopping to its thread pool

        def encode_semantic(self, text):
            return flatQ[int(text[1:])]

        def encode_sparse(self, text):
            qi, qv = flatS[int(text[1:])]
            return {"indices": qi.tolist(), "values": qv.tolist()}

        def encode_domain(self, text, domain=None):
            return np.zeros(768, np.float32)

    mgr.embedding_generator = Gen()
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    retr = HybridRetriever(mgr, RetrievalConfig(top_k=args.top_k))
    n = min(args.latency_queries, flatQ.shape[0])

    async def run():
        lat = []
        for i in range(n + 10):
            t0 = time.perf_counter()
            out = await retr.retrieve(f"q{i % flatQ.shape[0]}", profile_hint="default")
            dt = (time.perf_counter() - t0) * 1e3
            assert len(out) == args.top_k
            if i >= 10:
                lat.append(dt)
        return lat

    lat = asyncio.run(run())

    # filtered retrieve(): the expression is evaluated over every row on the first request and the boolean mask is kept
    # per (expression, rows, tombstone epoch); every request packs and uploads N/8 bytes and the scans test the bit
    async def run_filtered():
        t0 = time.perf_counter()
        out = await retr.retrieve("q0", filters={"chunk_index": {"$lt": 5}}, profile_hint="default")
        first = (time.perf_counter() - t0) * 1e3
        assert len(out) == args.top_k and all(int(r["id"].split("::")[1]) < 5 for r in out)
        latf = []
        for i in range(min(n, 50)):
            t0 = time.perf_counter()
            await retr.retrieve(f"q{i % flatQ.shape[0]}", filters={"chunk_index": {"$lt": 5}}, profile_hint="default")
            latf.append((time.perf_counter() - t0) * 1e3)
        return first, latf

    first_f, lat_f = asyncio.run(run_filtered()) if world == 1 else (None, None)
    mask_eval_ms = None
    if world == 1:   # what a NEW expression costs once the column is in HBM: one hr_filter_eval_dev launch + an 8-byte read-back
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mgr._global_device_mask("chunk_index < 3")
        torch.cuda.synchronize()
        mask_eval_ms = (time.perf_counter() - t0) * 1e3

    # the same queries through the whole public entry point, AdvancedRAGPipeline.retrieve(): query rewriting, the
    # retriever, the default rerank branch (20 -> 5), evaluation and the audit trail (SURVEY.md section 8d's latency
    # definition); its prints (SLA / risk warnings) must not reach stdout, which carries the ONE JSON line
    import contextlib
    from advanced_rag import AdvancedRAGPipeline, PipelineConfig
    pipe = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(top_k=args.top_k))
    pipe.index_manager = mgr
    pipe.retriever.index_manager = mgr

    async def run_pipeline():
        lat2 = []
        for i in range(n + 10):
            t0 = time.perf_counter()
            res, _metrics = await pipe.retrieve(f"q{i % flatQ.shape[0]}", context={"retrieval_profile": "default"})
            dt = (time.perf_counter() - t0) * 1e3
            assert 0 < len(res) <= pipe.config.rerank_top_k
            if i >= 10:
                lat2.append(dt)
        return lat2

    with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):  # the pipeline prints a line per SLA / risk warning
        lat2 = asyncio.run(run_pipeline())

    # ---- the drop-in API under the reference's concurrency model: 64 / 128 in-flight pipeline.retrieve() coroutines on
    # one event loop (service.py:136,149), with the batching front of MilvusIndexManager.search and without it
    api = None
    if world == 1 and not args.no_api_concurrent:
        api = {}
        for coalesce in (True, False):
            m2 = MilvusIndexManager(semantic_dim=args.dim, sparse_dim=SPARSE_DIM, connect=False, coalesce=coalesce)
            m2.attach_shards([h], synthetic_rows=h.num_rows)
            m2.embedding_generator = Gen()
            pipe2 = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(top_k=args.top_k))
            pipe2.index_manager = m2
            pipe2.retriever.index_manager = m2

            async def burst(n_total, in_flight):
                sem = asyncio.Semaphore(in_flight)
                lat3 = []

                async def one(i):
                    async with sem:
                        t0 = time.perf_counter()
                        res, _m = await pipe2.retrieve(f"q{i % flatQ.shape[0]}", context={"retrieval_profile": "default"})
                        lat3.append((time.perf_counter() - t0) * 1e3)
                        assert 0 < len(res) <= pipe2.config.rerank_top_k
                t0, c0 = time.perf_counter(), time.process_time()
                await asyncio.gather(*[one(i) for i in range(n_total)])
                return lat3, time.perf_counter() - t0, time.process_time() - c0

            for in_flight in (64, 128):
                n_total = (16 if coalesce else 2) * in_flight
                with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
                    asyncio.run(burst(in_flight, in_flight))  # warm-up (embedding cache, workspaces)
                    st0 = dict(m2._front.stats) if coalesce and m2._front else None
                    lat3, wall, cpu = asyncio.run(burst(n_total, in_flight))
                ent = {"requests": n_total, "qps": n_total / wall, "p50_ms": float(np.percentile(lat3, 50)),
                       "p95_ms": float(np.percentile(lat3, 95)), "process_cpu_us_per_request": cpu / n_total * 1e6}
                if st0 is not None:
                    st1 = m2._front.stats
                    scans = (st1["dense_launches"] + st1["sparse_launches"]) - (st0["dense_launches"] + st0["sparse_launches"])
                    ent.update({"scan_launches_per_request": scans / n_total,
                                "mean_queries_per_scan_launch": 2 * n_total / max(scans, 1),
                                "front_busy_frac": (st1["busy_s"] - st0["busy_s"]) / wall,
                                "redone_unproven": st1["redone_unproven"] - st0["redone_unproven"],
                                "rounds": st1["rounds"] - st0["rounds"]})
                else:
                    ent["scan_launches_per_request"] = 2.0
                api.setdefault(f"in_flight_{in_flight}", {})["coalesced" if coalesce else "uncoalesced"] = ent
            if coalesce and m2._front is not None:
                m2._front.close()
            m2.embedding_executor.shutdown(wait=False)
        api["note"] = ("AdvancedRAGPipeline.retrieve() coroutines on one event loop, host buffers in, Python result objects out; "
                       "engine `value` above is the same kernels fed with device-resident pre-batched queries")
    # ---- the same entry points with the QUERY ENCODER in the path (reference: 10 - 20 ms of a request's budget,
    # ARCHITECTURE.md:321-328; indexing.py:601-627 embeds each request alone): a random-init sentence encoder of the shard's
    # width (bge-base shape at 768), ~32-token query texts that are all cache misses; the batching front encodes the misses
    # of a round in ONE forward and hands the rows to the search without a host hop
    with_encoder = None
    if world == 1 and not args.no_api_concurrent:
        import torch
        from advanced_rag.encoders import EncoderConfig, SentenceEncoder
        ecfg = (EncoderConfig(hidden=args.dim, layers=12, heads=args.dim // 64, intermediate=4 * args.dim) if args.dim >= 768
                else EncoderConfig(hidden=args.dim))
        enc = SentenceEncoder(ecfg, device="cuda:0", max_len=64, batch_size=256)
        words = ("retrieval augmented generation pipeline vector index sparse dense hybrid fusion rerank latency throughput shard "
                 "memory bandwidth kernel query document chunk embedding filter metadata timestamp entropy redundancy").split()

        def qtext(i):   # ~30 tokens, distinct per i, ends in the number that keys the sparse table
            r = np.random.default_rng(i)
            return " ".join(r.choice(words, 24)) + f" about item number q{i}"

        class EncGen:
            run_inline = True           # the sparse side is a table lookup; the dense side goes through the batching front
            encode_to_device = enc.encode_to_device
            encode_semantic = enc.encode_semantic
            encode_semantic_batch = enc.encode_semantic_batch

            def encode_sparse(self, text):
                qi, qv = flatS[int(text.rsplit("q", 1)[1]) % len(flatS)]
                return {"indices": qi.tolist(), "values": qv.tolist()}

            def encode_domain(self, text, domain=None):
                return np.zeros(768, np.float32)

        m3 = MilvusIndexManager(semantic_dim=args.dim, sparse_dim=SPARSE_DIM, connect=False, device_embedding_cache=1 << 16)
        m3.attach_shards([h], synthetic_rows=h.num_rows)
        m3.embedding_generator = EncGen()
        pipe3 = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(top_k=args.top_k))
        pipe3.index_manager = m3
        pipe3.retriever.index_manager = m3
        retr3 = HybridRetriever(m3, RetrievalConfig(top_k=args.top_k))
        counter = [0]

        async def seq_enc(k):
            lat_e = []
            for _ in range(k + 5):
                counter[0] += 1
                t0 = time.perf_counter()
                out = await retr3.retrieve(qtext(counter[0]), profile_hint="default")
                dt = (time.perf_counter() - t0) * 1e3
                assert len(out) == args.top_k
                lat_e.append(dt)
            return lat_e[5:]

        async def burst_enc(n_total, in_flight):
            sem = asyncio.Semaphore(in_flight)
            lat3 = []

            async def one():
                async with sem:
                    counter[0] += 1
                    text = qtext(counter[0])
                    t0 = time.perf_counter()
                    res, _m = await pipe3.retrieve(text, context={"retrieval_profile": "default"})
                    lat3.append((time.perf_counter() - t0) * 1e3)
                    assert 0 < len(res) <= pipe3.config.rerank_top_k
            t0, c0 = time.perf_counter(), time.process_time()
            await asyncio.gather(*[one() for _ in range(n_total)])
            return lat3, time.perf_counter() - t0, time.process_time() - c0

        with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
            lat_e = asyncio.run(seq_enc(min(n, 100)))
            asyncio.run(burst_enc(64, 64))
            f0, st0 = enc.forwards, dict(m3._front.stats)
            lat3, wall, cpu = asyncio.run(burst_enc(16 * 64, 64))
        st1 = m3._front.stats
        with_encoder = {
            "encoder": f"random-init BERT hidden {ecfg.hidden} x {ecfg.layers} layers (fp16), ~30-token queries, every request a cache miss",
            "p50_retrieve_ms_with_encoder": float(np.percentile(lat_e, 50)), "p95_retrieve_ms_with_encoder": float(np.percentile(lat_e, 95)),
            "in_flight_64": {"requests": 16 * 64, "qps": 16 * 64 / wall, "p50_ms": float(np.percentile(lat3, 50)),
                             "p95_ms": float(np.percentile(lat3, 95)), "process_cpu_us_per_request": cpu / (16 * 64) * 1e6,
                             "encoder_forwards": enc.forwards - f0,
                             "texts_per_encoder_forward": (st1["encoded_texts"] - st0["encoded_texts"]) / max(1, enc.forwards - f0),
                             "rounds": st1["rounds"] - st0["rounds"]}}
        m3._front.close()
        m3.embedding_executor.shutdown(wait=False)
    if world > 1:
        mgr.stop_workers()
    mgr.embedding_executor.shutdown(wait=False)
    return {"retrieve_shards": world, **({"with_query_encoder": with_encoder,
                                          "p50_retrieve_ms_with_encoder": with_encoder["p50_retrieve_ms_with_encoder"]} if with_encoder else {}),
            "p50_retrieve_ms": float(np.percentile(lat, 50)), "p95_retrieve_ms": float(np.percentile(lat, 95)),
            "p50_pipeline_retrieve_ms": float(np.percentile(lat2, 50)),
            "p95_pipeline_retrieve_ms": float(np.percentile(lat2, 95)), "latency_queries": n,
            **({"filtered_retrieve": {"filter": "chunk_index < 5 (half of the rows)", "first_request_ms": first_f,
                                      "new_expression_mask_eval_ms": mask_eval_ms,
                                      "p50_ms": float(np.percentile(lat_f, 50)), "p95_ms": float(np.percentile(lat_f, 95)),
                                      "queries": len(lat_f)}} if lat_f else {}),
            **({"api_concurrent": api} if api else {})}


if __name__ == "__main__":
    main()
