"""The batched device path (engine.HybridSearchEngine: dense + sparse + RRF + rerank, all HIP
kernels on one stream) against the oracle, and the cross-shard merge with the all-gather buffer
layout: two shards on one GPU stand in for two ranks."""
import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat
from advanced_rag.engine import EngineConfig, HybridSearchEngine, ListPack, pack_sparse_queries, shard_range

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def corpus(n, d, V, nnz, B, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float16)
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    Q = rng.standard_normal((B, d)).astype(np.float32)
    SQ = [(np.sort(rng.choice(V, 3 * nnz, replace=False)).astype(np.int32),
           np.abs(rng.standard_normal(3 * nnz)).astype(np.float32)) for _ in range(B)]
    return X, ptr, idx, val, Q, SQ


def oracle_pipeline(X, ptr, idx, val, Q, SQ, cfg):
    kp = 2 * cfg.top_k
    di, ds = oracle.dense_search(X, Q, kp, oracle.COSINE)
    si, ss = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    fused, reranked = [], []
    for b in range(Q.shape[0]):
        fi, fs, fm = oracle.rrf(di[b], si[b][si[b] >= 0], (), cfg.dense_weight, cfg.sparse_weight, 0.2, cfg.rrf_k)
        fi, fs, fm = fi[:cfg.top_k], fs[:cfg.top_k], fm[:cfg.top_k]
        fused.append((fi, fs, fm))
        new = [cfg.base_weight * float(s) + cfg.method_bonus * float(bin(int(m)).count("1")) + cfg.recency_weight * 0.0
               for s, m in zip(fs, fm)]
        order = sorted(range(len(new)), key=lambda i: new[i], reverse=True)[:cfg.rerank_top_k]  # stable, like list.sort
        reranked.append((fi[order], np.array([new[i] for i in order]), fs[order]))
    return (di, ds), (si, ss), fused, reranked


@pytest.mark.parametrize("n,d,V,nnz,B,top_k", [(6000, 128, 800, 12, 9, 20), (20000, 768, 10000, 100, 64, 20),
                                                (300, 64, 100, 5, 3, 50),
                                                (60000, 256, 2000, 30, 70, 100)])   # k' = 200 = HR_MAX_TOPK, C = 304 groups
def test_engine_single_shard_matches_oracle(gpu, n, d, V, nnz, B, top_k):
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=n)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=top_k)
    eng = HybridSearchEngine(h, cfg)
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)))
    torch.cuda.synchronize()
    (di, ds), (si, ss), fused, reranked = oracle_pipeline(X, ptr, idx, val, Q, SQ, cfg)
    assert out["flags"].min().item() == 1
    assert np.array_equal(out["ids"][0].cpu().numpy(), di) and np.array_equal(out["ids"][1].cpu().numpy(), si)
    assert np.array_equal(out["scores"][0].cpu().numpy().view(np.uint32), ds.view(np.uint32))
    assert np.array_equal(out["scores"][1].cpu().numpy().view(np.uint32), ss.view(np.uint32))
    for b in range(B):
        fi, fs, fm = fused[b]
        nf = int(out["fused_n"][b])
        assert nf == len(fi)
        assert np.array_equal(out["fused_ids"][b, :nf].cpu().numpy(), fi)
        assert np.array_equal(out["fused_scores"][b, :nf].cpu().numpy().view(np.uint64), fs.view(np.uint64))
        assert np.array_equal(out["fused_methods"][b, :nf].cpu().numpy(), fm)
        ri, rs, ro = reranked[b]
        nr = len(ri)
        assert np.array_equal(out["rr_ids"][b, :nr].cpu().numpy(), ri)
        assert np.array_equal(out["rr_scores"][b, :nr].cpu().numpy().view(np.uint64), rs.view(np.uint64))
        assert np.array_equal(out["rr_orig"][b, :nr].cpu().numpy().view(np.uint64), ro.view(np.uint64))
    h.close()


def test_two_shards_merge_equals_global(gpu):
    n, d, V, nnz, B, kp = 9000, 96, 600, 10, 7, 40
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=5)
    X[17] = X[8000]  # cross-shard tie
    world = 2
    layout = ListPack(2, B, kp)
    gathered = torch.zeros((world, layout.nbytes), dtype=torch.uint8, device="cuda")
    handles = []
    dq = torch.from_numpy(Q).cuda()
    p, i_, v_, mx = pack_sparse_queries(SQ, 0.2)
    dp, di_, dv_ = torch.from_numpy(p).cuda(), torch.from_numpy(i_).cuda(), torch.from_numpy(v_).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        lo, hi = shard_range(n, r, world, align=64)
        h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
        h.set_row_offset(lo)
        h.add_dense(X[lo:hi])
        h.add_sparse(ptr[lo:hi + 1] - ptr[lo], idx[ptr[lo]:ptr[hi]], val[ptr[lo]:ptr[hi]])
        h.finalize()
        ids, scores = layout.views(gathered[r])
        h.search_dense_dev(dq.data_ptr(), B, kp, ids[0].data_ptr(), scores[0].data_ptr(), 0, 0, st)
        h.search_sparse_dev(dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), B, len(i_), mx, kp, ids[1].data_ptr(),
                            scores[1].data_ptr(), 0, 0, st)
        handles.append(h)
    gd = oracle.dense_search(X, Q, kp, oracle.COSINE)
    gs = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    for m, (want_i, want_s) in enumerate((gd, gs)):
        oi = torch.empty((B, kp), dtype=torch.int64, device="cuda")
        os_ = torch.empty((B, kp), dtype=torch.float32, device="cuda")
        sc_off, id_off, sc_stride, id_stride = layout.merge_args(m)
        nat.merge_topk_dev(gathered.data_ptr() + sc_off, gathered.data_ptr() + id_off, world, B, kp, kp, oi.data_ptr(),
                           os_.data_ptr(), st, score_stride=sc_stride, id_stride=id_stride)
        torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy(), want_i)
        assert np.array_equal(os_.cpu().numpy().view(np.uint32), want_s.view(np.uint32))
    for h in handles:
        h.close()


def test_index_manager_and_retriever_on_gpu(gpu):
    """The reference-shaped API end to end on the device: index_chunks -> retrieve (device RRF) -> filters -> delete."""
    import asyncio
    from advanced_rag import (AdvancedRAGPipeline, BM25SparseEncoder, HybridRetriever, MilvusIndexManager, PipelineConfig,
                              RetrievalConfig)
    from advanced_rag.constants import RetrievalConstants
    from advanced_rag.embedding_cache import initialize_caches
    from advanced_rag.retrieval import rrf_rank_lists

    initialize_caches()
    docs = [{"id": f"d{i}", "text": f"Document {i} talks about topic{i % 7} and widget{i % 5}. " * 3 + "Shared filler sentence here.",
             "metadata": {"source": "unit"}} for i in range(60)]
    rng = np.random.default_rng(0)
    table = {}

    class Gen:
        def __init__(self):
            self.bm25 = BM25SparseEncoder(sparse_dim=2048)

        def encode_semantic(self, text):
            if text not in table:
                table[text] = rng.standard_normal(64).astype(np.float32)
            return table[text]

        def encode_sparse(self, text):
            return self.bm25.encode_document(text)

        def encode_sparse_query(self, text):
            return self.bm25.encode_query(text)

        def encode_domain(self, text, domain=None):
            return self.encode_semantic(text)[:32].copy()

    gen = Gen()
    p = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False), semantic_dim=64, sparse_dim=2048,
                            domain_dim=32)
    p.index_manager.embedding_generator = gen
    gen.bm25.fit(d["text"] for d in docs)
    report = asyncio.run(p.ingest_documents(docs))
    summ = report["indexing_summary"]
    assert summ["indexed_semantic"] == summ["total_chunks"] == summ["indexed_sparse"] == summ["indexed_domain"] and not summ["errors"]
    mgr = p.index_manager
    assert mgr.get_collection_stats("semantic_index")["num_entities"] == summ["total_chunks"]
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 30.0
    try:
        retr = HybridRetriever(mgr, RetrievalConfig(top_k=10))
        query = mgr._cols["content"][7]
        out = asyncio.run(retr.retrieve(query, profile_hint="default"))
        assert out[0]["id"] == mgr._cols["id"][7] and set(out[0]["retrieval_methods"]) == {"semantic", "sparse"}
        # device RRF == host RRF on the same lists
        sem = asyncio.run(mgr.search(gen.encode_semantic(query), "semantic_index", 20))
        sp = asyncio.run(mgr.search(gen.encode_sparse_query(query), "sparse_index", 20, None,
                                    {"metric_type": "IP", "params": {"drop_ratio_search": 0.2}}))
        host = rrf_rank_lists([[h["id"] for h in sem], [h["id"] for h in sp], []], [0.7, 0.3, 0.2], 60)
        assert [o["id"] for o in out] == [h[0] for h in host][:10]
        assert [o["score"] for o in out] == [h[1] for h in host][:10]
        # filter expression -> row mask
        doc7 = mgr._cols["doc_id"][7]
        flt = asyncio.run(retr.retrieve(query, filters={"doc_id": doc7}, profile_hint="default"))
        assert flt and all(o["metadata"]["doc_id"] == doc7 for o in flt)
        # tombstones
        asyncio.run(mgr.delete_by_filter("semantic_index", f'doc_id == "{doc7}"'))
        gone = asyncio.run(retr.retrieve(query, profile_hint="default"))
        assert all(o["metadata"]["doc_id"] != doc7 for o in gone)
        results, metrics = asyncio.run(p.retrieve("topic3 widget2", context={"retrieval_profile": "default"}))
        assert len(results) == 5 and results[0].chunk_id
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old
        asyncio.run(p.close())


def test_pipelined_engine_matches_sequential(gpu):
    """Two batches in flight (scans on the heavy stream, finish on the light stream) give the same
    lists, fused results and rerank output as one batch at a time."""
    from advanced_rag.engine import PipelinedSearchEngine
    n, d, V, nnz, B = 30000, 128, 1000, 10, 16
    X, ptr, idx, val, _, _ = corpus(n, d, V, nnz, B, seed=21)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=20)
    seq = HybridSearchEngine(h, cfg)
    pipe = PipelinedSearchEngine(h, cfg, depth=2)
    rng = np.random.default_rng(3)
    batches = []
    for _ in range(5):
        Q = rng.standard_normal((B, d)).astype(np.float32)
        SQ = [(np.sort(rng.choice(V, 30, replace=False)).astype(np.int32), np.abs(rng.standard_normal(30)).astype(np.float32))
              for _ in range(B)]
        batches.append((torch.from_numpy(Q).cuda(), seq.upload_sparse(pack_sparse_queries(SQ, 0.2))))
    want = []
    for q, sq in batches:
        o = seq.search(q, sq)
        torch.cuda.synchronize()
        want.append({k: o[k].clone() for k in ("ids", "scores", "fused_ids", "fused_scores", "fused_methods", "rr_ids", "rr_scores")})
    got = []
    for q, sq in batches:  # submit everything without waiting; copy results out in stream order
        o = pipe.submit(q, sq)
        with torch.cuda.stream(pipe.light):
            got.append({k: o[k].clone() for k in want[0]})
    pipe.synchronize()
    assert pipe.all_flags_exact()
    for w, g in zip(want, got):
        for k in w:
            assert torch.equal(w[k], g[k]), k
    h.close()


def test_snapshot_roundtrip(gpu, tmp_path):
    """hr_save / hr_load: a reloaded shard answers exactly like the original (dense, sparse, masks, offsets)."""
    n, d, V, nnz, B = 5000, 200, 700, 9, 6
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=77)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.set_row_offset(1000)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    path = str(tmp_path / "shard.hbmrag")
    h.save(path)
    g = nat.ShardHandle.load(path, d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    assert (g.num_rows, g.num_sparse_rows) == (n, n)
    mask = np.packbits(np.random.default_rng(1).random(n) < 0.5, bitorder="little")
    for m in (None, mask):
        a, b = h.search_dense(Q, 40, m), g.search_dense(Q, 40, m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        a, b = h.search_sparse(SQ, 40, 0.2, m), g.search_sparse(SQ, 40, 0.2, m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert a[0].min() >= -1 and a[0][a[0] >= 0].min() >= 1000          # row offset survived
    g.add_dense(X[:10])                                                  # a loaded shard keeps accepting rows
    g.add_sparse(ptr[:11], idx[:ptr[10]], val[:ptr[10]])
    g.finalize()
    assert g.num_rows == n + 10
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.hbmrag"
        bad.write_bytes(b"not a snapshot")
        nat.ShardHandle.load(str(bad), d)
    h.close()
    g.close()


def test_index_manager_snapshot(gpu, tmp_path):
    import asyncio
    from advanced_rag import MilvusIndexManager
    rng = np.random.default_rng(5)
    m = MilvusIndexManager(semantic_dim=64, sparse_dim=500, domain_dim=32)
    X = rng.standard_normal((300, 64)).astype(np.float32)
    sp = (np.arange(301, dtype=np.int64) * 4, np.sort(rng.integers(0, 125, (300, 4)) + np.arange(4) * 125, axis=1).astype(np.int32).reshape(-1),
          np.abs(rng.standard_normal(1200)).astype(np.float32))
    m.add_rows(X, sp, ids=[f"c{i}" for i in range(300)], contents=[f"text {i}" for i in range(300)],
               entropy=rng.random(300).tolist())
    m.finalize()
    asyncio.run(m.delete_by_filter("semantic_index", 'chunk_id == "c7"'))
    q = X[7] + 0.01
    before = asyncio.run(m.search(q, "semantic_index", 5, filters="entropy >= 0.3"))
    m.save_snapshot(str(tmp_path / "snap"))
    m2 = MilvusIndexManager(semantic_dim=64, sparse_dim=500, domain_dim=32)
    m2.load_snapshot(str(tmp_path / "snap"))
    after = asyncio.run(m2.search(q, "semantic_index", 5, filters="entropy >= 0.3"))
    assert [(h["id"], h["score"], h["content"]) for h in before] == [(h["id"], h["score"], h["content"]) for h in after]
    assert all(h["id"] != "c7" for h in after)
    asyncio.run(m.close())
    asyncio.run(m2.close())


def test_engine_resolves_unproven_lists(gpu):
    """Ties at the candidate cut: the batched path flags them, resolve_inexact() repairs lists + fusion to the oracle."""
    rng = np.random.default_rng(31)
    d, V = 64, 300
    proto = rng.standard_normal(d).astype(np.float16)
    X = np.concatenate([rng.standard_normal((2000, d)).astype(np.float16), np.tile(proto, (5000, 1)),
                        rng.standard_normal((1000, d)).astype(np.float16)])
    n = X.shape[0]
    idx = np.sort(np.argpartition(rng.random((n, V)), 5, axis=1)[:, :6], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 6)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * 6
    Q = np.stack([proto.astype(np.float32)] + [rng.standard_normal(d).astype(np.float32) for _ in range(3)])
    SQ = [(np.sort(rng.choice(V, 20, replace=False)).astype(np.int32), np.abs(rng.standard_normal(20)).astype(np.float32))
          for _ in range(4)]
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=20)
    eng = HybridSearchEngine(h, cfg)
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)))
    torch.cuda.synchronize()
    assert out["flags"][0, 0].item() == 0
    assert eng.resolve_inexact(out, Q, SQ, 0.2) >= 1
    torch.cuda.synchronize()
    (di, ds), (si, ss), fused, reranked = oracle_pipeline(X, ptr, idx, val, Q, SQ, cfg)
    assert np.array_equal(out["ids"][0].cpu().numpy(), di) and np.array_equal(out["ids"][1].cpu().numpy(), si)
    for b in range(4):
        nf = int(out["fused_n"][b])
        assert np.array_equal(out["fused_ids"][b, :nf].cpu().numpy(), fused[b][0])
        assert np.array_equal(out["rr_ids"][b, :len(reranked[b][0])].cpu().numpy(), reranked[b][0])
    h.close()


def test_config3_1m_768_hybrid_matches_oracle(gpu):
    """BASELINE config 3: 1M x 768 (bge-base shape) hybrid dense + sparse + RRF (weights 0.7/0.3), 1 GPU.
    Lists, fused ids and fused scores against the oracle for a few queries of a 64-query batch."""
    from concurrent.futures import ThreadPoolExecutor
    n, d, V, nnz, B = 1_000_000, 768, 10000, 100, 64

    def block(b):
        rng = np.random.default_rng(1234 + b)
        x = rng.standard_normal((100_000, d), dtype=np.float32).astype(np.float16)
        idx = ((np.arange(nnz, dtype=np.int32) * (V // nnz))[None, :] + rng.integers(0, V // nnz, (100_000, nnz), dtype=np.int32))
        val = np.abs(rng.standard_normal((100_000, nnz), dtype=np.float32))
        return x, idx.reshape(-1), val.reshape(-1)

    with ThreadPoolExecutor(5) as pool:
        parts = list(pool.map(block, range(10)))
    X = np.concatenate([p[0] for p in parts])
    idx = np.concatenate([p[1] for p in parts])
    val = np.concatenate([p[2] for p in parts])
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    rng = np.random.default_rng(4321)
    Q = rng.standard_normal((B, d), dtype=np.float32)
    SQ = [(np.sort(rng.choice(V, nnz, replace=False)).astype(np.int32), np.abs(rng.standard_normal(nnz)).astype(np.float32))
          for _ in range(B)]
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=20)
    eng = HybridSearchEngine(h, cfg)
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)))
    torch.cuda.synchronize()
    assert out["flags"].min().item() == 1
    pick = [0, 31, 63]
    (di, ds), (si, ss), fused, reranked = oracle_pipeline(X, ptr, idx, val, Q[pick], [SQ[i] for i in pick], cfg)
    for j, b in enumerate(pick):
        assert np.array_equal(out["ids"][0, b].cpu().numpy(), di[j]) and np.array_equal(out["ids"][1, b].cpu().numpy(), si[j])
        assert np.array_equal(out["scores"][0, b].cpu().numpy().view(np.uint32), ds[j].view(np.uint32))
        assert np.array_equal(out["scores"][1, b].cpu().numpy().view(np.uint32), ss[j].view(np.uint32))
        fi, fs, _ = fused[j]
        assert np.array_equal(out["fused_ids"][b].cpu().numpy(), fi)
        assert np.max(np.abs(out["fused_scores"][b].cpu().numpy() - fs)) <= 1e-4   # north-star tolerance (here: 0)
        assert np.array_equal(out["fused_scores"][b].cpu().numpy().view(np.uint64), fs.view(np.uint64))
        assert np.array_equal(out["rr_ids"][b].cpu().numpy(), reranked[j][0])
    h.close()


def test_engine_with_domain_list_matches_oracle(gpu):
    """Third modality in the batched engine (reference _search_domain, retrieval.py:397-419): a second dense shard over
    the same rows, k = top_k (not 2k), fused with weight 0.2 after the semantic and sparse lists."""
    n, d, dd, V, nnz, B, top_k = 5000, 128, 96, 600, 10, 11, 20
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=5)
    rng = np.random.default_rng(6)
    Xd = rng.standard_normal((n, dd)).astype(np.float32).astype(np.float16)
    Qd = rng.standard_normal((B, dd)).astype(np.float32)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    hd = nat.ShardHandle(dd, nat.HR_F16, nat.HR_METRIC_COSINE)
    hd.add_dense(Xd)
    hd.finalize()
    cfg = EngineConfig(top_k=top_k)
    eng = HybridSearchEngine(h, cfg, domain_handle=hd)
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)), torch.from_numpy(Qd).cuda())
    torch.cuda.synchronize()
    kp = 2 * top_k
    di, _ = oracle.dense_search(X, Q, kp, oracle.COSINE)
    si, _ = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    ci, cs = oracle.dense_search(Xd, Qd, top_k, oracle.COSINE)
    assert np.array_equal(out["dom_ids"].cpu().numpy(), ci)
    assert np.array_equal(out["dom_scores"].cpu().numpy().view(np.uint32), cs.view(np.uint32))
    assert out["dom_flags"].min().item() == 1
    saw_domain_only = False
    for b in range(B):
        fi, fs, fm = oracle.rrf(di[b], si[b][si[b] >= 0], ci[b], cfg.dense_weight, cfg.sparse_weight, 0.2, cfg.rrf_k)
        fi, fs, fm = fi[:top_k], fs[:top_k], fm[:top_k]
        nf = int(out["fused_n"][b])
        assert nf == len(fi)
        assert np.array_equal(out["fused_ids"][b, :nf].cpu().numpy(), fi)
        assert np.array_equal(out["fused_scores"][b, :nf].cpu().numpy().view(np.uint64), fs.view(np.uint64))
        assert np.array_equal(out["fused_methods"][b, :nf].cpu().numpy(), fm)
        saw_domain_only |= bool((fm & 4).any())
    assert saw_domain_only
    # without domain queries the same engine gives the two-list answer again
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)))
    torch.cuda.synchronize()
    fi, fs, fm = oracle.rrf(di[0], si[0][si[0] >= 0], (), cfg.dense_weight, cfg.sparse_weight, 0.2, cfg.rrf_k)
    assert np.array_equal(out["fused_ids"][0, :top_k].cpu().numpy(), fi[:top_k])
    with pytest.raises(ValueError):
        HybridSearchEngine(h, cfg).search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2)),
                                          torch.from_numpy(Qd).cuda())
    h.close()
    hd.close()
