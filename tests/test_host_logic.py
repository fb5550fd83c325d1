"""Host-side logic of the product package against the reference's golden outputs; the
test bodies mirror the reference's own tests (test_extended.py:81-426, :533-548,
tests/test_embedding_cache.py, test_advanced_rag.py:287-300) with asyncio.run in place
of pytest-asyncio."""
import asyncio
import json
import os

import numpy as np
import pytest

from advanced_rag import (AdvancedRAGPipeline, HybridRetriever, LearnedRanker, MilvusIndexManager, PipelineConfig,
                          PipelineStage, RetrievalConfig, CrossEncoderReranker, QueryClassifier, BM25SparseEncoder)
from advanced_rag.chunking import chunk_id_for
from advanced_rag.constants import RetrievalConstants
from advanced_rag.embedding_cache import EmbeddingCache, get_domain_cache, get_semantic_cache, get_sparse_cache
from advanced_rag.query_rewriting import QueryRewriter
from advanced_rag import filters

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def hits(ids, prefix):
    return [{"id": i, "content": f"{prefix} {i}", "score": 1.0 - 0.01 * r} for r, i in enumerate(ids)]


def test_fuse_results_matches_reference_bit_for_bit():
    for c in gold("g1_fuse.json"):
        r = HybridRetriever(index_manager=None,
                            config=RetrievalConfig(dense_weight=c["dense_weight"], sparse_weight=c["sparse_weight"]))
        out = r._fuse_results(hits(c["semantic"], "s"), hits(c["sparse"], "p"), hits(c["domain"], "d"))
        assert [o["id"] for o in out] == c["ids"], c["label"]
        assert [float(o["score"]).hex() for o in out] == c["scores"], c["label"]
        assert [sorted(o["retrieval_methods"]) for o in out] == c["methods"]
        assert [o["content"].split()[0] for o in out] == c["payload_from"]  # which list supplied the payload


def test_rerank_paths_match_reference():
    g = {c["label"]: c for c in gold("g2_rerank.json")}
    c = g["learned-ranker"]
    r = HybridRetriever(index_manager=None, config=RetrievalConfig(enable_learned_ranker=True), learned_ranker=LearnedRanker())
    out = asyncio.run(r.rerank("q", r._fuse_results(hits(c["semantic"], "s"), hits(c["sparse"], "p"), []), top_k=c["top_k"]))
    assert [o["id"] for o in out] == c["ids"]
    assert [float(o["score"]).hex() for o in out] == c["scores"]
    assert [float(o["original_retrieval_score"]).hex() for o in out] == c["original"]

    class Inject:
        async def score(self, pairs):
            return list(g["injected-stable"]["inject"])

    r = HybridRetriever(index_manager=None)
    r.reranker = Inject()
    res = [{"id": x, "content": x, "score": 0.5 - 0.1 * i} for i, x in enumerate("ABCD")]
    out = asyncio.run(r.rerank("q", res, top_k=None))
    assert [o["id"] for o in out] == g["injected-stable"]["ids"] and len(out) == g["injected-stable"]["n"]
    r = HybridRetriever(index_manager=None, config=RetrievalConfig(enable_reranking=False))
    res = [{"id": x, "content": x, "score": 0.5} for x in "ABC"]
    assert [o["id"] for o in asyncio.run(r.rerank("q", list(res), top_k=2))] == g["disabled"]["top2"]
    assert [o["id"] for o in asyncio.run(r.rerank("q", list(res)))] == g["disabled"]["none"]


def test_rerank_placeholder_and_external_scores():  # reference test_extended.py:238-273
    r = HybridRetriever(index_manager=None)
    res = [{"id": "A", "content": "alpha", "score": 0.5}, {"id": "B", "content": "bravo", "score": 0.4},
           {"id": "C", "content": "charlie", "score": 0.3}]
    out = asyncio.run(r.rerank(query="q", results=res, top_k=2))
    assert len(out) == 2 and "rerank_score" in out[0]

    class Flip:
        async def score(self, pairs):
            return [0.1, 0.9]

    r.reranker = Flip()
    out = asyncio.run(r.rerank("q", [{"id": "A", "content": "a", "score": 0.5}, {"id": "B", "content": "b", "score": 0.4}], 2))
    assert [x["id"] for x in out] == ["B", "A"]
    assert len(asyncio.run(CrossEncoderReranker().score([("q", "d")] * 3))) == 3


def test_classifier_profiles_and_defaults_match_reference():
    g = gold("g3_profiles.json")
    clf = QueryClassifier()
    for q, want in g["classify"]:
        assert clf.classify(q) == want, q
    for key, profs in g["profiles"].items():
        k, rk = map(int, key.split(","))
        hr = HybridRetriever(index_manager=None, config=RetrievalConfig(top_k=k, rerank_top_k=rk))
        assert set(hr.profiles) == set(profs)
        for name, want in profs.items():
            p = hr.profiles[name]
            got = {"top_k": p.top_k, "rerank_top_k": p.rerank_top_k, "enable_mmr": p.enable_mmr, "mmr_lambda": p.mmr_lambda,
                   "enable_reranking": p.enable_reranking, "dense_weight": p.dense_weight, "sparse_weight": p.sparse_weight}
            assert got == want, (key, name)
    assert RetrievalConstants.MAX_TOP_K == g["max_top_k"] and RetrievalConstants.TIMEOUT_SECONDS == g["timeout_seconds"]
    cfg = RetrievalConfig()
    for field, want in g["default_config"].items():
        assert getattr(cfg, field) == want, field


def test_filter_expressions_match_reference():
    hr = HybridRetriever(index_manager=None)
    for c in gold("g4_filters.json"):
        if "error" in c:
            with pytest.raises(Exception) as ei:
                hr._build_filter_expression(c["filters"])
            assert type(ei.value).__name__ == c["error"], c
        else:
            assert hr._build_filter_expression(c["filters"]) == c["expr"], c


def test_filter_expression_evaluates_to_row_mask():
    cols = {"doc_id": np.array(["a", 'q"x', "a\\b", "z"]), "entropy": np.array([0.1, 0.5, 0.9, 0.3], np.float32),
            "chunk_index": np.array([0, 1, 2, 3]), "timestamp": np.array(["2023-05-01", "2024-02-01", "2024-12-31", "2025-01-01"])}
    hr = HybridRetriever(index_manager=None)
    expr = hr._build_filter_expression({"doc_id": 'q"x', "entropy": {"$gte": 0.2}})
    assert filters.evaluate(expr, cols, 4).tolist() == [False, True, False, False]
    expr = hr._build_filter_expression({"doc_id": "a\\b"})
    assert filters.evaluate(expr, cols, 4).tolist() == [False, False, True, False]
    expr = hr._build_filter_expression({"timestamp": {"$gte": "2024-01-01", "$lt": "2025-01-01"}, "chunk_index": {"$ne": 2}})
    assert filters.evaluate(expr, cols, 4).tolist() == [False, True, False, False]
    assert filters.pack(np.array([1, 0, 1, 1, 0, 0, 0, 0, 1], bool)).tolist() == [0b00001101, 0b1]
    # the host evaluator against the oracle's explicit statement of the semantics (float32-rounded literals on FLOAT fields,
    # integer or float64 comparison on INT64 fields, UTF-8 byte order on strings), over every expression golden g4 holds
    import oracle
    rng = np.random.default_rng(3)
    n = 4000
    pool = ['doc"123', "a\\b", "a >= b", "x and y", 'q"uo\\te', "", "0123456789abcdef-tail-A", "0123456789abcdef-tail-B", "doc9", "doc95", "ünï"]
    big = {"chunk_index": rng.integers(0, 12, n), "token_count": rng.integers(0, 2000, n),
           "entropy": (rng.integers(0, 11, n) / 10).astype(np.float32), "redundancy": (rng.integers(0, 11, n) / 10).astype(np.float32),
           "domain_density": rng.random(n).astype(np.float32), "doc_id": np.array([pool[i] for i in rng.integers(0, len(pool), n)]),
           "chunk_id": np.array([f"d::{i % 3}::abcd123{i % 10}" for i in range(n)]),
           "timestamp": np.array([f"202{i % 6}-0{1 + i % 9}-1{i % 9}" for i in range(n)])}
    exprs = [c["expr"] for c in gold("g4_filters.json") if c.get("expr")]
    exprs += ['doc_id == "ünï"', "chunk_index == true and entropy != 0.30000001192092896", "token_count >= 1e3", "chunk_index >= 2.5",
              'doc_id < "0123456789abcdef-tail-B" and doc_id >= "0123456789abcdef"', "entropy >= 1"]
    for e in exprs:
        assert np.array_equal(filters.evaluate(e, big, n), oracle.filter_mask(e, big, n)), e
    for bad in ('doc_id == 5', 'chunk_index == "x"', "nofield > 1", 'doc_id == "open'):
        for fn in (filters.evaluate, oracle.filter_mask):
            with pytest.raises(ValueError):
                fn(bad, big, n)
    with pytest.raises(ValueError):
        filters.evaluate('entropy >= "x"', cols, 4)
    with pytest.raises(ValueError):
        filters.evaluate("nosuch == 1", cols, 4)
    assert filters.parse('doc_id == "a and b" and chunk_index == 3') == [("doc_id", "==", "a and b"), ("chunk_index", "==", 3)]


def test_mmr_diversification():  # reference test_extended.py:189-213
    cfg = RetrievalConfig(hybrid_alpha=0.7, top_k=3, enable_mmr=True, mmr_lambda=0.6)
    r = HybridRetriever(index_manager=None, config=cfg)
    sem = [{"id": "A", "content": "alpha alpha content one", "score": 0.95}, {"id": "B", "content": "bravo content two", "score": 0.85},
           {"id": "C", "content": "alpha content three", "score": 0.80}]
    sp = [{"id": "A", "content": "alpha alpha content one", "score": 0.75}, {"id": "D", "content": "delta unique different", "score": 0.70},
          {"id": "E", "content": "echo also different", "score": 0.65}]
    fused = r._fuse_results(semantic_results=sem, sparse_results=sp, domain_results=[])
    assert len(fused) <= 3 and fused[0]["id"] == "A"
    assert {"D", "E", "C", "B"} & {x["id"] for x in fused}


class FakeIndexManager:
    def __init__(self, with_meta=True, delay=0.0):
        self.with_meta, self.delay, self.calls = with_meta, delay, []

    async def _generate_semantic_embedding(self, text):
        if self.delay:
            await asyncio.sleep(self.delay)
        return np.ones(4, dtype=np.float32)

    async def _generate_sparse_embedding(self, text):
        return np.zeros(4, dtype=np.float32)

    async def _generate_domain_embedding(self, text, domain):
        return np.full(4, 2.0, dtype=np.float32)

    async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
        self.calls.append((collection_name, top_k, filters))
        tag = {"semantic_index": "S", "sparse_index": "P", "domain_index": "D"}[collection_name]
        hit = {"id": tag, "content": tag.lower(), "score": 0.9}
        if self.with_meta:
            hit["metadata"] = {"doc_id": "d" + tag}
        return [hit]


def test_retrieve_with_domain_profile_tagging_and_k_prime():  # reference test_extended.py:276-331 + SURVEY App. A
    mgr = FakeIndexManager()
    r = HybridRetriever(index_manager=mgr)
    out = asyncio.run(r.retrieve(query="What is RAG?", filters={"doc_id": "x"}, use_domain_index=True, domain="tech"))
    assert {o["id"] for o in out} == {"S", "P", "D"}
    assert all(o["metadata"]["retrieval_profile"] == "faq" for o in out)
    assert ("semantic_index", 20, 'doc_id == "x"') in mgr.calls and ("domain_index", 10, 'doc_id == "x"') in mgr.calls
    mgr.calls.clear()
    asyncio.run(r.retrieve(query="What is RAG?", profile_hint="default"))
    assert mgr.calls[0][1] == 40  # 2 * top_k over-retrieval
    r2 = HybridRetriever(index_manager=FakeIndexManager(with_meta=False))
    out = asyncio.run(r2.retrieve(query="What is RAG?"))
    assert out and "retrieval_profile" in out[0]


def test_retrieve_timeout_and_weight_adapter():  # reference test_extended.py:334-388
    r = HybridRetriever(index_manager=FakeIndexManager(delay=0.02))
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 0.005
    try:
        assert asyncio.run(r.retrieve(query="slow query")) == []
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old
    r = HybridRetriever(index_manager=FakeIndexManager(), weight_adapter=lambda q: (1.5, -0.2))
    assert (r.config.dense_weight, r.config.sparse_weight) == (0.7, 0.3)
    asyncio.run(r.retrieve(query="q"))
    assert (r.config.dense_weight, r.config.sparse_weight) == (1.0, 0.0)


def test_sparse_skipped_when_collection_missing_and_search_errors_degrade():
    class M(FakeIndexManager):
        collections = {"semantic_index": 1}
    mgr = M()
    asyncio.run(HybridRetriever(index_manager=mgr).retrieve("q"))
    assert [c[0] for c in mgr.calls] == ["semantic_index"]

    class Broken(FakeIndexManager):
        async def search(self, *a, **k):
            raise RuntimeError("boom")
    assert asyncio.run(HybridRetriever(index_manager=Broken()).retrieve("q")) == []


def test_retrieve_end_to_end_matches_reference_on_config1():
    """BASELINE config 1 (1k x 384, dense-only and hybrid): this package's HybridRetriever over a numpy
    FLAT manager reproduces the reference's ids, fused scores and method tags (golden g5)."""
    g = gold("g5_retrieve_c1.json")
    N, D = g["N"], g["D"]
    X = np.random.default_rng(g["corpus_seed"]).standard_normal((N, D)).astype(np.float32)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    srng = np.random.default_rng(g["sparse_seed"])
    rows = []
    for _ in range(N):
        idx = np.arange(100) * 100 + srng.integers(0, 100, size=100)
        rows.append((idx.astype(np.int32), np.abs(srng.standard_normal(100)).astype(np.float32)))
    qrng = np.random.default_rng(g["query_seed"])
    Q = qrng.standard_normal((g["n_queries"], D)).astype(np.float32)
    SQ = [((np.arange(100) * 100 + qrng.integers(0, 100, size=100)).astype(np.int32),
           np.abs(qrng.standard_normal(100)).astype(np.float32)) for _ in range(g["n_queries"])]

    class NumpyManager:
        def __init__(self, with_sparse):
            self.collections = {"semantic_index": 1, **({"sparse_index": 1} if with_sparse else {})}
            self.q = None

        async def _generate_semantic_embedding(self, text):
            return self.q[0]

        async def _generate_sparse_embedding(self, text):
            return {"indices": self.q[1][0].tolist(), "values": self.q[1][1].tolist()}

        async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
            if collection_name == "semantic_index":
                s = Xn @ (query_embedding / np.linalg.norm(query_embedding))
            else:
                idx, val = np.asarray(query_embedding["indices"]), np.asarray(query_embedding["values"], np.float32)
                keep = np.sort(np.argsort(np.abs(val), kind="stable")[int(np.floor(0.2 * len(val))):])
                qd = np.zeros(10000)
                qd[idx[keep]] = val[keep]
                s = np.array([float(np.sum(qd[ri] * rv.astype(np.float64))) for ri, rv in rows], dtype=np.float32)
            order = [int(i) for i in np.lexsort((np.arange(len(s)), -s))[:top_k] if collection_name == "semantic_index" or s[i] > 0]
            return [{"id": f"doc{r // 10}::{r % 10}::{r:08x}", "content": f"row {r}", "score": float(s[r]),
                     "metadata": {"doc_id": f"doc{r // 10}", "chunk_index": r % 10}} for r in order]

    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    try:
        for run in g["runs"]:
            mgr = NumpyManager(run["with_sparse"])
            mgr.q = (Q[run["query"]], SQ[run["query"]])
            out = asyncio.run(HybridRetriever(mgr, RetrievalConfig(top_k=20)).retrieve("plain statement", profile_hint="default"))
            assert [o["id"] for o in out] == run["ids"]
            assert [float(o["score"]).hex() for o in out] == run["scores"]
            assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
            assert out[0]["metadata"]["retrieval_profile"] == run["profile"]
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old


def test_embedding_cache_trace_matches_reference():
    g = gold("g7_cache.json")
    cache = EmbeddingCache(max_size=g["max_size"], ttl_seconds=3600)
    for op, key, want in g["trace"]:
        if op == "put":
            cache._sync_put(key, np.full(2, ord(key), dtype=np.float32))
        else:
            v = cache._sync_get(key)
            assert (None if v is None else float(v[0])) == want, (op, key)
    st = cache.get_stats()
    for k, want in g["stats"].items():
        assert st[k] == want, k
    assert cache._materialize_key("a") == g["key_a"] and cache._materialize_key("text", "model") == g["key_text_model"]


def test_embedding_cache_api_sync_async_ttl_and_singletons():  # reference tests/test_embedding_cache.py
    c = EmbeddingCache(maxsize=2, ttl_seconds=1)
    assert c.max_size == c.maxsize == 2
    asyncio.run(_cache_roundtrip(c))
    c2 = EmbeddingCache(max_size=4, ttl_seconds=1)
    c2._sync_put("k", np.ones(1))
    c2._cache[c2._materialize_key("k")] = (0.0, np.ones(1))  # force expiry
    assert c2._sync_get("k") is None and c2.get_stats()["size"] == 0
    off = EmbeddingCache(enabled=False)
    off._sync_put("k", 1)
    assert off._sync_get("k") is None
    assert get_semantic_cache() is get_semantic_cache() and get_sparse_cache().max_size == 10000
    assert get_domain_cache().max_size == 5000
    c.clear()
    assert c.get_stats()["hits"] == 0 and c.get_stats()["size"] == 0


async def _cache_roundtrip(c):
    assert await c.get("x") is None
    await c.put("x", np.arange(3))
    assert (await c.get("x")).tolist() == [0, 1, 2]
    await c.put("text", "model", np.ones(2))
    assert (await c.get("text", "model")).tolist() == [1, 1]
    calls = []

    async def compute():
        calls.append(1)
        return np.full(2, 7.0)

    assert (await c.get_or_compute("y", compute))[0] == 7.0
    assert (await c.get_or_compute("y", compute))[0] == 7.0 and len(calls) == 1
    assert c.get_stats()["evictions"] >= 1


def test_chunk_ids_and_query_rewriting_match_reference():
    for c in gold("g8_chunk_ids.json"):
        assert chunk_id_for(c["doc_id"], c["index"], c["content"]) == c["chunk_id"]
    qr = QueryRewriter()
    for q, want in gold("g10_rewrite.json"):
        assert qr.rewrite(q, {}) == want


def test_learned_ranker_monotone_and_feedback():  # reference test_extended.py:533-548, :694-713
    lr = LearnedRanker()
    res = [{"id": "a", "score": 0.2, "retrieval_methods": ["semantic", "sparse"]}, {"id": "b", "score": 0.2, "retrieval_methods": ["semantic"]},
           {"id": "c", "score": 0.1}]
    s = asyncio.run(lr.score("q", res))
    assert s[0] > s[1] > s[2] and s[0] == 0.2 + 0.1 * 2
    lr.update_from_feedback("q", res, [{"id": "a", "label": 1.0}, {"id": "zzz", "label": 0.0}])
    assert len(lr.training_examples) == 1 and lr.training_examples[0].label == 1.0


def test_manager_and_pipeline_without_connect():  # reference test_extended.py:391-426, :798-805, :849-859, :957-961
    m = MilvusIndexManager(connect=False)
    assert m.collections == {} and (m.semantic_dim, m.sparse_dim, m.domain_dim, m.num_shards) == (1536, 10000, 768, 4)
    sem = asyncio.run(m._generate_semantic_embedding("hello"))
    assert sem.shape == (1536,) and sem.dtype == np.float32
    sp = asyncio.run(m._generate_sparse_embedding("hello"))
    assert len(sp["indices"]) == 100 and sp["indices"] == sorted(sp["indices"]) and min(sp["values"]) >= 0
    assert asyncio.run(m._generate_domain_embedding("x", "d")).shape == (768,)
    with pytest.raises(ValueError):
        asyncio.run(m.search(np.zeros(4), "semantic_index"))
    assert m.get_collection_stats("semantic_index") == {}

    class Gen:
        async def encode_semantic(self, t):
            return np.full(4, 1.0, np.float32)

        def encode_sparse(self, t):
            return np.arange(4, dtype=np.float32)

        async def encode_domain(self, t, d=""):
            return np.full(4, 2.0, np.float32)

    m2 = MilvusIndexManager(semantic_dim=4, sparse_dim=4, domain_dim=4, connect=False)
    m2.embedding_generator = Gen()
    assert np.all(asyncio.run(m2._generate_semantic_embedding("unique-text-1")) == 1.0)
    assert np.all(asyncio.run(m2._generate_sparse_embedding("x")) == np.arange(4))
    assert np.all(asyncio.run(m2._generate_domain_embedding("x", domain="d")) == 2.0)
    asyncio.run(m.close())
    asyncio.run(m2.close())

    p = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(enable_audit_logging=False))
    assert p.config.top_k == 20 and p.config.rerank_top_k == 5 and p.retriever.config.rerank_top_k == 5
    assert p.retriever.config.dense_weight == 0.7 and p.retriever.learned_ranker is not None
    for ms in (10.0, 20.0, 100.0):
        p._record_latency(PipelineStage.RETRIEVAL, ms)
    rep = p.get_performance_report()
    assert rep["stage_latencies"]["retrieval"]["p50"] == 20.0 and 0 < rep["sla_compliance"]["compliance_rate"] < 1
    asyncio.run(p.close())


def test_pipeline_retrieve_shape_with_fake_manager():
    p = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(enable_audit_logging=True))
    p.index_manager = FakeIndexManager()
    p.retriever.index_manager = p.index_manager
    p.retriever.config.enable_learned_ranker = True
    results, metrics = asyncio.run(p.retrieve("plain statement", context={"retrieval_profile": "default"}))
    assert results and results[0].chunk_id in {"S", "P"} and results[0].audit_trail is not None
    assert 0.0 <= metrics.hallucination_risk <= 1.0 and set(metrics.to_dict()) >= {"retrieval_precision", "ndcg_at_k"}
    plan = asyncio.run(p.plan_and_execute("alpha and beta"))
    assert plan["decomposition"]["sub_queries"] == ["alpha", "beta"] and len(plan["subqueries"]) == 2


def test_bm25_encoder_scores_equal_textbook_bm25():
    docs = ["the quick brown fox jumps", "the lazy dog sleeps all day long", "quick quick fox", "unrelated text entirely"]
    enc = BM25SparseEncoder(sparse_dim=4096).fit(docs)
    q = enc.encode_query("quick fox")
    scores = []
    for d in docs:
        dv = enc.encode_document(d)
        dd = dict(zip(dv["indices"], dv["values"]))
        scores.append(sum(v * dd.get(i, 0.0) for i, v in zip(q["indices"], q["values"])))
    assert scores[2] > scores[0] > 0 and scores[1] == 0 and scores[3] == 0
    import math
    toks = "quick quick fox".split()
    idf = lambda df: math.log(1 + (4 - df + 0.5) / (df + 0.5))
    avgdl = sum(len(d.split()) for d in docs) / 4
    want = sum(idf(2) * tf * 2.2 / (tf + 1.2 * (1 - 0.75 + 0.75 * 3 / avgdl)) for tf in (2, 1))
    assert abs(scores[2] - want) < 1e-5
    assert q["indices"] == sorted(q["indices"])


def test_bench_names_the_dense_scan_kernel_per_batch():
    """bench.py's roofline.kernel label follows the dispatch in csrc/hbmrag.hip::dense_search_enqueue."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert "dense_scan_kernel" in mod.scan_kernel_name(1, 768) and "dense_scan_kernel" in mod.scan_kernel_name(64, 768)
    assert "bigq" in mod.scan_kernel_name(65, 768) and "bigq" in mod.scan_kernel_name(128, 768)
    assert "qreg" in mod.scan_kernel_name(129, 768) and "qreg" in mod.scan_kernel_name(256, 700)   # 700 pads to 768
    assert "gemm" in mod.scan_kernel_name(256, 1024) and "gemm" in mod.scan_kernel_name(256, 384)  # KT != 24, >= 8
    assert "bigq" in mod.scan_kernel_name(256, 128) and "bigq" in mod.scan_kernel_name(128, 1024)


def test_bench_step_hbm_sums_the_committed_pmc_traffic_of_every_search_kernel():
    """bench.py's step_hbm block: bytes of all search kernels per step (profiles/r*_pmc_traffic*.json) over the step time;
    only for the workloads that were profiled."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the NEWEST collection of the workload is the one consulted (an older round's file lists kernels that no longer run)
    import glob
    newest = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic.json")))[-1]     # the default workload's collections
    k = json.load(open(newest))["kernels"]
    assert "finish_fused" in k and "refine_dense" not in k
    want = sum(k[name]["hbm_bytes_per_launch"] for name in mod.STEP_KERNELS if name in k)
    got = mod.step_hbm(10_000_000, 768, 128, 1, "uniform", True, 4.0)
    assert got is not None and abs(got["bytes_per_step"] - want) < 1.0
    assert abs(got["achieved"] - want / 4.0e-3 / 1e9) < 1e-6 and got["traffic_source"].startswith("profiles/")
    assert got["bytes_per_step"] > k["dense_scan"]["algorithmic_bytes_per_launch"]
    assert mod.step_hbm(1_000_000, 768, 128, 1, "uniform", True, 1.0) is None     # not a profiled workload
    c5 = mod.step_hbm(50_000_000, 1024, 256, 1, "uniform", False, 26.0)            # dense only: no sparse kernels, one select each
    assert c5 is not None and c5["bytes_per_step"] < 1.2e11


def test_pmc_traffic_tool_doubles_fetch_and_filters_small_launches(tmp_path):
    import importlib.util
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = "Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value\n"
    f = tmp_path / "f.csv"
    w = tmp_path / "w.csv"
    f.write_text(hdr + "1,hbmrag::dense_scan_bigq_kernel<x>,131072,FETCH_SIZE,1000.0\n"
                       "2,hbmrag::dense_scan_bigq_kernel<x>,131072,FETCH_SIZE,3000.0\n"
                       "3,hbmrag::dense_scan_kernel<x>,512,FETCH_SIZE,10.0\n"
                       "4,hbmrag::sparse_scan_kernel(a),1024,FETCH_SIZE,500.0\n")
    w.write_text(hdr + "1,hbmrag::dense_scan_bigq_kernel<x>,131072,WRITE_SIZE,40.0\n"
                       "4,hbmrag::sparse_scan_kernel(a),1024,WRITE_SIZE,8.0\n")
    out = subprocess.run([sys.executable, os.path.join(root, "profiles", "make_pmc_traffic.py"), str(f), str(w),
                          "1000", "768", "128"], capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    dense = d["kernels"]["dense_scan"]
    assert dense["launches_averaged"] == 1                       # 1000 KiB and 10 KiB launches are below half the largest
    assert dense["fetch_bytes_per_launch"] == 2 * 3000.0 * 1024  # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    assert dense["write_bytes_per_launch"] == 40.0 * 1024
    assert dense["algorithmic_bytes_per_launch"] == 1000 * 768 * 2 + 4 * 1000
    assert d["kernels"]["sparse_scan"]["hbm_bytes_per_launch"] == (2 * 500.0 + 8.0) * 1024


def test_search_hit_formatting_matches_reference_g6():
    """Golden G6 (tests/golden/gen_golden_g6.py): the dicts the REFERENCE's MilvusIndexManager.search builds from
    Milvus hits (reference indexing.py:533-551) — key order, id := chunk_id, the six metadata keys, float32-rounded
    FLOAT fields — against this package's _format_hits fed the same rows and scores; plus the two ValueErrors of
    indexing.py:466-467, :497-498 (same messages) and the default search params per collection (:472-484)."""
    from advanced_rag.indexing import ShardCollection
    g = gold("g6_search_format.json")
    rows = g["rows"]
    mgr = MilvusIndexManager(semantic_dim=g["dim"], sparse_dim=g["sparse_dim"], connect=False)
    c = mgr._cols
    for r in rows:
        c["id"].append(r["chunk_id"])
        for k in ("doc_id", "content", "chunk_index", "token_count", "timestamp", "metadata_json"):
            c[k].append(r[k])
        for k in ("entropy", "redundancy", "domain_density"):
            c[k].append(float(np.float32(r[k])))
    row_of = {r["chunk_id"]: i for i, r in enumerate(rows)}
    for case in g["cases"]:
        want = case["results"]
        ids = np.array([row_of[w["id"]] for w in want] + [-1], dtype=np.int64)   # -1 padding ends a list
        scores = np.array([w["score"] for w in want] + [0.0], dtype=np.float32)
        got = mgr._format_hits(ids, scores)
        assert len(got) == len(want)
        for o, w in zip(got, want):
            assert o.pop("_row") == row_of[w["id"]]
            assert list(o) == list(w) == ["id", "content", "score", "metadata"]
            assert list(o["metadata"]) == list(w["metadata"])
            assert o == w, case["label"]  # scores round-trip through float32 exactly: they came from a float32
        call = case["collection_search_call"]
        if case["search_params"] is None:  # defaults chosen by collection kind
            assert call["param"] == ({"metric_type": "IP"} if case["collection"] == "sparse_index"
                                     else {"metric_type": "COSINE", "params": {"ef": 64}})
    errs = {e["label"]: e for e in g["errors"]}
    with pytest.raises(ValueError) as ei:
        asyncio.run(mgr.search(np.zeros(4, np.float32), "nope"))
    assert str(ei.value) == errs["unknown-collection"]["message"]
    mgr.collections["sparse_index"] = ShardCollection(mgr, "sparse_index", "sparse", None, g["sparse_dim"], "IP")
    with pytest.raises(ValueError) as ei:
        asyncio.run(mgr.search(np.zeros(4, np.float32), "sparse_index"))
    assert str(ei.value) == errs["sparse-bad-payload"]["message"]
    mgr.collections.clear()
    asyncio.run(mgr.close())


def test_synthetic_payload_shards_filter_on_what_the_row_number_encodes():
    """Bulk-ingested (payload-free) shards derive chunk_index = row % 10 for filter expressions; other fields need
    payload columns and say so.  The mask is cached per (expression, rows, tombstone epoch)."""
    from advanced_rag.indexing import MilvusIndexManager
    m = MilvusIndexManager.__new__(MilvusIndexManager)
    m._synthetic_rows, m._mask_cache, m._delete_epoch, m._deleted = 25, {}, 0, None
    m._cols = {"id": []}
    keep = m._row_mask("chunk_index < 3")
    assert keep.dtype == bool and keep.tolist() == [(r % 10) < 3 for r in range(25)]
    assert m._row_mask("chunk_index < 3") is keep                      # cached
    assert m._row_mask(None) is None
    with pytest.raises(ValueError, match="chunk_index"):
        m._row_mask('doc_id == "doc1"')


def test_payload_columns_are_append_only_arrays_not_python_objects():
    """columns.PayloadColumns (the manager's host payload store): ids / doc ids as offset-encoded UTF-8, numeric fields
    as numpy arrays; two million rows cost well under 400 MB (10M rows < 2 GB) and an append costs O(batch); the
    16-byte prefix keys order rows like their strings wherever the prefixes differ."""
    import time
    from advanced_rag.columns import PayloadColumns, StringColumn
    c = PayloadColumns()
    n, step = 2_000_000, 250_000
    costs = []
    for lo in range(0, n, step):
        t0 = time.perf_counter()
        c["id"].extend(f"doc{r // 10}::{r % 10}::{r:08x}" for r in range(lo, lo + step))
        c["doc_id"].extend(f"doc{r // 10}" for r in range(lo, lo + step))
        c["content"].extend("" for _ in range(step))
        c["timestamp"].extend("" for _ in range(step))
        c["metadata_json"].extend("" for _ in range(step))
        c["chunk_index"].extend(np.arange(lo, lo + step) % 10)
        c["token_count"].extend(np.zeros(step, np.int64))
        for k in ("entropy", "redundancy", "domain_density"):
            c[k].extend(np.zeros(step, np.float32))
        costs.append(time.perf_counter() - t0)
    assert c.n_rows == n and c.nbytes < 400e6, c.nbytes
    assert costs[-1] < 3 * costs[0] + 0.05                  # no rebuild of what is already there
    assert c["id"][1234567] == "doc123456::7::0012d687" and c["doc_id"][-1] == f"doc{(n - 1) // 10}"
    assert c["chunk_id"] is c["id"] and c["chunk_index"][13] == 3 and isinstance(c["entropy"][0], float)
    s = StringColumn()
    vals = ["b", "a", "", "aé", "0123456789abcdefX", "0123456789abcdefA", "zz", "a" * 40, "a\x00"]
    s.extend(vals)
    k = s.keys()
    for i in range(len(vals)):
        for j in range(len(vals)):
            ki, kj = (int(k[i, 0]), int(k[i, 1])), (int(k[j, 0]), int(k[j, 1]))
            bi, bj = vals[i].encode(), vals[j].encode()
            if ki != kj:
                assert (ki < kj) == (bi < bj), (vals[i], vals[j])
            else:
                assert bi[:16].ljust(16, b"\x00") == bj[:16].ljust(16, b"\x00")
    assert s.compare_rows(np.array([4, 5]), "<", "0123456789abcdefB").tolist() == [False, True]
    s.append("tail")
    assert s.keys().shape == (len(vals) + 1, 2) and StringColumn.key_of("tail") == (int(s.keys()[-1, 0]), int(s.keys()[-1, 1]))


# --------------------------------------------------------------------------- G11 / G12 (round 4): MMR, profiles, pipeline
def test_fuse_results_with_mmr_matches_reference_g11():
    """`_fuse_results` with enable_mmr=True (reference retrieval.py:488-516) at lambda 0.5 / 0.7 / 0.8 / 0 / 1: this
    package's host fusion + `_mmr_diversify` return the imported reference's order, float64 scores and method tags."""
    cases = gold("g11_mmr.json")["fuse"]
    for c in cases:
        hit = lambda i, r: {"id": i, "content": c["content"][i], "score": 1.0 - 0.01 * r}   # noqa: E731
        r = HybridRetriever(index_manager=None, config=RetrievalConfig(dense_weight=c["dense_weight"], sparse_weight=c["sparse_weight"],
                                                                       top_k=c["top_k"], enable_mmr=True, mmr_lambda=c["mmr_lambda"]))
        out = r._fuse_results([hit(i, n) for n, i in enumerate(c["semantic"])], [hit(i, n) for n, i in enumerate(c["sparse"])],
                              [hit(i, n) for n, i in enumerate(c["domain"])])
        assert [o["id"] for o in out] == c["ids"], c["label"]
        assert [float(o["score"]).hex() for o in out] == c["scores"], c["label"]
        assert [sorted(o["retrieval_methods"]) for o in out] == c["methods"], c["label"]


def test_retrieve_under_mmr_profiles_matches_reference_g11():
    import g5_data
    g, X, csr, Q, SQ = g5_data.inputs()
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    try:
        for run in gold("g11_mmr.json")["retrieve"]:
            mgr = g5_data.NumpyFlatManager(X, csr, Q, SQ, run["with_sparse"], fixed_query=run["query"])
            retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
            out = asyncio.run(retr.retrieve("plain statement", profile_hint=run["profile_hint"]))
            assert sorted(set(mgr.seen_top_k)) == run["search_top_k"]
            assert [o["id"] for o in out] == run["ids"], (run["profile_hint"], run["query"])
            assert [float(o["score"]).hex() for o in out] == run["scores"]
            assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
            assert out[0]["metadata"]["retrieval_profile"] == run["profile"]
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old


def test_pipeline_retrieve_matches_reference_g12():
    """AdvancedRAGPipeline.retrieve() on its deterministic branch (learned ranker) against what the REFERENCE's pipeline
    returned over fake collections with the same rows (reference pipeline.py:217-309): order, scores, retrieval_method,
    metadata, the number of results for PipelineConfig.rerank_top_k = 5 / 7 / 12, reranking disabled, top_k = 10."""
    import g5_data
    g, X, csr, Q, SQ = g5_data.inputs()
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    try:
        for c in gold("g12_pipeline.json")["cases"]:
            p = AdvancedRAGPipeline(connect_to_milvus=False,
                                    config=PipelineConfig(enable_audit_logging=False, rerank_top_k=c["pipeline_rerank_top_k"],
                                                          enable_reranking=c["enable_reranking"], top_k=c["top_k"]))
            p.index_manager = g5_data.NumpyFlatManager(X, csr, Q, SQ, c["with_sparse"])
            p.retriever.index_manager = p.index_manager
            p.retriever.config.enable_learned_ranker = True
            assert p.retriever.config.rerank_top_k == c["retriever_rerank_top_k"]
            results, metrics = asyncio.run(p.retrieve(c["query"], context=c["context"]))
            assert sorted(set(p.index_manager.seen_top_k)) == c["search_limits"]
            assert len(results) == c["n"], c["label"]
            assert [r.chunk_id for r in results] == c["chunk_ids"], c["label"]
            assert [float(r.score).hex() for r in results] == c["scores"], c["label"]
            assert [r.retrieval_method for r in results] == c["retrieval_methods"]
            assert [r.content for r in results] == c["contents"]
            assert [r.metadata["doc_id"] for r in results] == c["doc_ids"]
            assert [r.metadata.get("retrieval_profile") for r in results] == c["profiles"]
            assert sorted(k for k in results[0].metadata if k != "recency") == c["metadata_keys"]
            assert sorted(results[0].__dataclass_fields__) == c["result_fields"]
            assert type(metrics).__name__ == c["metrics_type"]
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old
