"""Generate golden vectors from the REFERENCE implementation (container-only).

Run here, never on the GPU box:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
It imports the reference's logic modules the way the reference's own tests do
(tests/test_embedding_cache.py:22-26 there): an empty parent package whose
__path__ points at /root/reference/src/advanced_rag, so __init__.py — and with
it pymilvus — is never executed.  No third-party stand-ins are used; modules
that need pymilvus (indexing.py, pipeline.py) are not imported.

Outputs (small JSON, committed):  tests/golden/*.json — inputs and expected
outputs only.  float64 values are stored as float.hex() strings so they compare
bit for bit.
"""
import asyncio
import importlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/advanced_rag"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def ref_module(name):
    if "advanced_rag" not in sys.modules or getattr(sys.modules["advanced_rag"], "__path__", None) != [REF]:
        pkg = types.ModuleType("advanced_rag")
        pkg.__path__ = [REF]
        sys.modules["advanced_rag"] = pkg
    return importlib.import_module(f"advanced_rag.{name}")


def hexf(x):
    return float(x).hex()


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=None if name.startswith("g1_") else 1)  # insertion order matters (filter dicts)
    print("wrote", name)


def hits(ids, prefix="c"):
    return [{"id": i, "content": f"{prefix} {i}", "score": 1.0 - 0.01 * r} for r, i in enumerate(ids)]


def main():
    R = ref_module("retrieval")
    K = ref_module("ranker")
    C = ref_module("constants")
    EC = ref_module("embedding_cache")
    CH = ref_module("chunking")
    QR = ref_module("query_rewriting")

    # ---- G1: _fuse_results -------------------------------------------------------------
    rng = np.random.default_rng(20240601)
    cases = []

    def fuse_case(sem, sp, dom, dw, sw, label):
        r = R.HybridRetriever(index_manager=None, config=R.RetrievalConfig(dense_weight=dw, sparse_weight=sw))
        out = r._fuse_results(hits(sem, "s"), hits(sp, "p"), hits(dom, "d"))
        cases.append({"label": label, "semantic": sem, "sparse": sp, "domain": dom, "dense_weight": dw,
                      "sparse_weight": sw, "ids": [o["id"] for o in out], "scores": [hexf(o["score"]) for o in out],
                      "methods": [sorted(o["retrieval_methods"]) for o in out],
                      "payload_from": [o["content"].split()[0] for o in out]})

    fuse_case(["d0", "d1", "d2", "d3", "d4"], ["d3", "d7", "d0"], [], 0.7, 0.3, "survey-appendix-a")
    fuse_case(["S1", "S2"], ["P1", "P2"], [], 0.5, 0.5, "tie-keeps-insertion-order")
    fuse_case([], [], [], 0.7, 0.3, "all-empty")
    fuse_case(["a"], [], [], 0.7, 0.3, "one-semantic")
    fuse_case([], ["a", "b"], [], 0.7, 0.3, "sparse-only")
    fuse_case(["a", "b", "c"], ["c", "b", "a"], ["b", "z", "a"], 0.7, 0.3, "with-domain")
    for n_sem, n_sp, n_dom, ov in ((40, 40, 0, 10), (40, 40, 20, 15), (80, 80, 0, 33), (200, 200, 100, 50),
                                   (40, 7, 0, 7), (3, 40, 5, 2)):
        pool = [f"r{int(x)}" for x in rng.permutation(5000)[:n_sem + n_sp + n_dom]]
        sem = pool[:n_sem]
        sp = list(rng.permutation(sem[:ov] + pool[n_sem:n_sem + n_sp - ov]))
        dom = list(rng.permutation((sem[::5] + sp[::7] + pool[n_sem + n_sp:])[:n_dom])) if n_dom else []
        dom = list(dict.fromkeys(dom))
        sp = [str(x) for x in sp]
        dom = [str(x) for x in dom]
        for dw, sw in ((0.7, 0.3), (0.5, 0.5), (0.1, 0.9)):
            fuse_case(sem, sp, dom, dw, sw, f"random-{n_sem}-{n_sp}-{n_dom}-{dw}")
    dump("g1_fuse.json", cases)

    # ---- G2: rerank ------------------------------------------------------------------------
    g2 = []
    r = R.HybridRetriever(index_manager=None, config=R.RetrievalConfig(enable_learned_ranker=True),
                          learned_ranker=K.LearnedRanker())
    fused = r._fuse_results(hits(["d0", "d1", "d2", "d3"], "s"), hits(["d2", "d9"], "p"), [])
    out = asyncio.run(r.rerank("q", fused, top_k=3))
    g2.append({"label": "learned-ranker", "semantic": ["d0", "d1", "d2", "d3"], "sparse": ["d2", "d9"], "top_k": 3,
               "ids": [o["id"] for o in out], "scores": [hexf(o["score"]) for o in out],
               "original": [hexf(o["original_retrieval_score"]) for o in out]})

    class Inject:
        def __init__(self, s):
            self.s = s

        async def score(self, pairs):
            return list(self.s)

    r = R.HybridRetriever(index_manager=None)
    r.reranker = Inject([0.1, 0.9, 0.9, 0.3])
    res = [{"id": x, "content": x, "score": 0.5 - 0.1 * i} for i, x in enumerate("ABCD")]
    out = asyncio.run(r.rerank("q", res, top_k=None))
    g2.append({"label": "injected-stable", "inject": [0.1, 0.9, 0.9, 0.3], "ids": [o["id"] for o in out],
               "n": len(out), "default_top_k": r.config.rerank_top_k})
    r = R.HybridRetriever(index_manager=None, config=R.RetrievalConfig(enable_reranking=False))
    res = [{"id": x, "content": x, "score": 0.5} for x in "ABC"]
    g2.append({"label": "disabled", "top2": [o["id"] for o in asyncio.run(r.rerank("q", list(res), top_k=2))],
               "none": [o["id"] for o in asyncio.run(r.rerank("q", list(res)))]})
    dump("g2_rerank.json", g2)

    # ---- G3: classifier + profiles ------------------------------------------------------------
    queries = ["", "   ", "What is RAG?", "plain statement query", "there is an error here", "give me a summary",
               "Please provide a summary or overview of RAG.", "I see an error: connection failed", "x" * 250,
               "x" * 199, "short?", "y" * 80 + "?", "y" * 79 + "?", "tl;dr please", "Stack Trace attached", "bug?",
               "Is this a failure?", "overview?"]
    clf = R.QueryClassifier()
    g3 = {"classify": [[q, clf.classify(q)] for q in queries], "profiles": {}}
    for k, rk in ((5, 5), (20, 5), (50, 20), (100, 5), (150, 120), (1, 1)):
        hr = R.HybridRetriever(index_manager=None, config=R.RetrievalConfig(top_k=k, rerank_top_k=rk))
        g3["profiles"][f"{k},{rk}"] = {name: {"top_k": p.top_k, "rerank_top_k": p.rerank_top_k,
                                               "enable_mmr": p.enable_mmr, "mmr_lambda": p.mmr_lambda,
                                               "enable_reranking": p.enable_reranking,
                                               "dense_weight": p.dense_weight, "sparse_weight": p.sparse_weight}
                                        for name, p in hr.profiles.items()}
    g3["max_top_k"] = C.RetrievalConstants.MAX_TOP_K
    g3["timeout_seconds"] = C.RetrievalConstants.TIMEOUT_SECONDS
    cfg = R.RetrievalConfig()
    g3["default_config"] = {"hybrid_alpha": cfg.hybrid_alpha, "top_k": cfg.top_k, "rerank_top_k": cfg.rerank_top_k,
                            "enable_reranking": cfg.enable_reranking, "dense_weight": cfg.dense_weight,
                            "sparse_weight": cfg.sparse_weight, "enable_mmr": cfg.enable_mmr,
                            "mmr_lambda": cfg.mmr_lambda, "enable_learned_ranker": cfg.enable_learned_ranker,
                            "semantic_search_params": cfg.semantic_search_params,
                            "sparse_search_params": cfg.sparse_search_params}
    dump("g3_profiles.json", g3)

    # ---- G4: filter expressions -----------------------------------------------------------------
    hr = R.HybridRetriever(index_manager=None)
    g4 = []
    for f in ({"doc_id": 'doc"123', "entropy": {"$gte": 0.2}},
              {"redundancy": {"$lt": 0.5, "$gt": 0.1, "$eq": 0.2, "$ne": 0.3}, "chunk_index": 1},
              {}, {"doc_id": "a\\b"}, {"timestamp": {"$gte": "2024-01-01", "$lt": "2025-01-01"}},
              {"token_count": {"$lte": 512}, "domain_density": 0.5, "chunk_id": "d::0::abcd1234"},
              {"chunk_index": True}, {"entropy": {"$gte": 1}},
              {"domain": "x"}, {"source": "y"}, {"doc_id": {"$in": ["a"]}}, {"doc_id": {"$eq": [1]}},
              {"doc_id": None}, {"entropy": [1, 2]}):
        try:
            g4.append({"filters": f, "expr": hr._build_filter_expression(f)})
        except Exception as e:
            g4.append({"filters": f, "error": type(e).__name__})
    dump("g4_filters.json", g4)

    # ---- G5: HybridRetriever.retrieve end to end on BASELINE config 1 ------------------------------
    class NumpyManager:
        """Exact FLAT stand-in for the Milvus collections (the 'Milvus replaced by
        numpy' path of BASELINE.md §3), fp32 arithmetic."""

        def __init__(self, X, sp_rows, with_sparse):
            self.X = X
            self.Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
            self.sp = sp_rows
            self.collections = {"semantic_index": 1}
            if with_sparse:
                self.collections["sparse_index"] = 1
            self.q = None

        async def _generate_semantic_embedding(self, text):
            return self.q[0]

        async def _generate_sparse_embedding(self, text):
            return {"indices": self.q[1][0].tolist(), "values": self.q[1][1].tolist()}

        async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
            if collection_name == "semantic_index":
                qn = query_embedding / np.linalg.norm(query_embedding)
                s = self.Xn @ qn
            else:
                idx = np.asarray(query_embedding["indices"])
                val = np.asarray(query_embedding["values"], dtype=np.float32)
                order = np.argsort(np.abs(val), kind="stable")
                keep = np.sort(order[int(np.floor(0.2 * len(val))):])
                qd = np.zeros(10000, dtype=np.float64)
                qd[idx[keep]] = val[keep]
                s = np.array([float(np.sum(qd[ri] * rv.astype(np.float64))) for ri, rv in self.sp], dtype=np.float32)
            order = np.lexsort((np.arange(len(s)), -s))[:top_k]
            order = [int(i) for i in order if collection_name == "semantic_index" or s[i] > 0]
            return [{"id": f"doc{r // 10}::{r % 10}::{r:08x}", "content": f"row {r}", "score": float(s[r]),
                     "metadata": {"doc_id": f"doc{r // 10}", "chunk_index": r % 10}} for r in order]

    N, D = 1000, 384
    X = np.random.default_rng(1234).standard_normal((N, D)).astype(np.float32)
    srng = np.random.default_rng(5678)
    sp_rows = []
    for _ in range(N):
        idx = np.arange(100) * 100 + srng.integers(0, 100, size=100)
        sp_rows.append((idx.astype(np.int32), np.abs(srng.standard_normal(100)).astype(np.float32)))
    qrng = np.random.default_rng(4321)
    Q = qrng.standard_normal((8, D)).astype(np.float32)
    SQ = [((np.arange(100) * 100 + qrng.integers(0, 100, size=100)).astype(np.int32),
           np.abs(qrng.standard_normal(100)).astype(np.float32)) for _ in range(8)]
    C.RetrievalConstants.TIMEOUT_SECONDS = 60.0
    g5 = {"N": N, "D": D, "corpus_seed": 1234, "sparse_seed": 5678, "query_seed": 4321, "n_queries": 8, "runs": []}
    for with_sparse in (False, True):
        mgr = NumpyManager(X, sp_rows, with_sparse)
        hr = R.HybridRetriever(index_manager=mgr, config=R.RetrievalConfig(top_k=20))
        for qi in range(8):
            mgr.q = (Q[qi], SQ[qi])
            out = asyncio.run(hr.retrieve("plain statement", profile_hint="default"))
            g5["runs"].append({"with_sparse": with_sparse, "query": qi, "ids": [o["id"] for o in out],
                               "scores": [hexf(o["score"]) for o in out],
                               "methods": [sorted(o["retrieval_methods"]) for o in out],
                               "profile": out[0]["metadata"]["retrieval_profile"]})
    dump("g5_retrieve_c1.json", g5)

    # ---- G7: EmbeddingCache trace ---------------------------------------------------------------------
    cache = EC.EmbeddingCache(max_size=3, ttl_seconds=3600)
    trace = []
    for op, key in (("get", "a"), ("put", "a"), ("put", "b"), ("get", "a"), ("put", "c"), ("put", "d"), ("get", "a"),
                    ("get", "b"), ("put", "b"), ("get", "d"), ("put", "e"), ("get", "c"), ("get", "e")):
        if op == "put":
            cache._sync_put(key, np.full(2, ord(key), dtype=np.float32))
            trace.append([op, key, None])
        else:
            v = cache._sync_get(key)
            trace.append([op, key, None if v is None else float(v[0])])
    st = cache.get_stats()
    g7 = {"max_size": 3, "trace": trace, "stats": {k: st[k] for k in ("size", "hits", "misses", "evictions", "hit_rate")},
          "key_a": cache._materialize_key("a"), "key_text_model": cache._materialize_key("text", "model")}
    dump("g7_cache.json", g7)

    # ---- G8: chunk ids; G10: query rewriting -----------------------------------------------------------------
    ch = CH.AdaptiveChunker()
    g8 = [{"doc_id": d, "index": i, "content": c, "chunk_id": ch._generate_chunk_id(d, i, c)}
          for d, i, c in (("doc1", 0, "Alpha beta gamma."), ("doc-2", 7, "The quick brown fox."), (42, 3, "ünïcode ✓"))]
    dump("g8_chunk_ids.json", g8)
    qr = QR.QueryRewriter()
    g10 = [[q, qr.rewrite(q, {})] for q in ("What is RAG?", "rag and llm", "retrieval augmented generation with rag",
                                             "LLM basics", "", "  padded  ", "nothing special", "dragon")]
    dump("g10_rewrite.json", g10)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    main()
