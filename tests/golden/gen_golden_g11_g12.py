"""Generate golden fixtures G11 and G12 from the REFERENCE (container-only).

Run here, never on the GPU box:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_g11_g12.py

G11  the MMR branch of `_fuse_results` (reference src/advanced_rag/retrieval.py:488-516) and the retrieval profiles
     that switch it on (retrieval.py:142-213):
       a) `_fuse_results` with enable_mmr=True at lambda = 0.5 / 0.7 / 0.8 on seeded rank lists whose `content` strings
          share tokens (token-Jaccard similarity is what MMR diversifies on);
       b) `HybridRetriever.retrieve` on BASELINE config 1 (g5's corpus, sparse rows and queries, rebuilt from the seeds
          g5 stores) under profile_hint = troubleshooting / analysis / summary (top_k 30 / 30 / 40, searches with
          k' = 60 / 60 / 80, MMR on / on / off), the collection's `content` being g5_data.mmr_content(row).
G12  the reference's own `AdvancedRAGPipeline(connect_to_milvus=False).retrieve()` (pipeline.py:217-309) with
     `enable_learned_ranker=True` (retrieval.py:544-545; ranker.py:109-125) over fake Milvus collections holding the same
     corpus: chunk_id, score, retrieval_method, order, and the `rerank_top_k` quirk (PipelineConfig.rerank_top_k is what
     `rerank` is cut to; the retriever's own RetrievalConfig.rerank_top_k is never forwarded, pipeline.py:104-110).

Import method: logic-only for G11 (an empty parent package whose __path__ points at the reference sources, as the
reference's own tests do); for G12 additionally the inert `pymilvus` / `tenacity` stand-ins of SURVEY.md App. B.2 (module
objects carrying the imported names only; nothing of Milvus is implemented — the collections are exact numpy FLAT
fakes shaped like pymilvus' results, as in gen_golden_g6.py).

Outputs: tests/golden/g11_mmr.json, tests/golden/g12_pipeline.json — inputs (or the seeds they are rebuilt from) and
expected outputs only; float64 values as float.hex() strings.
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
sys.path.insert(0, os.path.dirname(HERE))
import g5_data  # noqa: E402  (the seeds -> inputs restatement the tests use too)


def install_parent_package():
    pkg = types.ModuleType("advanced_rag")
    pkg.__path__ = [REF]
    sys.modules["advanced_rag"] = pkg


def install_stand_ins():
    pm = types.ModuleType("pymilvus")
    for name in ("connections", "Collection", "CollectionSchema", "FieldSchema", "DataType", "utility"):
        setattr(pm, name, type(name, (), {}))
    sys.modules["pymilvus"] = pm
    tn = types.ModuleType("tenacity")
    tn.retry = lambda *a, **k: (lambda fn: fn)
    tn.stop_after_attempt = tn.wait_exponential = tn.retry_if_exception_type = lambda *a, **k: None
    sys.modules["tenacity"] = tn


def hexf(x):
    return float(x).hex()


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=None)
    print("wrote", name)


WORDS = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]


def g11(R, C):
    out = {"fuse": [], "retrieve": []}
    rng = np.random.default_rng(1111)

    # ---- a) _fuse_results with MMR ----------------------------------------------------------------------------
    def fuse_case(label, n_sem, n_sp, overlap, n_words, top_k, lam, dw=0.7, sw=0.3, n_dom=0):
        pool = [f"r{int(x)}" for x in rng.permutation(900)[: n_sem + n_sp + n_dom]]
        sem = pool[:n_sem]
        sp = [str(x) for x in rng.permutation(sem[:overlap] + pool[n_sem:n_sem + n_sp - overlap])]
        dom = [str(x) for x in rng.permutation(sem[::4] + pool[n_sem + n_sp:])][:n_dom]
        content = {}
        for i in sem + sp + dom:
            if i not in content:
                k = int(rng.integers(0, n_words + 1))
                content[i] = " ".join(str(w) for w in rng.choice(WORDS, size=k, replace=False)) if k else ""
        # a document with no content at all and one with content None, as the reference tolerates both (r.get("content") or "")
        hit = lambda i, r: {"id": i, "content": content[i], "score": 1.0 - 0.01 * r}   # noqa: E731
        r = R.HybridRetriever(index_manager=None,
                              config=R.RetrievalConfig(dense_weight=dw, sparse_weight=sw, top_k=top_k, enable_mmr=True, mmr_lambda=lam))
        res = r._fuse_results([hit(i, n) for n, i in enumerate(sem)], [hit(i, n) for n, i in enumerate(sp)],
                              [hit(i, n) for n, i in enumerate(dom)])
        out["fuse"].append({"label": label, "semantic": sem, "sparse": sp, "domain": dom, "content": content,
                            "dense_weight": dw, "sparse_weight": sw, "top_k": top_k, "mmr_lambda": lam,
                            "ids": [o["id"] for o in res], "scores": [hexf(o["score"]) for o in res],
                            "methods": [sorted(o["retrieval_methods"]) for o in res]})

    for lam in (0.5, 0.7, 0.8):
        fuse_case(f"mmr-40-40-lam{lam}", 40, 40, 12, 5, 20, lam)
        fuse_case(f"mmr-60-60-k30-lam{lam}", 60, 60, 20, 4, 30, lam)
        fuse_case(f"mmr-short-lists-lam{lam}", 7, 5, 2, 3, 30, lam)          # fewer fused rows than top_k
        fuse_case(f"mmr-domain-lam{lam}", 40, 40, 10, 6, 25, lam, 0.5, 0.5, 12)
    fuse_case("mmr-identical-contents", 12, 0, 0, 0, 5, 0.5)                   # every content "" -> similarity 0/1
    fuse_case("mmr-lambda-1", 30, 30, 8, 5, 10, 1.0)                          # pure relevance
    fuse_case("mmr-lambda-0", 30, 30, 8, 5, 10, 0.0)                          # pure diversity: first wins on ties

    # ---- b) retrieve() under the MMR profiles on config 1 ----------------------------------------------------------
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs()
    N = X.shape[0]
    sp_rows = [(idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]]) for r in range(N)]

    class NumpyManager:
        """The exact FLAT stand-in of gen_golden.py (g5), with contents that share tokens."""

        def __init__(self, with_sparse):
            self.Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
            self.collections = {"semantic_index": 1}
            if with_sparse:
                self.collections["sparse_index"] = 1
            self.q = None
            self.seen_top_k = []

        async def _generate_semantic_embedding(self, text):
            return self.q[0]

        async def _generate_sparse_embedding(self, text):
            return {"indices": self.q[1][0].tolist(), "values": self.q[1][1].tolist()}

        async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
            self.seen_top_k.append(top_k)
            if collection_name == "semantic_index":
                qn = query_embedding / np.linalg.norm(query_embedding)
                s = self.Xn @ qn
            else:
                qi = np.asarray(query_embedding["indices"])
                qv = np.asarray(query_embedding["values"], dtype=np.float32)
                order = np.argsort(np.abs(qv), kind="stable")
                keep = np.sort(order[int(np.floor(0.2 * len(qv))):])
                qd = np.zeros(g5_data.SPARSE_DIM, dtype=np.float64)
                qd[qi[keep]] = qv[keep]
                s = np.array([float(np.sum(qd[ri] * rv.astype(np.float64))) for ri, rv in sp_rows], dtype=np.float32)
            order = np.lexsort((np.arange(len(s)), -s))[:top_k]
            order = [int(i) for i in order if collection_name == "semantic_index" or s[i] > 0]
            return [{"id": g5_data.row_id(r), "content": g5_data.mmr_content(r), "score": float(s[r]),
                     "metadata": {"doc_id": f"doc{r // 10}", "chunk_index": r % 10}} for r in order]

    C.RetrievalConstants.TIMEOUT_SECONDS = 60.0
    for with_sparse in (True, False):
        mgr = NumpyManager(with_sparse)
        for profile in ("troubleshooting", "analysis", "summary"):
            for qi in (range(8) if with_sparse else (0, 5)):
                hr = R.HybridRetriever(index_manager=mgr, config=R.RetrievalConfig(top_k=20))
                mgr.q = (Q[qi], SQ[qi])
                mgr.seen_top_k = []
                res = asyncio.run(hr.retrieve("plain statement", profile_hint=profile))
                out["retrieve"].append({"with_sparse": with_sparse, "query": qi, "profile_hint": profile,
                                        "search_top_k": sorted(set(mgr.seen_top_k)),
                                        "ids": [o["id"] for o in res], "scores": [hexf(o["score"]) for o in res],
                                        "methods": [sorted(o["retrieval_methods"]) for o in res],
                                        "profile": res[0]["metadata"]["retrieval_profile"],
                                        "enable_mmr": hr.config.enable_mmr, "mmr_lambda": hr.config.mmr_lambda,
                                        "top_k": hr.config.top_k})
    dump("g11_mmr.json", out)


def g12():
    install_stand_ins()
    P = importlib.import_module("advanced_rag.pipeline")
    C = importlib.import_module("advanced_rag.constants")
    EC = importlib.import_module("advanced_rag.embedding_cache")
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs()
    N = X.shape[0]
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    sp_rows = [(idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]]) for r in range(N)]
    rows = [g5_data.payload_row(r) for r in range(N)]

    class Entity:
        def __init__(self, row):
            self.row = row

        def get(self, field):
            v = self.row[field]
            return float(np.float32(v)) if field in ("entropy", "redundancy", "domain_density") else v

    class Hit:
        def __init__(self, row, score):
            self.entity, self.score = Entity(row), float(score)

    class FakeCollection:
        def __init__(self, kind):
            self.kind = kind
            self.limits = []
            self.params = []

        def search(self, data, anns_field, param, limit, expr=None, output_fields=None):
            self.limits.append(limit)
            self.params.append(param)
            if self.kind == "dense":
                q = np.asarray(data[0], dtype=np.float32)
                s = Xn @ (q / np.linalg.norm(q))
            else:
                # drop_ratio_search is a parameter of the search (the retriever's sparse_search_params,
                # retrieval.py:63-66): the server ignores that share of the query's smallest entries
                drop = float((param.get("params") or {}).get("drop_ratio_search", 0.0))
                csr = data.tocsr()
                qi, qv = csr.indices, csr.data.astype(np.float32)
                order = np.argsort(np.abs(qv), kind="stable")
                keep = np.sort(order[int(np.floor(drop * len(qv))):])
                qd = np.zeros(g5_data.SPARSE_DIM, dtype=np.float64)
                qd[qi[keep]] = qv[keep]
                s = np.array([float(np.sum(qd[i] * v.astype(np.float64))) for i, v in sp_rows], dtype=np.float32)
            order = [int(i) for i in np.lexsort((np.arange(len(s)), -s)) if self.kind == "dense" or s[int(i)] > 0][:limit]
            return [[Hit(rows[i], s[i]) for i in order]]

        def release(self):
            pass

    class Gen:
        """Embeddings keyed by the trailing query number of the text ("... q<i>")."""

        def encode_semantic(self, text):
            return Q[int(text.rsplit("q", 1)[1])]

        def encode_sparse(self, text):
            qi, qv = SQ[int(text.rsplit("q", 1)[1])]
            return {"indices": qi.tolist(), "values": qv.astype(float).tolist()}

        def encode_domain(self, text, domain=None):
            return np.zeros(8, np.float32)

    C.RetrievalConstants.TIMEOUT_SECONDS = 60.0
    cases = []

    def run(label, query, with_sparse, rerank_top_k, context=None, enable_reranking=True, top_k=20):
        EC.initialize_caches() if hasattr(EC, "initialize_caches") else None
        pipe = P.AdvancedRAGPipeline(connect_to_milvus=False,
                                     config=P.PipelineConfig(enable_audit_logging=False, rerank_top_k=rerank_top_k,
                                                             enable_reranking=enable_reranking, top_k=top_k))
        im = pipe.index_manager
        im.semantic_dim, im.sparse_dim = X.shape[1], g5_data.SPARSE_DIM
        im.embedding_generator = Gen()
        im.collections["semantic_index"] = FakeCollection("dense")
        if with_sparse:
            im.collections["sparse_index"] = FakeCollection("sparse")
        else:
            im.collections.pop("sparse_index", None)
        pipe.retriever.config.enable_learned_ranker = True     # the deterministic rerank branch (retrieval.py:544-545)
        results, metrics = asyncio.run(pipe.retrieve(query, context=context))
        cases.append({
            "label": label, "query": query, "with_sparse": with_sparse, "context": context,
            "pipeline_rerank_top_k": rerank_top_k, "enable_reranking": enable_reranking, "top_k": top_k,
            "retriever_rerank_top_k": pipe.retriever.config.rerank_top_k,
            "search_limits": sorted(set(im.collections["semantic_index"].limits)),
            "sparse_search_params": im.collections["sparse_index"].params[0] if with_sparse else None,
            "n": len(results),
            "chunk_ids": [r.chunk_id for r in results], "scores": [hexf(r.score) for r in results],
            "retrieval_methods": [r.retrieval_method for r in results],
            "contents": [r.content for r in results],
            "doc_ids": [r.metadata["doc_id"] for r in results],
            "chunk_indexes": [r.metadata["chunk_index"] for r in results],
            "profiles": [r.metadata.get("retrieval_profile") for r in results],
            "metadata_keys": sorted(k for k in results[0].metadata if k != "recency") if results else [],
            "result_fields": sorted(results[0].__dataclass_fields__) if results else [],
            "audit_trails": [r.audit_trail for r in results],
            "metrics_type": type(metrics).__name__,
        })

    for qi in range(8):
        run(f"default-hybrid-q{qi}", f"plain statement q{qi}", True, 5)
    run("dense-only-q0", "plain statement q0", False, 5)
    run("dense-only-q6", "plain statement q6", False, 5)
    run("rerank-top-k-7-quirk-q1", "plain statement q1", True, 7)          # PipelineConfig.rerank_top_k decides, not 5
    run("rerank-top-k-12-q2", "plain statement q2", True, 12)
    run("profile-hint-default-on-a-question-q3", "what is q3", True, 5, {"retrieval_profile": "default"})
    run("reranking-disabled-slices-q4", "plain statement q4", True, 5, None, False)
    run("top-k-10-q5", "plain statement q5", True, 5, None, True, 10)
    dump("g12_pipeline.json", {"corpus": "g5 (tests/g5_data.py: seeds in g5_retrieve_c1.json)", "cases": cases})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    install_parent_package()
    R = importlib.import_module("advanced_rag.retrieval")
    C = importlib.import_module("advanced_rag.constants")
    g11(R, C)
    g12()
