"""The HIP path against outputs the REFERENCE itself produced (tests/golden/*.json, generated from the
imported reference by tests/golden/gen_golden.py and gen_golden_g6.py) — not against the build's own
oracle.  Everything goes through the C ABI of libhbmrag.so:

  g5  HybridRetriever.retrieve on BASELINE config 1 (reference retrieval.py:215-491): a real
      MilvusIndexManager on the device driven by this package's HybridRetriever must return the
      reference's ids (bit-exact), float64 fused scores (bit-equal), method tags and profile.
  g1  _fuse_results (retrieval.py:421-491): every case through hr_fuse_rrf and hr_fuse_rrf_dev.
  g2  rerank, learned-ranker branch (retrieval.py:518-563, ranker.py:109-125) through hr_rerank_linear_dev.
  g6  MilvusIndexManager.search hit formatting (indexing.py:533-551) through the device search.
"""
import asyncio
import json
import os

import numpy as np
import pytest

import g5_data
from advanced_rag import _native as nat
from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
from advanced_rag.constants import RetrievalConstants

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _intern(*lists):
    table = {}
    out = [[table.setdefault(x, len(table)) for x in lst] for lst in lists]
    return out, {v: k for k, v in table.items()}


NAMES = ("semantic", "sparse", "domain")


def method_names(mask):
    return sorted(n for bit, n in enumerate(NAMES) if (int(mask) >> bit) & 1)


# --------------------------------------------------------------------------- g5
def _g5_manager(dtype, X, sparse_csr, with_sparse, **kw):
    N, D = X.shape
    mgr = MilvusIndexManager(semantic_dim=D, sparse_dim=g5_data.SPARSE_DIM, dtype=dtype, enable_domain=False, **kw)
    if not with_sparse:  # BASELINE config 1 proper: dense only -> _search_sparse returns [] (retrieval.py:376-378)
        del mgr.collections["sparse_index"]
    mgr.add_rows(X, sparse_csr if with_sparse else None, ids=[g5_data.row_id(r) for r in range(N)],
                 contents=[f"row {r}" for r in range(N)])
    mgr.finalize()
    return mgr


def _run_g5(mgr, Q, SQ, run):
    q = run["query"]

    class Gen:
        def encode_semantic(self, text):
            return Q[q]

        def encode_sparse(self, text):
            return {"indices": SQ[q][0].tolist(), "values": SQ[q][1].tolist()}

    mgr.embedding_generator = Gen()
    from advanced_rag.embedding_cache import initialize_caches
    initialize_caches()  # the query text is the same for every run: no stale cached embedding
    retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
    return asyncio.run(retr.retrieve("plain statement", profile_hint="default"))


@pytest.fixture()
def long_timeout():
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    yield
    RetrievalConstants.TIMEOUT_SECONDS = old


def test_g5_retrieve_on_device_fp32_matches_reference(gpu, long_timeout):
    """fp32 shard (config 1's storage type): all 16 reference runs, ids / fused scores / methods / profile."""
    g, X, csr, Q, SQ = g5_data.inputs()
    for with_sparse in (False, True):
        mgr = _g5_manager("float32", X, csr, with_sparse)
        try:
            for run in (r for r in g["runs"] if r["with_sparse"] == with_sparse):
                out = _run_g5(mgr, Q, SQ, run)
                assert [o["id"] for o in out] == run["ids"], (with_sparse, run["query"])
                assert [float(o["score"]).hex() for o in out] == run["scores"]
                assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
                assert out[0]["metadata"]["retrieval_profile"] == run["profile"]
                assert all(o["content"] == f"row {int(o['id'].rsplit('::', 1)[1], 16)}" for o in out)
        finally:
            asyncio.run(mgr.close())


def test_g5_retrieve_on_device_fp16_reports_rank_flips(gpu, long_timeout):
    """The same corpus rounded to fp16 rows (the storage type of configs 3-5).  Rounding the corpus perturbs
    cosine scores by ~1e-4 relative, so a rank flip against the fp32 reference would be possible where two scores
    are closer than that — on this corpus there is none: all 16 runs return the reference's ids and fused scores (the
    kernels are deterministic, so this is a property of the fixture, asserted as such; a flip, should a change ever
    produce one, is reported with its fp32 score gap)."""
    g, X, csr, Q, SQ = g5_data.inputs()
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    identical, flips = 0, []
    for with_sparse in (False, True):
        mgr = _g5_manager("float16", X, csr, with_sparse)
        try:
            for run in (r for r in g["runs"] if r["with_sparse"] == with_sparse):
                out = _run_g5(mgr, Q, SQ, run)
                got = [o["id"] for o in out]
                if got == run["ids"]:
                    identical += 1
                    assert [float(o["score"]).hex() for o in out] == run["scores"]
                    continue
                s32 = Xn @ (Q[run["query"]] / np.linalg.norm(Q[run["query"]]))
                for pos, (a, b) in enumerate(zip(got, run["ids"])):
                    if a != b:
                        ra, rb = int(a.rsplit("::", 1)[1], 16), int(b.rsplit("::", 1)[1], 16)
                        flips.append((with_sparse, run["query"], pos, a, b, abs(float(s32[ra]) - float(s32[rb]))))
        finally:
            asyncio.run(mgr.close())
    print(f"g5 fp16: {identical}/16 runs identical to the fp32 reference; flips (sparse, query, pos, got, want, fp32 gap): {flips}")
    assert identical == 16 and not flips, flips


def test_g13_retrieve_on_device_at_config2_size_matches_reference(gpu, long_timeout):
    """G13: the reference's retrieve() at 100 000 x 384 fp32 (BASELINE config 2's size) through the HIP path — fp32 shard,
    dense-only and hybrid, all 16 runs: ids, float64 fused scores bit for bit, method tags."""
    g, X, csr, Q, SQ = g5_data.inputs(g5_data.load_g13())
    for with_sparse in (False, True):
        mgr = _g5_manager("float32", X, csr, with_sparse)
        try:
            for run in (r for r in g["runs"] if r["with_sparse"] == with_sparse):
                out = _run_g5(mgr, Q, SQ, run)
                assert [o["id"] for o in out] == run["ids"], (with_sparse, run["query"])
                assert [float(o["score"]).hex() for o in out] == run["scores"]
                assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
        finally:
            asyncio.run(mgr.close())


# --------------------------------------------------------------------------- g1 / g2
def test_g1_every_reference_fusion_case_through_the_rrf_kernel(gpu):
    cases = gold("g1_fuse.json")
    assert len(cases) >= 20
    h = nat.ShardHandle(8)
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for c in cases:
        (a, b, d), back = _intern(c["semantic"], c["sparse"], c["domain"])
        # host-buffer form
        ids, scores, methods = h.fuse_rrf(a, b, d, c["dense_weight"], c["sparse_weight"], 0.2, 60)
        assert [back[int(i)] for i in ids] == c["ids"], c["label"]
        assert [float(s).hex() for s in scores] == c["scores"], c["label"]
        assert [method_names(m) for m in methods] == c["methods"], c["label"]
        # device form (batch of one; a list must be readable, so empty lists become one -1 entry)
        total = len(a) + len(b) + len(d)
        if total == 0:
            continue
        ta = torch.tensor(a or [-1], dtype=torch.int64, device=dev)
        tb = torch.tensor(b or [-1], dtype=torch.int64, device=dev)
        td = torch.tensor(d or [-1], dtype=torch.int64, device=dev)
        fo = torch.empty((1, total), dtype=torch.int64, device=dev)
        fs = torch.empty((1, total), dtype=torch.float64, device=dev)
        fm = torch.empty((1, total), dtype=torch.int32, device=dev)
        fn = torch.empty((1,), dtype=torch.int32, device=dev)
        nat.fuse_rrf_dev(ta.data_ptr(), ta.numel(), tb.data_ptr(), tb.numel(), td.data_ptr() if d else 0,
                         td.numel() if d else 0, 1, c["dense_weight"], c["sparse_weight"], 0.2, 60, total,
                         fo.data_ptr(), fs.data_ptr(), fm.data_ptr(), fn.data_ptr(), st)
        torch.cuda.synchronize()
        n = int(fn[0])
        assert [back[int(i)] for i in fo[0, :n].cpu()] == c["ids"], c["label"]
        assert [float(s).hex() for s in fs[0, :n].cpu()] == c["scores"], c["label"]
        assert [method_names(m) for m in fm[0, :n].cpu()] == c["methods"], c["label"]
    h.close()


def test_g2_learned_ranker_rerank_through_the_kernel(gpu):
    c = {x["label"]: x for x in gold("g2_rerank.json")}["learned-ranker"]
    (a, b), back = _intern(c["semantic"], c["sparse"])
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    total = len(a) + len(b)
    ta = torch.tensor(a, dtype=torch.int64, device=dev)
    tb = torch.tensor(b, dtype=torch.int64, device=dev)
    fo = torch.empty((1, total), dtype=torch.int64, device=dev)
    fs = torch.empty((1, total), dtype=torch.float64, device=dev)
    fm = torch.empty((1, total), dtype=torch.int32, device=dev)
    fn = torch.empty((1,), dtype=torch.int32, device=dev)
    nat.fuse_rrf_dev(ta.data_ptr(), len(a), tb.data_ptr(), len(b), 0, 0, 1, 0.7, 0.3, 0.2, 60, total, fo.data_ptr(),
                     fs.data_ptr(), fm.data_ptr(), fn.data_ptr(), st)
    k = c["top_k"]
    ro = torch.empty((1, k), dtype=torch.int64, device=dev)
    rs = torch.empty((1, k), dtype=torch.float64, device=dev)
    rorig = torch.empty((1, k), dtype=torch.float64, device=dev)
    # LearnedRankerConfig defaults (reference ranker.py:27-30): base 1.0, method bonus 0.1, recency 0.0
    nat.rerank_linear_dev(fo.data_ptr(), fs.data_ptr(), fm.data_ptr(), fn.data_ptr(), 1, total, 1.0, 0.1, 0.0, k,
                          ro.data_ptr(), rs.data_ptr(), rorig.data_ptr(), st)
    torch.cuda.synchronize()
    assert [back[int(i)] for i in ro[0].cpu()] == c["ids"]
    assert [float(s).hex() for s in rs[0].cpu()] == c["scores"]
    assert [float(s).hex() for s in rorig[0].cpu()] == c["original"]


# --------------------------------------------------------------------------- g6
def test_g6_search_results_have_the_references_shape_on_device(gpu):
    g = gold("g6_search_format.json")
    rows = g["rows"]
    dense = np.asarray(g["dense"], dtype=np.float32)
    sp = g["sparse"]
    indptr = np.cumsum([0] + [len(s["indices"]) for s in sp]).astype(np.int64)
    csr = (indptr, np.concatenate([np.asarray(s["indices"], np.int32) for s in sp]),
           np.concatenate([np.asarray(s["values"], np.float32) for s in sp]))
    mgr = MilvusIndexManager(semantic_dim=g["dim"], sparse_dim=g["sparse_dim"], dtype="float32", enable_domain=False)
    cols = {k: [r[k] for r in rows] for k in ("doc_id", "chunk_index", "token_count", "entropy", "redundancy",
                                                "domain_density", "timestamp", "metadata_json")}
    mgr.add_rows(dense, csr, ids=[r["chunk_id"] for r in rows], contents=[r["content"] for r in rows], **cols)
    mgr.finalize()
    try:
        for c in g["cases"]:
            q = np.asarray(c["query"], np.float32) if c["collection"] != "sparse_index" else c["query"]
            kw = {} if c["search_params"] is None else {"search_params": c["search_params"]}
            out = asyncio.run(mgr.search(q, c["collection"], top_k=c["top_k"], filters=c["filters"], **kw))
            want = c["results"]
            assert len(out) == len(want), c["label"]
            for o, w in zip(out, want):
                o = dict(o)
                o.pop("_row")  # the one extra key: local row number used by the device-side rank fusion
                assert list(o) == list(w) and list(o["metadata"]) == list(w["metadata"]), c["label"]
                assert o["id"] == w["id"] and o["content"] == w["content"] and o["metadata"] == w["metadata"], c["label"]
                assert type(o["score"]) is float and abs(o["score"] - w["score"]) <= 1e-4, c["label"]
        for e in g["errors"]:
            with pytest.raises(ValueError) as ei:
                asyncio.run(mgr.search(np.zeros(4, np.float32), "nope" if e["label"] == "unknown-collection" else "sparse_index"))
            assert str(ei.value) == e["message"]
    finally:
        asyncio.run(mgr.close())


# --------------------------------------------------------------------------- G11 / G12 (round 4)
def _mmr_manager(dtype, X, sparse_csr, with_sparse, **kw):
    """Config 1 with the contents and scalar fields of the G11 / G12 fixtures (g5_data.mmr_content / payload_row)."""
    N, D = X.shape
    mgr = MilvusIndexManager(semantic_dim=D, sparse_dim=g5_data.SPARSE_DIM, dtype=dtype, enable_domain=False, **kw)
    if not with_sparse:
        del mgr.collections["sparse_index"]
    _fill_g12_rows(mgr, X, sparse_csr if with_sparse else None)
    return mgr


def _fill_g12_rows(mgr, X, sparse_csr):
    rows = [g5_data.payload_row(r) for r in range(X.shape[0])]
    cols = {k: [row[k] for row in rows] for k in ("doc_id", "chunk_index", "token_count", "entropy", "redundancy",
                                                   "domain_density", "timestamp", "metadata_json")}
    mgr.add_rows(X, sparse_csr, ids=[row["chunk_id"] for row in rows], contents=[row["content"] for row in rows], **cols)
    mgr.finalize()


class _KeyedGen:
    """Embeddings keyed by the trailing query number of the text ("... q<i>"), as in gen_golden_g11_g12.py."""

    def __init__(self, Q, SQ, fixed=None):
        self.Q, self.SQ, self.fixed = Q, SQ, fixed

    def _i(self, text):
        return self.fixed if self.fixed is not None else int(text.rsplit("q", 1)[1])

    def encode_semantic(self, text):
        return self.Q[self._i(text)]

    def encode_sparse(self, text):
        qi, qv = self.SQ[self._i(text)]
        return {"indices": qi.tolist(), "values": qv.astype(float).tolist()}

    def encode_domain(self, text, domain=None):
        return np.zeros(8, np.float32)


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_g11_retrieve_under_mmr_profiles_on_device_matches_reference(gpu, long_timeout, dtype):
    """G11b on the device: the reference's HybridRetriever.retrieve under profile_hint = troubleshooting / analysis /
    summary (k' = 60 / 60 / 80; MMR at lambda 0.5 / 0.8 / off, reference retrieval.py:142-213, :488-516) — ids in order,
    float64 fused scores bit for bit, method tags, profile; the MMR profiles take the general path of the front (the
    whole fused list is needed), `summary` the one-round path."""
    from advanced_rag.embedding_cache import initialize_caches
    g, X, csr, Q, SQ = g5_data.inputs()
    runs = gold("g11_mmr.json")["retrieve"]
    flips = []
    for with_sparse in (True, False):
        mgr = _mmr_manager(dtype, X, csr, with_sparse)
        try:
            for run in (r for r in runs if r["with_sparse"] == with_sparse):
                initialize_caches()
                mgr.embedding_generator = _KeyedGen(Q, SQ, fixed=run["query"])
                retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
                out = asyncio.run(retr.retrieve("plain statement", profile_hint=run["profile_hint"]))
                want = (run["ids"], run["scores"], run["methods"])
                if dtype == "float16" and ([o["id"] for o in out] != run["ids"] or [float(o["score"]).hex() for o in out] != run["scores"]):
                    # rows rounded to fp16 move cosine scores by ~1e-4 relative: deep in a 60- / 80-long list two neighbours
                    # may swap against the fp32 reference (k' = 40 lists never did, test_g5_..._fp16) — which shows as other ids, or,
                    # when only one of the two rows survives the MMR cut, as another fused score of that row.  Such a run is held
                    # against the oracle chain on the SAME fp16 rows instead (ids, fused scores, methods: exact), and the
                    # swap must be explained by an fp32 score gap below the rounding.
                    import oracle
                    X16 = X.astype(np.float16)
                    kp, top_k = 2 * run["top_k"], run["top_k"]
                    di, _ = oracle.dense_search(X16, Q[run["query"]:run["query"] + 1], kp, oracle.COSINE)
                    sl = ()
                    if with_sparse:
                        si, _ = oracle.sparse_search(csr[0], csr[1], csr[2], [SQ[run["query"]]], kp, 0.2)
                        sl = si[0][si[0] >= 0]
                    ids, scores, methods = oracle.rrf(di[0], sl, (), 0.7, 0.3, 0.2, 60)
                    sel = (oracle.mmr(ids, [float(x) for x in scores], [g5_data.mmr_content(int(r)) for r in ids], top_k, run["mmr_lambda"])
                           if run["enable_mmr"] else list(range(len(ids))))[:top_k]
                    want = ([g5_data.row_id(int(ids[i])) for i in sel], [float(scores[i]).hex() for i in sel],
                            [method_names(methods[i]) for i in sel])
                    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
                    s32 = Xn @ (Q[run["query"]] / np.linalg.norm(Q[run["query"]]))
                    d32 = np.lexsort((np.arange(len(s32)), -s32))[:kp]
                    gaps = [abs(float(s32[x]) - float(s32[y])) for x, y in zip(d32, di[0]) if x != y]   # rows that changed places
                    assert gaps and max(gaps) < 5e-4, gaps
                    flips.append((with_sparse, run["profile_hint"], run["query"]))
                assert [o["id"] for o in out] == want[0], (with_sparse, run["profile_hint"], run["query"])
                assert [float(o["score"]).hex() for o in out] == want[1]
                assert [sorted(o["retrieval_methods"]) for o in out] == want[2]
                assert out[0]["metadata"]["retrieval_profile"] == run["profile"]
                assert (retr.config.enable_mmr, retr.config.mmr_lambda, retr.config.top_k) == (run["enable_mmr"], run["mmr_lambda"], run["top_k"])
        finally:
            asyncio.run(mgr.close())
    print(f"g11 {dtype}: {len(runs) - len(flips)}/{len(runs)} runs identical to the fp32 reference; runs with a near-tie swap: {flips}")
    assert dtype == "float16" or not flips
    assert len(flips) <= len(runs) // 3


def test_g12_pipeline_retrieve_on_device_matches_reference(gpu, long_timeout):
    """G12 on the device: this package's AdvancedRAGPipeline.retrieve() with the learned ranker on, over an in-HBM shard
    holding config 1, returns what the REFERENCE's AdvancedRAGPipeline(connect_to_milvus=False).retrieve() returned over
    fake Milvus collections with the same rows (reference pipeline.py:217-309): chunk_id order, float64 scores bit for
    bit, retrieval_method, contents, metadata keys and values, the RetrievalResult fields, and the rerank_top_k quirk."""
    from advanced_rag import AdvancedRAGPipeline, PipelineConfig
    from advanced_rag.embedding_cache import initialize_caches
    g, X, csr, Q, SQ = g5_data.inputs()
    cases = gold("g12_pipeline.json")["cases"]
    for c in cases:
        initialize_caches()
        # (connect_to_milvus=True is what creates the collections here — in HBM; the reference's run used
        # connect_to_milvus=False and put fake collections into the manager)
        pipe = AdvancedRAGPipeline(connect_to_milvus=True,
                                   config=PipelineConfig(enable_audit_logging=False, rerank_top_k=c["pipeline_rerank_top_k"],
                                                         enable_reranking=c["enable_reranking"], top_k=c["top_k"]),
                                   semantic_dim=X.shape[1], sparse_dim=g5_data.SPARSE_DIM, dtype="float32", enable_domain=False)
        mgr = pipe.index_manager
        try:
            if not c["with_sparse"]:
                del mgr.collections["sparse_index"]
            _fill_g12_rows(mgr, X, csr if c["with_sparse"] else None)
            mgr.embedding_generator = _KeyedGen(Q, SQ)
            pipe.retriever.config.enable_learned_ranker = True
            assert pipe.retriever.config.rerank_top_k == c["retriever_rerank_top_k"]      # never forwarded (quirk kept)
            results, metrics = asyncio.run(pipe.retrieve(c["query"], context=c["context"]))
            assert len(results) == c["n"], c["label"]
            assert [r.chunk_id for r in results] == c["chunk_ids"], c["label"]
            assert [float(r.score).hex() for r in results] == c["scores"], c["label"]
            assert [r.retrieval_method for r in results] == c["retrieval_methods"]
            assert [r.content for r in results] == c["contents"]
            assert [r.metadata["doc_id"] for r in results] == c["doc_ids"]
            assert [r.metadata["chunk_index"] for r in results] == c["chunk_indexes"]
            assert [r.metadata.get("retrieval_profile") for r in results] == c["profiles"]
            assert sorted(k for k in results[0].metadata if k != "recency") == c["metadata_keys"]
            assert sorted(results[0].__dataclass_fields__) == c["result_fields"]
            assert [r.audit_trail for r in results] == c["audit_trails"]
            assert type(metrics).__name__ == c["metrics_type"]
            for r in results:       # FLOAT fields come back as the float32 value, as pymilvus hands them over
                row = g5_data.payload_row(int(r.chunk_id.rsplit("::", 1)[1], 16))
                assert (r.metadata["entropy"], r.metadata["redundancy"], r.metadata["domain_density"], r.metadata["timestamp"]) == \
                       (row["entropy"], row["redundancy"], row["domain_density"], row["timestamp"])
        finally:
            asyncio.run(pipe.close())
