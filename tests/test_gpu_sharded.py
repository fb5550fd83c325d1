"""A collection spread over several shard handles (MilvusIndexManager(devices=[...])) must answer exactly as one
shard holding all rows: dense, sparse, filters, tombstones, the fused retrieve() — and the reference's g5 outputs.
Several shards on ONE GPU stand in for one shard per GPU (the driver's box has one device); the code path — one
handle per entry of `devices`, per-shard search threads, host merge by (score desc, global row asc) — is the same."""
import asyncio

import numpy as np
import pytest

import g5_data
from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
from advanced_rag.constants import RetrievalConstants

pytestmark = pytest.mark.gpu


def _corpus(n, d, V, nnz, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[n // 3] = X[2 * n // 3]  # a tie that straddles shards
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    return X, (ptr, idx, val)


def _manager(devices, X, csr, batches):
    n, d = X.shape
    mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=600, dtype="float16", enable_domain=False, devices=devices)
    ptr, idx, val = csr
    lo = 0
    for b in batches:  # ragged appends: every batch is cut into per-shard pieces
        hi = min(n, lo + b)
        mgr.add_rows(X[lo:hi], (ptr[lo:hi + 1], idx, val), ids=[f"c{r}" for r in range(lo, hi)],
                     contents=[f"text {r}" for r in range(lo, hi)], doc_id=[f"doc{r % 7}" for r in range(lo, hi)],
                     entropy=[(r % 10) / 10 for r in range(lo, hi)])
        lo = hi
    assert lo == n
    mgr.finalize()
    return mgr


@pytest.fixture()
def long_timeout():
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    yield
    RetrievalConstants.TIMEOUT_SECONDS = old


def test_sharded_manager_equals_single_shard(gpu, long_timeout):
    n, d = 5000, 96
    X, csr = _corpus(n, d, 600, 10, seed=3)
    one = _manager([0], X, csr, [n])
    three = _manager([0, 0, 0], X, csr, [1700, 13, 900, 2387])
    assert [len(r) for r in three._main.rows_of] == [1667, 1667, 1666] or sum(len(r) for r in three._main.rows_of) == n
    assert three.get_collection_stats("semantic_index")["num_entities"] == n
    assert three.get_collection_stats("sparse_index")["num_entities"] == n
    rng = np.random.default_rng(9)
    try:
        for trial in range(6):
            q = X[n // 3] if trial == 0 else rng.standard_normal(d).astype(np.float32)
            sq = {"indices": sorted(rng.choice(600, 25, replace=False).tolist()),
                  "values": np.abs(rng.standard_normal(25)).astype(np.float32).tolist()}
            for flt in (None, 'doc_id == "doc3"', 'entropy >= 0.5 and doc_id != "doc1"'):
                for coll, emb, params in (("semantic_index", q, None),
                                          ("sparse_index", sq, {"metric_type": "IP", "params": {"drop_ratio_search": 0.2}})):
                    a = asyncio.run(one.search(emb, coll, 40, flt, params))
                    b = asyncio.run(three.search(emb, coll, 40, flt, params))
                    assert [h["id"] for h in a] == [h["id"] for h in b], (trial, coll, flt)
                    assert [h["score"] for h in a] == [h["score"] for h in b]
                    assert [h["_row"] for h in a] == [h["_row"] for h in b]
        # tombstones
        for m in (one, three):
            asyncio.run(m.delete_by_filter("semantic_index", 'doc_id == "doc2"'))
        a = asyncio.run(one.search(X[5], "semantic_index", 30))
        b = asyncio.run(three.search(X[5], "semantic_index", 30))
        assert [h["id"] for h in a] == [h["id"] for h in b] and all(h["metadata"]["doc_id"] != "doc2" for h in b)

        # the fused retrieve()
        class Gen:
            def encode_semantic(self, text):
                return X[int(text)]

            def encode_sparse(self, text):
                r = int(text)
                return {"indices": csr[1][r * 10:(r + 1) * 10].tolist(), "values": csr[2][r * 10:(r + 1) * 10].tolist()}

        from advanced_rag.embedding_cache import initialize_caches
        outs = []
        for m in (one, three):
            initialize_caches()
            m.embedding_generator = Gen()
            outs.append(asyncio.run(HybridRetriever(m, RetrievalConfig(top_k=20)).retrieve("77", profile_hint="default")))
        assert [o["id"] for o in outs[0]] == [o["id"] for o in outs[1]]
        assert [o["score"] for o in outs[0]] == [o["score"] for o in outs[1]]
        assert outs[1][0]["id"] == "c77"
    finally:
        asyncio.run(one.close())
        asyncio.run(three.close())


@pytest.mark.parametrize("n_shards", [2, 8])
def test_g5_reference_runs_on_two_and_eight_shards(gpu, long_timeout, n_shards):
    """The reference's own retrieve() outputs (golden g5) through a manager whose rows sit on two shards, and on eight
    (BASELINE config 4's shard count; all of them on the one GPU of the test box)."""
    from test_gpu_golden import _g5_manager, _run_g5
    g, X, csr, Q, SQ = g5_data.inputs()
    mgr = _g5_manager("float32", X, csr, True, devices=[0] * n_shards)
    try:
        assert mgr._main.n_shards == n_shards and min(len(r) for r in mgr._main.rows_of) == 1000 // n_shards
        for run in (r for r in g["runs"] if r["with_sparse"]):
            out = _run_g5(mgr, Q, SQ, run)
            assert [o["id"] for o in out] == run["ids"]
            assert [float(o["score"]).hex() for o in out] == run["scores"]
            assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
    finally:
        asyncio.run(mgr.close())


def test_sharded_snapshot_roundtrip(gpu, tmp_path, long_timeout):
    n, d = 1200, 64
    X, csr = _corpus(n, d, 600, 8, seed=5)
    mgr = _manager([0, 0], X, csr, [700, 500])
    q = X[11]
    before = asyncio.run(mgr.search(q, "semantic_index", 15, 'doc_id == "doc4"'))
    mgr.save_snapshot(str(tmp_path))
    fresh = MilvusIndexManager(semantic_dim=d, sparse_dim=600, dtype="float16", enable_domain=False, devices=[0, 0])
    fresh.load_snapshot(str(tmp_path))
    after = asyncio.run(fresh.search(q, "semantic_index", 15, 'doc_id == "doc4"'))
    assert [h["id"] for h in before] == [h["id"] for h in after] and [h["score"] for h in before] == [h["score"] for h in after]
    with pytest.raises(ValueError):
        MilvusIndexManager(semantic_dim=d, sparse_dim=600, dtype="float16", enable_domain=False).load_snapshot(str(tmp_path))
    asyncio.run(mgr.close())
    asyncio.run(fresh.close())


def test_bad_sparse_payloads_do_not_shift_rows(gpu, long_timeout):
    """A plugin encoder that returns duplicate, out-of-range or non-finite sparse entries for some chunks: those chunks
    lose their sparse row (reported), every other row keeps its number in all collections."""
    from advanced_rag import AdvancedRAGPipeline, PipelineConfig
    docs = [{"id": f"d{i}", "text": f"alpha{i} beta{i % 3} gamma. " * 4, "metadata": {}} for i in range(12)]
    rng = np.random.default_rng(0)
    vec = {}

    class Gen:
        calls = 0

        def encode_semantic(self, text):
            return vec.setdefault(text, rng.standard_normal(32).astype(np.float32))

        def encode_sparse(self, text):
            Gen.calls += 1
            base = {"indices": [3, 7, 11 + Gen.calls % 50], "values": [1.0, 0.5, 2.0]}
            if Gen.calls == 2:
                return {"indices": [5, 5, 9], "values": [1.0, 2.0, 1.0]}       # duplicate: merged, accepted
            if Gen.calls == 4:
                return {"indices": [1, 999999], "values": [1.0, 1.0]}          # out of range: rejected
            if Gen.calls == 6:
                return {"indices": [2], "values": [float("nan")]}              # non-finite: rejected
            return base

        def encode_domain(self, text, domain=None):
            return self.encode_semantic(text)[:16].copy()

    p = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False), semantic_dim=32, sparse_dim=256, domain_dim=16,
                            devices=[0, 0])
    p.index_manager.embedding_generator = Gen()
    rep = asyncio.run(p.ingest_documents(docs))["indexing_summary"]
    mgr = p.index_manager
    n = rep["total_chunks"]
    assert rep["indexed_semantic"] == n and mgr._main.num_rows == n == mgr._main.num_sparse_rows == mgr._domain.num_rows
    assert rep["indexed_sparse"] == n - 2 and sum("sparse_embedding_failed" in str(e) for e in rep["errors"]) == 2
    for r in (0, n // 2, n - 1):  # a row's dense vector finds its own payload
        hit = asyncio.run(mgr.search(vec[mgr._cols["content"][r]], "semantic_index", 1))[0]
        assert hit["id"] == mgr._cols["id"][r]
    sp = asyncio.run(mgr.search({"indices": [5, 9], "values": [1.0, 1.0]}, "sparse_index", 3))
    assert sp and sp[0]["id"] == mgr._cols["id"][1] and abs(sp[0]["score"] - 4.0) < 1e-6   # the merged duplicate: (1+2)*1 + 1*1
    asyncio.run(p.close())
