"""The batching front of MilvusIndexManager.search (advanced_rag/batching.py): concurrent retrieve() coroutines —
the reference's concurrency model, service.py:136,149 + retrieval.py:293-306 — are packed into batched device
searches and must get exactly what sequential single-query calls get."""
import asyncio

import numpy as np
import pytest

import g5_data
from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
from advanced_rag.constants import RetrievalConstants
from test_gpu_golden import _g5_manager

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture()
def long_timeout():
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 120.0
    yield
    RetrievalConstants.TIMEOUT_SECONDS = old


class TableGen:
    """Embedding generator keyed by the query text "q<i>"."""

    def __init__(self, Q, SQ):
        self.Q, self.SQ = Q, SQ

    def encode_semantic(self, text):
        return self.Q[int(text[1:])]

    def encode_sparse(self, text):
        qi, qv = self.SQ[int(text[1:])]
        return {"indices": qi.tolist(), "values": qv.tolist()}

    def encode_domain(self, text, domain=None):
        return np.zeros(8, np.float32)


def _strip(hits):
    """Everything a hit dict of retrieve() carries, in comparable form."""
    return [(h["id"], float(h["score"]).hex(), tuple(h["retrieval_methods"]), h["metadata"]["chunk_index"], h["method"],
             float(h["original_score"]).hex(), h["content"], h["metadata"]["doc_id"], h["metadata"].get("retrieval_profile"),
             tuple(sorted(h)))
            for h in hits]


@pytest.mark.parametrize("one_round", [True, False])
def test_g5_reference_runs_issued_concurrently(gpu, long_timeout, one_round):
    """The 8 hybrid g5 runs of the reference, issued as concurrent coroutines through ONE manager: every coroutine gets
    the reference's ids / fused scores / methods although its searches shared launches with the others — as one "hybrid"
    request per retrieve() (both searches + the fusion in one round) and as two searches + a fusion round."""
    from advanced_rag.embedding_cache import initialize_caches
    g, X, csr, Q, SQ = g5_data.inputs()
    mgr = _g5_manager("float32", X, csr, True)
    initialize_caches()
    mgr.embedding_generator = TableGen(Q, SQ)
    if not one_round:
        mgr.hybrid_search = None
    retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
    runs = [r for r in g["runs"] if r["with_sparse"]]

    async def go():
        return await asyncio.gather(*[retr.retrieve(f"q{r['query']}", profile_hint="default") for r in runs])

    try:
        outs = asyncio.run(go())
        for run, out in zip(runs, outs):
            assert [o["id"] for o in out] == run["ids"], run["query"]
            assert [float(o["score"]).hex() for o in out] == run["scores"]
            assert [sorted(o["retrieval_methods"]) for o in out] == run["methods"]
        st = mgr._front.stats
        if one_round:
            assert st["requests"] == len(runs) + 3 * st["redone_unproven"]   # one request per retrieve()
            assert 1 <= st["hybrid_launches"] < len(runs) and st["fuse_launches"] == 0
        else:
            assert st["requests"] == 3 * len(runs)                    # dense + sparse + fusion per retrieve()
            assert st["hybrid_launches"] == 0
        assert st["dense_launches"] + st["sparse_launches"] < 2 * len(runs)   # they did share launches
    finally:
        asyncio.run(mgr.close())


def test_128_concurrent_retrieves_equal_sequential_calls(gpu, long_timeout):
    from advanced_rag.embedding_cache import initialize_caches
    rng = np.random.default_rng(3)
    n, d, V, nq = 50000, 96, 2000, 128
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[40000] = X[17]  # a tie
    idx = np.sort(np.argpartition(rng.random((n, V)), 19, axis=1)[:, :20], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 20)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * 20
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[5] = X[17]
    SQ = [(np.sort(rng.choice(V, 40, replace=False)).astype(np.int32), np.abs(rng.standard_normal(40)).astype(np.float32))
          for _ in range(nq)]
    results = {}
    for coalesce, one_round in ((False, False), (True, False), (True, True)):
        initialize_caches()
        mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float16", enable_domain=False, coalesce=coalesce)
        mgr.add_rows(X, (ptr, idx, val), chunk_index=(np.arange(n) % 10).tolist())
        mgr.finalize()
        mgr.embedding_generator = TableGen(Q, SQ)
        if not one_round:
            mgr.hybrid_search = None
        retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))

        async def sequential():
            return [await retr.retrieve(f"q{i}", profile_hint="default") for i in range(nq)]

        async def concurrent():
            plain = [retr.retrieve(f"q{i}", profile_hint="default") for i in range(nq)]
            filtered = [retr.retrieve(f"q{i}", filters={"chunk_index": {"$lt": 5}}, profile_hint="default") for i in range(0, nq, 4)]
            return await asyncio.gather(*plain, *filtered)

        async def wide():   # top_k = 100: lists of 200 per modality (k' = 200 of HR_MAX_TOPK = 256), 304 candidate groups
            r100 = HybridRetriever(mgr, RetrievalConfig(top_k=100))
            return await asyncio.gather(*[r100.retrieve(f"q{i}", profile_hint="default") for i in range(0, nq, 16)])

        try:
            results.setdefault("wide", {})[(coalesce, one_round)] = [_strip(o) for o in asyncio.run(wide())]
            if coalesce:
                st0 = dict(mgr._front.stats)
                outs = asyncio.run(concurrent())
                tag = "one_round" if one_round else "concurrent"
                results[tag] = [_strip(o) for o in outs[:nq]]
                results[tag + "_filtered"] = [_strip(o) for o in outs[nq:]]
                st = {k: (v - st0[k] if k != "max_batch_seen" else v) for k, v in mgr._front.stats.items()}
                assert st["dense_launches"] + st["sparse_launches"] <= (2 * (nq + nq // 4)) // 4, st   # >= 4 queries per launch
                assert st["max_batch_seen"] >= 16, st
                assert (st["hybrid_launches"] > 0) == one_round and (st["fuse_launches"] == 0 or not one_round or st["redone_unproven"] > 0)
                if one_round:
                    # groups of EQUAL size in one round (8 plain + 8 filtered requests) and two chunks of one group (136 + 8
                    # plain requests: chunks of 128 and 16): the engine's per-batch-size buffers must not be shared between
                    # launches that are read back together
                    async def same_size():
                        plain = [retr.retrieve(f"q{i}", profile_hint="default") for i in list(range(8)) + list(range(nq))]
                        filt = [retr.retrieve(f"q{i}", filters={"chunk_index": {"$lt": 5}}, profile_hint="default") for i in range(0, 32, 4)]
                        return await asyncio.gather(*plain, *filt)
                    o2 = asyncio.run(same_size())
                    results["same_size_groups"] = ([_strip(o) for o in o2[:8 + nq]], [_strip(o) for o in o2[8 + nq:]])
                seq2 = asyncio.run(sequential())          # the front also serves one caller at a time
                got2 = [_strip(o) for o in seq2]
                bad = [(i, j, a, b) for i, (x, y) in enumerate(zip(got2, results[tag])) for j, (a, b) in enumerate(zip(x, y)) if a != b]
                assert not bad and got2 == results[tag], (len(bad), bad[:2])
            else:
                results["sequential"] = [_strip(o) for o in asyncio.run(sequential())]

                async def filtered_seq():
                    return [await retr.retrieve(f"q{i}", filters={"chunk_index": {"$lt": 5}}, profile_hint="default")
                            for i in range(0, nq, 4)]
                results["filtered_sequential"] = [_strip(o) for o in asyncio.run(filtered_seq())]
        finally:
            asyncio.run(mgr.close())
    assert all(len(r) == 20 for r in results["sequential"])
    wide = results["wide"]
    assert all(len(r) == 100 for r in wide[(False, False)])
    assert wide[(True, False)] == wide[(False, False)] and wide[(True, True)] == wide[(False, False)]
    bad = [(i, j, a, b) for i, (x, y) in enumerate(zip(results["concurrent"], results["sequential"])) for j, (a, b) in enumerate(zip(x, y)) if a != b]
    assert not bad, (len(bad), bad[:2])
    assert results["concurrent"] == results["sequential"]
    assert results["concurrent_filtered"] == results["filtered_sequential"]
    assert results["same_size_groups"][0] == results["sequential"][:8] + results["sequential"]
    assert results["same_size_groups"][1] == results["filtered_sequential"][:8]
    assert results["one_round"] == results["sequential"]
    assert results["one_round_filtered"] == results["filtered_sequential"]
    assert all(r[3] < 5 for lst in results["one_round_filtered"] for r in lst)


def test_one_round_requests_with_other_profiles_weights_and_a_bad_query(gpu, long_timeout):
    """What the one-round path must hand to the general path or key apart: a weight adapter (its own engine per weight
    pair), a profile with MMR (needs the whole fused list), a malformed sparse query inside a batch (that retrieve()
    degrades to its dense hits, the others are untouched), an empty sparse query."""
    from advanced_rag.embedding_cache import initialize_caches
    rng = np.random.default_rng(11)
    n, d, V, nq = 20000, 64, 500, 24
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = np.sort(np.argpartition(rng.random((n, V)), 9, axis=1)[:, :10], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 10)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    SQ = [(np.sort(rng.choice(V, 12, replace=False)).astype(np.int32), np.abs(rng.standard_normal(12)).astype(np.float32))
          for _ in range(nq)]
    SQ[3] = (np.array([7, 7, 9], np.int32), np.array([1.0, 0.5, 0.25], np.float32))     # duplicate index: refused
    SQ[4] = (np.zeros(0, np.int32), np.zeros(0, np.float32))                            # no terms at all
    outs = {}
    for one_round in (False, True):
        initialize_caches()
        mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float32", enable_domain=False)
        mgr.add_rows(X, (np.arange(n + 1, dtype=np.int64) * 10, idx, val), chunk_index=(np.arange(n) % 10).tolist())
        mgr.finalize()
        mgr.embedding_generator = TableGen(Q, SQ)
        if not one_round:
            mgr.hybrid_search = None
        retr = HybridRetriever(mgr, RetrievalConfig(top_k=10), weight_adapter=lambda q: (0.5, 0.5) if int(q[1:]) % 2 else (0.9, 0.2))

        async def go():
            # (concurrent requests share the retriever's active profile and adapted weights, as in the reference: only
            # the shape of the answers is checked here)
            plain = [retr.retrieve(f"q{i}", profile_hint="default") for i in range(nq)]
            return await asyncio.gather(*plain)

        hints = ("default", "faq", "troubleshooting")    # top_k 10 / 10 / 30 with MMR
        try:
            # the adapter rewrites the shared config per request (as the reference does): sequential calls, so that both
            # paths see the same weights for the same query
            seq = [asyncio.run(retr.retrieve(f"q{i}", profile_hint=hints[i % 3])) for i in range(nq)]
            outs[one_round] = [_strip(o) for o in seq]
            conc = asyncio.run(go())
            assert [len(o) for o in conc] == [10] * nq
            if one_round:
                assert mgr._front.stats["hybrid_launches"] > 0
        finally:
            asyncio.run(mgr.close())
    assert outs[True] == outs[False]
    assert all(m == "semantic" for r in outs[True][3] for m in [r[4]]) and len(outs[True][3]) == 10   # dense hits only


def test_a_bad_request_fails_alone(gpu, long_timeout):
    """One malformed sparse query (duplicate index) inside a coalesced batch: that search fails (the retriever turns it
    into "no sparse hits", retrieval.py:387-389), the others are served."""
    rng = np.random.default_rng(5)
    n, d, V = 5000, 32, 300
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = np.sort(np.argpartition(rng.random((n, V)), 9, axis=1)[:, :10], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 10)).astype(np.float32)
    mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float16", enable_domain=False)
    mgr.add_rows(X, (np.arange(n + 1, dtype=np.int64) * 10, idx, val))
    mgr.finalize()
    good = {"indices": [1, 5, 9], "values": [1.0, 0.5, 0.25]}
    bad = {"indices": [1, 1, 9], "values": [1.0, 0.5, 0.25]}

    async def go():
        calls = [mgr.search(good, "sparse_index", 10) for _ in range(6)] + [mgr.search(bad, "sparse_index", 10)]
        return await asyncio.gather(*calls, return_exceptions=True)

    try:
        outs = asyncio.run(go())
        assert isinstance(outs[-1], Exception)
        assert all(isinstance(o, list) and len(o) == 10 for o in outs[:-1])
        assert all([h["id"] for h in o] == [h["id"] for h in outs[0]] for o in outs[:-1])
    finally:
        asyncio.run(mgr.close())


def test_cancelled_and_timed_out_callers_do_not_hurt_their_batch_mates(gpu, long_timeout):
    """Callers that go away — cancelled tasks, a wait_for that expires — leave cancelled Futures in a round; the others
    of the same launch must still get their (correct) hits and the worker must survive (set_result on a
    cancelled Future raised InvalidStateError into the whole chunk)."""
    rng = np.random.default_rng(9)
    n, d, V, nq = 20000, 64, 400, 48
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = np.sort(np.argpartition(rng.random((n, V)), 9, axis=1)[:, :10], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 10)).astype(np.float32)
    mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float16", enable_domain=False)
    mgr.add_rows(X, (np.arange(n + 1, dtype=np.int64) * 10, idx, val))
    mgr.finalize()
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    SQ = [{"indices": np.sort(rng.choice(V, 8, replace=False)).tolist(), "values": np.abs(rng.standard_normal(8)).tolist()}
          for _ in range(nq)]

    async def sequential():
        return [await mgr.hybrid_search(Q[i], SQ[i], 10, None, (0.7, 0.3)) for i in range(nq)]

    async def with_dropouts():
        tasks = [asyncio.ensure_future(mgr.hybrid_search(Q[i], SQ[i], 10, None, (0.7, 0.3))) for i in range(nq)]
        dense = [asyncio.ensure_future(mgr.search(Q[i], "semantic_index", 10)) for i in range(nq)]
        await asyncio.sleep(0)                      # every task has submitted its request
        for i in range(0, nq, 5):
            tasks[i].cancel()
            dense[i].cancel()
        outs = await asyncio.gather(*tasks, *dense, return_exceptions=True)
        # a second wave whose callers give up after 0 s: wait_for cancels the wrapped Future while the round runs
        gone = await asyncio.gather(*[asyncio.wait_for(mgr.hybrid_search(Q[i], SQ[i], 10, None, (0.7, 0.3)), timeout=0)
                                      for i in range(8)], return_exceptions=True)
        after = await asyncio.gather(*[mgr.hybrid_search(Q[i], SQ[i], 10, None, (0.7, 0.3)) for i in range(nq)])
        return outs[:nq], outs[nq:], gone, after

    def ids(res):
        return [(h["id"], float(s).hex(), m) for h, s, m in res]

    try:
        ref = [ids(r) for r in asyncio.run(sequential())]
        hy, de, gone, after = asyncio.run(with_dropouts())
        for i in range(nq):
            if i % 5 == 0:
                assert isinstance(hy[i], asyncio.CancelledError) and isinstance(de[i], asyncio.CancelledError)
            else:
                assert ids(hy[i]) == ref[i], i
                assert isinstance(de[i], list) and len(de[i]) == 10
        assert all(isinstance(g, (asyncio.TimeoutError, list)) for g in gone)
        assert [ids(r) for r in after] == ref          # the worker is alive and still right
        assert mgr._front.stats.get("worker_failures", 0) == 0
        assert mgr._front._thread.is_alive()
    finally:
        asyncio.run(mgr.close())


def test_requests_with_different_fusion_weights_share_one_round(gpu, long_timeout):
    """A weight_adapter picks the fusion weights per request (reference retrieval.py:251-262): they are an operand of the
    post kernel ([B, 3] doubles), not part of the batch key — 32 concurrent requests with 32 different weight pairs run
    as one launch and return what each returns alone, fused scores bit for bit."""
    rng = np.random.default_rng(21)
    n, d, V, nq = 20000, 64, 400, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = np.sort(np.argpartition(rng.random((n, V)), 9, axis=1)[:, :10], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 10)).astype(np.float32)
    mgr = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float32", enable_domain=False)
    mgr.add_rows(X, (np.arange(n + 1, dtype=np.int64) * 10, idx, val))
    mgr.finalize()
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    SQ = [{"indices": np.sort(rng.choice(V, 8, replace=False)).tolist(), "values": np.abs(rng.standard_normal(8)).tolist()}
          for _ in range(nq)]
    W = [(0.5 + 0.01 * i, 0.5 - 0.013 * i) for i in range(nq)]

    async def alone():
        return [await mgr.hybrid_search(Q[i], SQ[i], 10, None, W[i]) for i in range(nq)]

    async def together():
        return await asyncio.gather(*[mgr.hybrid_search(Q[i], SQ[i], 10, None, W[i]) for i in range(nq)])

    try:
        a = asyncio.run(alone())
        st0 = dict(mgr._front.stats)
        b = asyncio.run(together())
        st = mgr._front.stats
        assert st["hybrid_launches"] - st0["hybrid_launches"] <= 4, (st0, st)     # not one launch per weight pair
        assert len(mgr._front._engines) == 1
        for x, y in zip(a, b):
            assert [(h["id"], float(s).hex(), m) for h, s, m in x] == [(h["id"], float(s).hex(), m) for h, s, m in y]
        # and the weights matter: the host fusion with each request's own weights gives the same fused scores
        for i in (0, 7, 31):
            dl = asyncio.run(mgr.search(Q[i], "semantic_index", 20))
            sl = asyncio.run(mgr.search(SQ[i], "sparse_index", 20))
            fused = {}
            for lst, w in ((dl, W[i][0]), (sl, W[i][1])):
                for rank, h in enumerate(lst, 1):
                    fused[h["id"]] = fused.get(h["id"], 0.0) + (1.0 / (60 + rank)) * w
            top = sorted(fused.items(), key=lambda kv: -kv[1])[:10]
            assert [float(s) for _, s, _ in b[i]] == [v for _, v in top]
    finally:
        asyncio.run(mgr.close())


@pytest.mark.parametrize("device_cache", [256, 0])
def test_concurrent_retrieves_share_encoder_forwards(gpu, long_timeout, device_cache):
    """The query encoder inside retrieve(): 64 concurrent AdvancedRAGPipeline.retrieve() calls with a SentenceEncoder as the
    embedding generator encode their cache misses in ONE forward per round of the front (the reference embeds each request
    alone, indexing.py:601-627), the rows stay on the device (device-resident table) or go once to the host cache, and
    every caller gets what a sequential caller gets with the same embedding."""
    from advanced_rag import AdvancedRAGPipeline, PipelineConfig
    from advanced_rag.embedding_cache import initialize_caches
    from advanced_rag.encoders import EncoderConfig, SentenceEncoder
    rng = np.random.default_rng(17)
    n, d, V, nq = 30000, 384, 1000, 64
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = np.sort(np.argpartition(rng.random((n, V)), 9, axis=1)[:, :10], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * 10)).astype(np.float32)
    enc = SentenceEncoder(EncoderConfig(), device="cuda:0", max_len=64, batch_size=16)
    texts = [f"question number {i} about topic {i % 7} and the matter of item {i * 31 % 101}" for i in range(nq)]

    class Gen:
        """The encoder for the dense side; a fixed sparse query per text (BM25 is not what is under test)."""
        encode_to_device = enc.encode_to_device
        encode_semantic = enc.encode_semantic
        encode_semantic_batch = enc.encode_semantic_batch

        def encode_sparse(self, text):
            r = np.random.default_rng(abs(hash(text)) % (2 ** 32))
            qi = np.sort(r.choice(V, 12, replace=False))
            return {"indices": qi.tolist(), "values": np.abs(r.standard_normal(12)).tolist()}

        def encode_domain(self, text, domain=None):
            return np.zeros(8, np.float32)

    initialize_caches()
    pipe = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False), semantic_dim=d, sparse_dim=V, dtype="float16",
                               enable_domain=False, device_embedding_cache=device_cache)
    mgr = pipe.index_manager
    mgr.add_rows(X, (np.arange(n + 1, dtype=np.int64) * 10, idx, val))
    mgr.finalize()
    mgr.embedding_generator = Gen()
    pipe.retriever.config.enable_learned_ranker = True

    def strip(res):
        return [(r.chunk_id, float(r.score).hex(), r.retrieval_method) for r in res]

    async def concurrent():
        return await asyncio.gather(*[pipe.retrieve(t, context={"retrieval_profile": "default"}) for t in texts])

    async def sequential():
        return [await pipe.retrieve(t, context={"retrieval_profile": "default"}) for t in texts]

    import contextlib, io
    try:
        f0 = getattr(enc, "forwards", 0)
        with contextlib.redirect_stdout(io.StringIO()):
            conc = asyncio.run(concurrent())
        st = mgr._front.stats
        forwards = enc.forwards - f0
        assert st["encoded_texts"] == nq and forwards == st["encode_launches"], (st, forwards)
        assert forwards <= 4, forwards                     # one forward per round of the front, not one per request
        assert all(len(r) == 5 for r, _ in conc)
        with contextlib.redirect_stdout(io.StringIO()):
            seq = asyncio.run(sequential())                # cache hits: the same embeddings, one request at a time
        assert enc.forwards - f0 == forwards               # nothing was encoded again
        assert [strip(r) for r, _ in conc] == [strip(r) for r, _ in seq]
        if device_cache:
            assert mgr.device_cache_stats["misses"] == nq and mgr.device_cache_stats["hits"] >= nq
        # the batched forward gives the embedding a lone forward gives (up to fp16 padding effects)
        lone = enc.encode_to_device([texts[3]])[0]
        from advanced_rag.embedding_cache import EmbeddingCache
        got = (mgr._dev_cache.lookup(EmbeddingCache._materialize_key(texts[3])) if device_cache
               else torch.from_numpy(asyncio.run(mgr._generate_semantic_embedding(texts[3]))).cuda())
        assert torch.allclose(got, lone, atol=3e-3), (got - lone).abs().max()
    finally:
        asyncio.run(pipe.close())
