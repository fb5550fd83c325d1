"""The oracle against the reference's own outputs (tests/golden/*.json, generated from the
imported reference by tests/golden/gen_golden.py) and against itself (C vs numpy)."""
import json
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _intern(*lists):
    table = {}
    out = []
    for lst in lists:
        out.append([table.setdefault(x, len(table)) for x in lst])
    return out, {v: k for k, v in table.items()}


@pytest.mark.parametrize("impl", [oracle.rrf, oracle.rrf_py])
def test_rrf_matches_reference_fuse_results(impl):
    cases = gold("g1_fuse.json")
    assert len(cases) >= 20
    for c in cases:
        (a, b, d), back = _intern(c["semantic"], c["sparse"], c["domain"])
        ids, scores, methods = impl(a, b, d, c["dense_weight"], c["sparse_weight"], 0.2, 60)
        assert [back[int(i)] for i in ids] == c["ids"], c["label"]
        assert [float(s).hex() for s in scores] == c["scores"], c["label"]  # float64, bit for bit
        names = ("semantic", "sparse", "domain")
        assert [sorted(n for bit, n in enumerate(names) if (int(m) >> bit) & 1) for m in methods] == c["methods"]


def test_rrf_known_values_from_survey():
    ids, scores, _ = oracle.rrf([0, 1, 2, 3, 4], [3, 7, 0])
    assert ids.tolist() == [0, 3, 1, 2, 4, 7]
    assert scores[0] == 0.016237314597970336 and scores[1] == 0.015855532786885247
    assert 0.7 / 61 == 0.011475409836065573


def test_dense_c_and_numpy_agree_bit_for_bit():
    rng = np.random.default_rng(0)
    for dt in (np.float16, np.float32):
        X = rng.standard_normal((777, 130)).astype(np.float32).astype(dt)
        X[5] = 0
        q = rng.standard_normal(130).astype(np.float32)
        for metric in (oracle.IP, oracle.COSINE):
            a, b = oracle.dense_scores(X, q, metric), oracle.dense_scores_np(X, q, metric)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert oracle.dense_scores(X, q, oracle.COSINE)[5] == 0.0
        assert np.all(oracle.dense_scores(X, np.zeros(130, np.float32), oracle.COSINE) == 0.0)


def test_dense_close_to_fp32_blas_path():
    """Canonical scores vs the numpy fp32 'reference path' (BASELINE.md §3): within 1e-4, same ids when gaps allow."""
    rng = np.random.default_rng(1)
    X = rng.standard_normal((5000, 384)).astype(np.float32)
    Q = rng.standard_normal((4, 384)).astype(np.float32)
    Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
    ids, sc = oracle.cpu_dense_topk(Xn, Q, 40)
    oids, osc = oracle.dense_search(X, Q, 40, oracle.COSINE)
    assert np.max(np.abs(sc - osc)) < 1e-4
    assert np.array_equal(ids, oids)


def test_topk_tie_rule_mask_and_padding():
    s = np.array([0.5, 0.9, 0.5, 0.9, -1.0, 0.0], dtype=np.float32)
    ids, sc = oracle.topk(s, 4)
    assert ids.tolist() == [1, 3, 0, 2]
    ids, _ = oracle.topk(s, 8, only_positive=True)
    assert ids.tolist() == [1, 3, 0, 2, -1, -1, -1, -1]
    mask = np.packbits(np.array([1, 0, 1, 1, 1, 1], bool), bitorder="little")
    ids, _ = oracle.topk(s, 3, mask, row_offset=100)
    assert ids.tolist() == [103, 100, 102]


def test_drop_query_semantics():
    i, v = oracle.drop_query([5, 1, 9, 3, 7], [0.5, 0.1, 0.3, 0.1, 0.9], 0.4)
    assert i.tolist() == [5, 7, 9] and np.allclose(v, [0.5, 0.9, 0.3])
    i, v = oracle.drop_query([4, 2], [1.0, 1.0], 0.5)  # equal magnitudes: the later entry goes first
    assert i.tolist() == [4]
    i, v = oracle.drop_query([], [], 0.2)
    assert len(i) == 0


def test_sparse_scores_vs_scipy():
    sp = pytest.importorskip("scipy.sparse")
    rng = np.random.default_rng(2)
    n, V, nnz = 400, 300, 12
    idx = np.stack([np.sort(rng.choice(V, nnz, replace=False)) for _ in range(n)]).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    qi = np.sort(rng.choice(V, 20, replace=False)).astype(np.int32)
    qv = np.abs(rng.standard_normal(20)).astype(np.float32)
    got = oracle.sparse_scores(ptr, idx, val, qi, qv)
    qd = np.zeros(V, np.float64)
    qd[qi] = qv
    want = sp.csr_matrix((val.astype(np.float64), idx, ptr), shape=(n, V)) @ qd
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


def test_half_conversion_is_numpy_astype():
    rng = np.random.default_rng(3)
    v = (rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 6, 4000)).astype(np.float32)
    v[:6] = [0.0, -0.0, 65504.0, 65520.0, 1e-8, 6.1e-5]
    assert np.array_equal(oracle.float_to_half_bits(v), v.astype(np.float16).view(np.uint16))


def test_oracle_search_chain_reproduces_the_references_retrieve_on_config1():
    """oracle.dense_search / oracle.sparse_search(drop 0.2) -> oracle.rrf on g5's inputs gives the ids
    (bit-exact), the float64 fused scores (bit-equal: RRF depends only on ranks) and the method tags that the
    REFERENCE's HybridRetriever.retrieve produced on BASELINE config 1 (reference retrieval.py:249-339,
    :421-491 over an exact numpy FLAT collection; generated by tests/golden/gen_golden.py:159-219).  This ties
    the oracle's dense/sparse ranking (k-ordered fp64, tie = lower row, drop rule) to reference-run output."""
    import g5_data
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs()
    assert len(g["runs"]) == 16
    kp, top_k = 40, 20
    di, _ = oracle.dense_search(X, Q, kp, oracle.COSINE)
    si, _ = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    names = ("semantic", "sparse", "domain")
    for run in g["runs"]:
        q = run["query"]
        sparse_list = si[q][si[q] >= 0] if run["with_sparse"] else ()
        ids, scores, methods = oracle.rrf(di[q], sparse_list, (), 0.7, 0.3, 0.2, 60)
        ids, scores, methods = ids[:top_k], scores[:top_k], methods[:top_k]
        assert [g5_data.row_id(int(r)) for r in ids] == run["ids"], (run["with_sparse"], q)
        assert [float(s).hex() for s in scores] == run["scores"]
        assert [sorted(n for bit, n in enumerate(names) if (int(m) >> bit) & 1) for m in methods] == run["methods"]


def test_oracle_search_chain_reproduces_the_references_retrieve_at_config2_size():
    """G13 (tests/golden/gen_golden_g13.py): the reference's HybridRetriever.retrieve over an exact numpy FLAT collection of
    100 000 x 384 fp32 rows + 100-term sparse rows (BASELINE config 2's size: seven 16 384-doc ranges, thousands of
    candidate groups), 8 queries, dense-only and hybrid -> the oracle chain gives the same ids, float64 fused scores (bit
    for bit) and method tags."""
    import g5_data
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs(g5_data.load_g13())
    assert (g["N"], g["D"], len(g["runs"])) == (100_000, 384, 16)
    kp, top_k = 40, 20
    di, _ = oracle.dense_search(X, Q, kp, oracle.COSINE)
    si, _ = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    names = ("semantic", "sparse", "domain")
    for run in g["runs"]:
        q = run["query"]
        sparse_list = si[q][si[q] >= 0] if run["with_sparse"] else ()
        ids, scores, methods = oracle.rrf(di[q], sparse_list, (), 0.7, 0.3, 0.2, 60)
        assert [g5_data.row_id(int(r)) for r in ids[:top_k]] == run["ids"], (run["with_sparse"], q)
        assert [float(s).hex() for s in scores[:top_k]] == run["scores"]
        assert [sorted(n for bit, n in enumerate(names) if (int(m) >> bit) & 1) for m in methods[:top_k]] == run["methods"]


# --------------------------------------------------------------------------- G11 / G12 (round 4)
NAMES3 = ("semantic", "sparse", "domain")


def _names(mask):
    return sorted(n for bit, n in enumerate(NAMES3) if (int(mask) >> bit) & 1)


def test_oracle_mmr_matches_reference_fuse_results_with_mmr():
    """G11a: `_fuse_results` of the imported reference with enable_mmr=True (retrieval.py:488-516) at lambda 0.5 / 0.7 /
    0.8 (+ 0 and 1, empty contents, lists shorter than top_k, a domain list): oracle.rrf -> oracle.mmr gives the same
    order, the same float64 fused scores and method tags."""
    cases = gold("g11_mmr.json")["fuse"]
    assert len(cases) >= 15 and {c["mmr_lambda"] for c in cases} >= {0.5, 0.7, 0.8}
    for c in cases:
        (a, b, d), back = _intern(c["semantic"], c["sparse"], c["domain"])
        ids, scores, methods = oracle.rrf(a, b, d, c["dense_weight"], c["sparse_weight"], 0.2, 60)
        names = [back[int(i)] for i in ids]
        sel = oracle.mmr(names, [float(s) for s in scores], [c["content"][n] for n in names], c["top_k"], c["mmr_lambda"])
        assert [names[i] for i in sel] == c["ids"], c["label"]
        assert [float(scores[i]).hex() for i in sel] == c["scores"], c["label"]
        assert [_names(methods[i]) for i in sel] == c["methods"], c["label"]


def _oracle_fused(X, ptr, idx, val, Q, SQ, q, kp, with_sparse, wa=0.7, wb=0.3):
    di, _ = oracle.dense_search(X, Q[q:q + 1], kp, oracle.COSINE)
    if with_sparse:
        si, _ = oracle.sparse_search(ptr, idx, val, [SQ[q]], kp, 0.2)
        sl = si[0][si[0] >= 0]
    else:
        sl = ()
    return di[0], sl, oracle.rrf(di[0], sl, (), wa, wb, 0.2, 60)


def test_oracle_chain_reproduces_the_references_retrieve_under_the_mmr_profiles():
    """G11b: the reference's HybridRetriever.retrieve on config 1 with profile_hint = troubleshooting / analysis /
    summary (top_k 30 / 30 / 40, searches with k' = 60 / 60 / 80, MMR at 0.5 / 0.8 / off)."""
    import g5_data
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs()
    runs = gold("g11_mmr.json")["retrieve"]
    assert len(runs) == 30 and {r["profile_hint"] for r in runs} == {"troubleshooting", "analysis", "summary"}
    for run in runs:
        top_k = run["top_k"]
        assert run["search_top_k"] == [2 * top_k] and run["profile"] == run["profile_hint"]
        _, _, (ids, scores, methods) = _oracle_fused(X, ptr, idx, val, Q, SQ, run["query"], 2 * top_k, run["with_sparse"])
        if run["enable_mmr"]:
            sel = oracle.mmr(ids, [float(s) for s in scores], [g5_data.mmr_content(int(r)) for r in ids], top_k, run["mmr_lambda"])
        else:
            sel = list(range(len(ids)))
        sel = sel[:top_k]
        assert [g5_data.row_id(int(ids[i])) for i in sel] == run["ids"], (run["profile_hint"], run["query"])
        assert [float(scores[i]).hex() for i in sel] == run["scores"]
        assert [_names(methods[i]) for i in sel] == run["methods"]


def test_oracle_chain_reproduces_the_references_pipeline_retrieve():
    """G12: the reference's AdvancedRAGPipeline(connect_to_milvus=False).retrieve() with the learned ranker switched on
    (pipeline.py:217-309, retrieval.py:544-563, ranker.py:109-125) over fake collections holding config 1: chunk ids in
    order, float64 scores bit for bit, the payload's method tag, and how many come back for PipelineConfig.rerank_top_k =
    5 / 7 / 12 (the retriever's own rerank_top_k stays 5 and is never consulted)."""
    import g5_data
    g, X, (ptr, idx, val), Q, SQ = g5_data.inputs()
    cases = gold("g12_pipeline.json")["cases"]
    assert len(cases) >= 15
    for c in cases:
        q = int(c["query"].rsplit("q", 1)[1])
        top_k = c["top_k"]
        assert c["search_limits"] == [2 * top_k] and c["retriever_rerank_top_k"] == 5
        dl, sl, (ids, scores, methods) = _oracle_fused(X, ptr, idx, val, Q, SQ, q, 2 * top_k, c["with_sparse"])
        ids, scores, methods = ids[:top_k], scores[:top_k], methods[:top_k]
        if c["enable_reranking"]:
            order, new = oracle.learned_rank(scores, [bin(int(m)).count("1") for m in methods], c["pipeline_rerank_top_k"])
        else:
            order = list(range(min(len(ids), c["pipeline_rerank_top_k"])))
            new = [float(scores[i]) for i in order]
        assert c["n"] == len(order) == min(c["pipeline_rerank_top_k"], len(ids))
        assert [g5_data.row_id(int(ids[i])) for i in order] == c["chunk_ids"], c["label"]
        assert [float(s).hex() for s in new] == c["scores"], c["label"]
        dense_set = set(int(r) for r in dl)
        assert ["semantic" if int(ids[i]) in dense_set else "sparse" for i in order] == c["retrieval_methods"]
        assert c["contents"] == [g5_data.mmr_content(int(ids[i])) for i in order]
        assert c["profiles"] == ["default"] * len(order)
