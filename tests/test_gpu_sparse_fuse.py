"""Sparse TAAT + RRF + cross-shard merge parity (HIP through the C ABI vs the oracle)."""
import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_sparse(rng, n, V, nnz):
    """Reference placeholder shape (indexing.py:647-654): nnz distinct indices, |N(0,1)| values, sorted."""
    indptr = np.arange(n + 1, dtype=np.int64) * nnz
    # nnz distinct indices per row: the nnz smallest of V random keys
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    return indptr, idx, val


def make_queries(rng, B, V, nnz):
    out = []
    for _ in range(B):
        qi = np.sort(rng.choice(V, size=nnz, replace=False)).astype(np.int32)
        out.append((qi, np.abs(rng.standard_normal(nnz)).astype(np.float32)))
    return out


@pytest.mark.parametrize("n,V,nnz,B,k,drop", [(500, 1000, 20, 3, 10, 0.0), (5000, 10000, 100, 8, 40, 0.2),
                                               (9000, 2000, 30, 17, 40, 0.2), (70, 50, 5, 4, 100, 0.5),
                                               (40000, 500, 12, 5, 40, 0.2), (33000, 60, 6, 3, 200, 0.0)])
def test_sparse_matches_oracle(gpu, n, V, nnz, B, k, drop):
    rng = np.random.default_rng(n + V)
    indptr, idx, val = make_sparse(rng, n, V, nnz)
    queries = make_queries(rng, B, V, min(nnz, V))
    h = nat.ShardHandle(0, sparse_dim=V)
    half = n // 2
    h.add_sparse(indptr[:half + 1], idx, val)  # two ragged CSR batches
    h.add_sparse(indptr[half:], idx, val)
    h.finalize()
    assert h.num_sparse_rows == n
    ids, sc = h.search_sparse(queries, k, drop)
    oids, osc = oracle.sparse_search(indptr, idx, val, queries, k, drop)
    assert np.array_equal(ids, oids)
    assert np.array_equal(_bits(sc), _bits(osc))
    h.close()


def test_sparse_edge_cases(gpu):
    rng = np.random.default_rng(11)
    V = 300
    # ragged rows incl. empty ones, an empty query, a query with no match
    rows = [np.sort(rng.choice(V - 10, size=s, replace=False)).astype(np.int32) for s in (0, 1, 5, 0, 40, 3, 0, 12)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    idx = np.concatenate(rows).astype(np.int32)
    val = np.abs(rng.standard_normal(len(idx))).astype(np.float32)
    queries = [(np.zeros(0, np.int32), np.zeros(0, np.float32)),
               (np.array([V - 1, V - 2], np.int32), np.ones(2, np.float32)),
               (rows[4][:7], np.abs(rng.standard_normal(7)).astype(np.float32))]
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(indptr, idx, val)
    h.finalize()
    ids, sc = h.search_sparse(queries, 5, 0.0)
    oids, osc = oracle.sparse_search(indptr, idx, val, queries, 5, 0.0)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sc), _bits(osc))
    assert (ids[0] == -1).all() and (ids[1] == -1).all()
    # row mask
    allow = np.array([1, 1, 0, 1, 0, 1, 1, 1], dtype=bool)
    mask = np.packbits(allow, bitorder="little")
    ids, sc = h.search_sparse(queries, 5, 0.0, mask)
    oids, osc = oracle.sparse_search(indptr, idx, val, queries, 5, 0.0, mask)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sc), _bits(osc))
    with pytest.raises(ValueError):
        h.add_sparse(np.array([0, 2], np.int64), np.array([5, 5], np.int32), np.ones(2, np.float32))  # not ascending
    with pytest.raises(ValueError):
        h.add_sparse(np.array([0, 1], np.int64), np.array([V], np.int32), np.ones(1, np.float32))  # out of range
    h.close()


@pytest.mark.parametrize("na,nb,nc,overlap", [(5, 3, 0, True), (40, 40, 0, True), (40, 40, 20, True), (0, 7, 0, False),
                                               (6, 0, 0, False), (80, 80, 0, True), (1, 1, 1, True)])
def test_rrf_matches_oracle(gpu, na, nb, nc, overlap):
    rng = np.random.default_rng(na * 100 + nb)
    pool = rng.permutation(10 * (na + nb + nc + 1)).astype(np.int64)
    a = pool[:na]
    b = np.concatenate([a[::3], pool[na:]])[:nb] if overlap else pool[na:na + nb]
    b = rng.permutation(b)
    c = rng.permutation(np.concatenate([a[1::4], b[::2], pool[-nc:]]))[:nc] if nc else np.zeros(0, np.int64)
    _, uniq = np.unique(c, return_index=True)
    c = c[np.sort(uniq)]
    h = nat.ShardHandle(8)
    for wa, wb, wc in ((0.7, 0.3, 0.2), (0.5, 0.5, 0.5), (1.0, 0.0, 0.2)):
        gi, gs, gm = h.fuse_rrf(a, b, c, wa, wb, wc, 60)
        oi, os_, om = oracle.rrf(a, b, c, wa, wb, wc, 60)
        assert np.array_equal(gi, oi)
        assert np.array_equal(gs.view(np.uint64), os_.view(np.uint64))  # float64, bit for bit
        assert np.array_equal(gm, om)
    h.close()


def test_merge_topk_and_batched_rrf_dev(gpu):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    G, B, k = 4, 6, 40
    scores = rng.standard_normal((G, B, k)).astype(np.float32)
    scores[:, :, ::5] = 0.25  # ties across shards
    scores = -np.sort(-scores, axis=2)
    ids = rng.permutation(G * B * k).reshape(G, B, k).astype(np.int64)
    ids[1, :, 30:] = -1  # a short shard
    ts, ti = torch.from_numpy(scores).to(dev), torch.from_numpy(ids).to(dev)
    oi = torch.empty((B, k), dtype=torch.int64, device=dev)
    os_ = torch.empty((B, k), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nat.merge_topk_dev(ts.data_ptr(), ti.data_ptr(), G, B, k, k, oi.data_ptr(), os_.data_ptr(), st)
    torch.cuda.synchronize()
    for b in range(B):
        flat_s, flat_i = scores[:, b, :].ravel(), ids[:, b, :].ravel()
        keep = flat_i >= 0
        order = np.lexsort((flat_i[keep], -flat_s[keep]))[:k]
        assert np.array_equal(oi[b].cpu().numpy(), flat_i[keep][order])
        assert np.array_equal(os_[b].cpu().numpy(), flat_s[keep][order])
    # batched RRF on device
    a = np.stack([rng.permutation(1000)[:40] for _ in range(B)]).astype(np.int64)
    bb = np.stack([np.concatenate([a[i, ::2], 2000 + rng.permutation(100)[:20]]) for i in range(B)]).astype(np.int64)
    bb[2, 25:] = -1
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(bb).to(dev)
    top_k = 20
    fo = torch.empty((B, top_k), dtype=torch.int64, device=dev)
    fs = torch.empty((B, top_k), dtype=torch.float64, device=dev)
    fm = torch.empty((B, top_k), dtype=torch.int32, device=dev)
    fn = torch.empty((B,), dtype=torch.int32, device=dev)
    nat.fuse_rrf_dev(ta.data_ptr(), 40, tb.data_ptr(), 40, 0, 0, B, 0.7, 0.3, 0.2, 60, top_k, fo.data_ptr(),
                     fs.data_ptr(), fm.data_ptr(), fn.data_ptr(), st)
    torch.cuda.synchronize()
    for i in range(B):
        oi_, os2, om = oracle.rrf(a[i], bb[i], (), 0.7, 0.3, 0.2, 60)
        n = min(top_k, len(oi_))
        assert int(fn[i]) == n
        assert np.array_equal(fo[i, :n].cpu().numpy(), oi_[:n])
        assert np.array_equal(fs[i, :n].cpu().numpy().view(np.uint64), os2[:n].view(np.uint64))
        assert np.array_equal(fm[i, :n].cpu().numpy(), om[:n])


def test_sparse_ties_and_few_positive_docs(gpu):
    """(a) many identical sparse rows -> ties at the cut -> escalation, lowest ids win; (b) a query that matches
    fewer than k docs returns exactly those, -1 padded, and is still proven exact (nothing outside can be > 0)."""
    import torch
    rng = np.random.default_rng(13)
    V, n = 400, 40000
    idx = np.tile(np.array([3, 50, 77, 200], np.int32), n)
    val = np.tile(np.array([0.5, 1.0, 0.25, 2.0], np.float32), n)
    ptr = np.arange(n + 1, dtype=np.int64) * 4
    idx = idx.copy()
    idx[4 * 123 + 1] = 51                      # one row differs
    rare = np.array([399], np.int32)
    idx[4 * 20000 + 3] = 399                   # exactly one doc holds term 399
    queries = [(np.array([50, 200], np.int32), np.array([1.0, 1.0], np.float32)), (rare, np.ones(1, np.float32))]
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    ids, sc = h.search_sparse(queries, 60, 0.0)
    oids, osc = oracle.sparse_search(ptr, idx, val, queries, 60, 0.0)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sc), _bits(osc))
    assert ids[1].tolist() == [20000] + [-1] * 59
    st = torch.cuda.current_stream().cuda_stream
    from advanced_rag.engine import pack_sparse_queries
    p, i_, v_, mx = pack_sparse_queries(queries, 0.0)
    dp, di_, dv_ = (torch.from_numpy(a).cuda() for a in (p, i_, v_))
    oi = torch.empty((2, 60), dtype=torch.int64, device="cuda")
    os_ = torch.empty((2, 60), dtype=torch.float32, device="cuda")
    fl = torch.ones((2,), dtype=torch.int32, device="cuda")
    h.search_sparse_dev(dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), 2, len(i_), mx, 60, oi.data_ptr(), os_.data_ptr(),
                        fl.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert fl.tolist() == [0, 1]
    h.close()


def test_zipfian_postings_and_long_queries_match_oracle(gpu):
    """SURVEY §7 'sparse skew': postings with a Zipf(1.1) term distribution — some twenty terms occur in EVERY doc
    (runs of 16 384 postings per range = a whole slot table), df = 50 % around term 32 — searched with short
    stop-word-heavy queries, a query of the 40 most frequent terms, and a 1 000-term query (several term chunks per
    range).  200 k docs = 13 ranges.  ids and scores bit-exact, every list proven."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import sparse_block_zipf, zipf_queries, zipf_term_probs
    n, V = 200_000, 10000
    ptr, idx, val = sparse_block_zipf(0, n, seed=99)
    df = np.bincount(idx, minlength=V) / n
    assert df[0] == 1.0 and 0.45 < df[32] < 0.55 and (df >= 0.5).sum() >= 30
    rng = np.random.default_rng(4)
    queries = zipf_queries(rng, 6)
    queries.append((np.arange(40, dtype=np.int32), np.abs(rng.standard_normal(40)).astype(np.float32)))          # the heaviest terms
    big = np.sort(rng.choice(V, 1000, replace=False, p=zipf_term_probs(V) / zipf_term_probs(V).sum())).astype(np.int32)
    queries.append((big, np.abs(rng.standard_normal(1000)).astype(np.float32)))                                  # 4 term chunks per range
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    for drop in (0.0, 0.2):
        ids, sc = h.search_sparse(queries, 40, drop)
        oids, osc = oracle.sparse_search(ptr, idx, val, queries, 40, drop)
        assert np.array_equal(ids, oids), drop
        assert np.array_equal(_bits(sc), _bits(osc))
    # the device form proves every list at the first attempt (no escalation needed for skewed runs)
    torch = pytest.importorskip("torch")
    from advanced_rag.engine import pack_sparse_queries
    p_, i_, v_, mx = pack_sparse_queries(queries, 0.2, V)
    dev = torch.device("cuda:0")
    d_ids = torch.empty((len(queries), 40), dtype=torch.int64, device=dev)
    d_sc = torch.empty((len(queries), 40), dtype=torch.float32, device=dev)
    d_fl = torch.zeros((len(queries),), dtype=torch.int32, device=dev)
    tp, ti, tv = (torch.from_numpy(a).to(dev) for a in (p_, i_, v_))
    h.search_sparse_dev(tp.data_ptr(), ti.data_ptr(), tv.data_ptr(), len(queries), int(i_.shape[0]), int(mx), 40,
                        d_ids.data_ptr(), d_sc.data_ptr(), d_fl.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_fl.min().item() == 1
    assert np.array_equal(d_ids.cpu().numpy(), oids)
    h.close()


def test_dense_run_format_boundaries_and_appends(gpu):
    """The dense form of frequent terms' runs (csrc/sparse.h: one fp16 weight per doc of a range when at least 8 189 of the
    range's 16 384 docs have the term): 50 terms in EVERY doc (more dense runs in one query than a step's list holds: 32),
    terms at exactly 8 188 / 8 189 / 8 192 / 8 193 docs of the first range (last sparse and first dense lengths), zero
    weights and tiny weights among the dense ones, a partial second range in which the same terms are sparse again — and
    the shard grown by appends, so that runs switch form when a flush rebuilds their range.  Every state against the
    oracle, bit for bit."""
    rng = np.random.default_rng(23)
    n, V = 20000, 300
    rows_idx, rows_val = [], []
    special = {60: 8188, 61: 8189, 62: 8192, 63: 8193}
    member = {t: set(rng.choice(16384, c, replace=False).tolist()) for t, c in special.items()}
    for d in range(n):
        terms = list(range(50))                                             # in every doc
        terms += [t for t in special if d in member[t]]                     # counted inside range 0 only
        terms += (100 + rng.choice(200, 6, replace=False)).tolist()         # ordinary sparse terms
        terms = np.unique(np.asarray(terms, dtype=np.int32))
        w = np.abs(rng.standard_normal(terms.shape[0])).astype(np.float32) + 0.01
        if d % 977 == 0:
            w[0] = 0.0                                                      # a stored zero weight in a dense run
        if d % 1009 == 0:
            w[1] = 1e-9                                                     # rounds to the smallest fp16 subnormal
        rows_idx.append(terms)
        rows_val.append(w)
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows_idx])]).astype(np.int64)
    idx, val = np.concatenate(rows_idx), np.concatenate(rows_val)
    queries = [(np.arange(50, dtype=np.int32), np.abs(rng.standard_normal(50)).astype(np.float32) + 0.1),      # 50 dense runs
               (np.array([3, 60, 61, 62, 63, 150], np.int32), np.array([0.5, 1.0, 1.0, 1.0, 1.0, 2.0], np.float32)),
               (np.array([0, 1, 120, 130], np.int32), np.array([1.0, 1.0, 3.0, 3.0], np.float32)),
               (np.arange(0, 300, 3, dtype=np.int32), np.abs(rng.standard_normal(100)).astype(np.float32))]
    h = nat.ShardHandle(0, sparse_dim=V)
    done = 0
    for upto in (5000, 9000, 16384, 16500, n):   # 5 000 / 9 000 docs: "every doc" terms switch from sparse to dense in range 0
        h.add_sparse(ptr[done: upto + 1] - ptr[done], idx[ptr[done]: ptr[upto]], val[ptr[done]: ptr[upto]])
        h.finalize()
        done = upto
        ids, sc = h.search_sparse(queries, 40)
        oids, osc = oracle.sparse_search(ptr[: upto + 1], idx[: ptr[upto]], val[: ptr[upto]], queries, 40)
        assert np.array_equal(ids, oids), upto
        assert np.array_equal(_bits(sc), _bits(osc)), upto
    mask = np.packbits(rng.random(n) < 0.4, bitorder="little")
    ids, sc = h.search_sparse(queries, 40, 0.0, mask)
    oids, osc = oracle.sparse_search(ptr, idx, val, queries, 40, 0.0, mask)
    assert np.array_equal(ids, oids) and np.array_equal(_bits(sc), _bits(osc))
    h.close()


def test_refine_lookup_forms_agree_with_the_oracle(gpu):
    """The canonical sparse refine looks every doc entry up in the query: queries of up to 256 terms as a hash table behind
    the membership filter (csrc/sparse.h SparseLookupHash), longer ones by a lower-bound search in the sorted terms.  Both
    forms, both finishing paths (the fused kernel and the multi-launch chain), at the lengths either side of the table
    sizes (128 | 129 terms: 512 -> 1024 slots; 256 | 257: table -> search) — ids and scores bit for bit.  (A term that
    occurs twice in a query is refused at the boundary: "duplicate query index".)"""
    rng = np.random.default_rng(77)
    n, V, nnz = 30000, 4096, 40
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32) + 0.01
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    queries = []
    for nt in (3, 128, 129, 256, 257, 600):
        qi = np.sort(rng.choice(V, nt, replace=False)).astype(np.int32)
        queries.append((qi, np.abs(rng.standard_normal(nt)).astype(np.float32) + 0.01))
    # terms that collide in the 512-slot table (same upper hash bits): multiples of 2^23 / 0x9E3779B1 do not exist, so take
    # what the hash gives: among 4096 terms many pairs share a slot — a 128-term query has ~14 occupied-slot collisions
    qi = np.sort(rng.choice(V, 128, replace=False)).astype(np.int32)
    queries.append((qi, np.abs(rng.standard_normal(128)).astype(np.float32) + 0.01))
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    try:
        for group in ([queries[0], queries[1], queries[6]], [queries[2], queries[3]], [queries[4], queries[5]], queries):
            want_i, want_s = oracle.sparse_search(ptr, idx, val, group, 40, 0.0)
            for mode in (1, 2):
                nat.debug_option(nat.HR_DEBUG_FINISH_MODE, mode)
                ids, sc = h.search_sparse(group, 40, 0.0)
                assert np.array_equal(ids, want_i), (mode, len(group))
                assert np.array_equal(_bits(sc), _bits(want_s)), (mode, len(group))
        with pytest.raises(ValueError, match="duplicate query index"):
            h.search_sparse([(np.array([7, 7, 9], np.int32), np.ones(3, np.float32))], 40, 0.0)
    finally:
        nat.debug_option(nat.HR_DEBUG_FINISH_MODE, 0)
        h.close()
