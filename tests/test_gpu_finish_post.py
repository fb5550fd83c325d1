"""Round-3 kernels: the fused finishing kernel (finish.h) against the multi-launch chain, the LDS merge of sorted
per-shard lists and the one-launch post-exchange kernel (merge + RRF + rerank) against the separate entry points,
the prep stream / CU-masked streams of the pipelined engine, and the restartable sparse build."""
import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat
from advanced_rag.engine import (EngineConfig, HybridSearchEngine, ListPack, PipelinedSearchEngine, pack_sparse_queries)
from test_gpu_engine import corpus, oracle_pipeline

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture
def finish_mode():
    """Force a finishing path for the duration of a test (process-wide hook), back to automatic afterwards."""
    yield lambda m: nat.debug_option(nat.HR_DEBUG_FINISH_MODE, m)
    nat.debug_option(nat.HR_DEBUG_FINISH_MODE, 0)


def _lists(G, B, k, rng, ties=True, short=True, sorted_ids=False):
    scores = rng.standard_normal((G, B, k)).astype(np.float32)
    if ties:
        scores[:, :, ::5] = 0.25  # the same score in every list
    scores = -np.sort(-scores, axis=2)
    ids = rng.permutation(G * B * k).reshape(G, B, k).astype(np.int64)
    if sorted_ids:  # what the search kernels write: (score desc, id asc)
        for g in range(G):
            for b in range(B):
                o = np.lexsort((ids[g, b], -scores[g, b]))
                ids[g, b], scores[g, b] = ids[g, b][o], scores[g, b][o]
    if short and G > 1:
        ids[1, :, k - k // 4:] = -1  # a short shard
        ids[G - 1, 0, :] = -1        # an empty one for query 0
    return scores, ids


def _merge_numpy(scores, ids, k_out):
    G, B, k = scores.shape
    oi = np.full((B, k_out), -1, np.int64)
    os_ = np.zeros((B, k_out), np.float32)
    for b in range(B):
        fs, fi = scores[:, b, :].ravel(), ids[:, b, :].ravel()
        keep = fi >= 0
        order = np.lexsort((fi[keep], -fs[keep]))[:k_out]
        oi[b, :len(order)] = fi[keep][order]
        os_[b, :len(order)] = fs[keep][order]
    return oi, os_


@pytest.mark.parametrize("G,k,k_out", [(1, 40, 40), (2, 40, 40), (8, 40, 40), (8, 200, 200), (8, 40, 20), (3, 7, 40),
                                        (64, 200, 200)])   # 64 x 200 does not fit LDS: the global form
def test_merge_of_sorted_lists_matches_lexsort(gpu, G, k, k_out):
    rng = np.random.default_rng(G * 1000 + k)
    B = 5
    scores, ids = _lists(G, B, k, rng)
    ts, ti = torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda()
    oi = torch.full((B, k_out), -7, dtype=torch.int64, device="cuda")
    os_ = torch.full((B, k_out), -7.0, dtype=torch.float32, device="cuda")
    nat.merge_topk_dev(ts.data_ptr(), ti.data_ptr(), G, B, k, k_out, oi.data_ptr(), os_.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wi, ws = _merge_numpy(scores, ids, k_out)
    assert np.array_equal(oi.cpu().numpy(), wi)
    assert np.array_equal(os_.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_merge_of_duplicated_lists_is_a_permutation(gpu):
    """The same list fed W times (bench --simulate-ranks without the id shift): equal (score, id) pairs are ordered by
    list number, every output slot is written exactly once."""
    rng = np.random.default_rng(9)
    B, k, G = 3, 40, 8
    scores, ids = _lists(1, B, k, rng, ties=False, short=False)
    scores, ids = np.repeat(scores, G, axis=0), np.repeat(ids, G, axis=0)
    ts, ti = torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda()
    oi = torch.full((B, k), -7, dtype=torch.int64, device="cuda")
    os_ = torch.full((B, k), -7.0, dtype=torch.float32, device="cuda")
    nat.merge_topk_dev(ts.data_ptr(), ti.data_ptr(), G, B, k, k, oi.data_ptr(), os_.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for b in range(B):
        assert np.array_equal(oi[b].cpu().numpy(), np.repeat(ids[0, b, :k // G], G))


@pytest.mark.parametrize("W,top_k,use_domain", [(1, 20, False), (2, 20, False), (8, 20, True), (8, 100, False), (1, 20, True)])
def test_post_lists_equals_separate_launches(gpu, W, top_k, use_domain):
    """hr_post_lists_dev (one launch) against hr_merge_topk_dev -> hr_fuse_rrf_dev -> hr_rerank_linear_dev on the
    gathered-buffer layout of the engine: merged lists, fused lists (float64 bits), rerank output, aggregated flags."""
    rng = np.random.default_rng(W * 100 + top_k)
    B, kp = 6, 2 * top_k
    n_mod = 3 if use_domain else 2
    lay = ListPack(n_mod, B, kp)
    g = torch.zeros((W, lay.nbytes), dtype=torch.uint8, device="cuda")
    flags_np = (rng.random((W, n_mod, B)) > 0.2).astype(np.int32)
    for r in range(W):
        ids_v, sc_v = lay.views(g[r])
        for m in range(n_mod):
            kk = top_k if (use_domain and m == n_mod - 1) else kp
            sc, ids = _lists(1, B, kk, rng, short=False, sorted_ids=True)
            ids = ids % 500 + 1000 * r          # overlap between the modalities, distinct across the ranks
            for b in range(B):                   # unique ids inside a list
                _, first = np.unique(ids[0, b], return_index=True)
                dup = np.setdiff1d(np.arange(kk), first)
                ids[0, b, dup] = 600 + 1000 * r + np.arange(len(dup))
            full_i = np.full((B, kp), -1, np.int64)
            full_s = np.zeros((B, kp), np.float32)
            full_i[:, :kk], full_s[:, :kk] = ids[0], sc[0]
            ids_v[m].copy_(torch.from_numpy(full_i))
            sc_v[m].copy_(torch.from_numpy(full_s))
        lay.flags_view(g[r]).copy_(torch.from_numpy(flags_np[r]))
    st = torch.cuda.current_stream().cuda_stream
    k_fuse = [kp, kp, top_k]
    dev = "cuda"
    # ---- reference: the separate entry points
    want_m_ids, want_m_sc = [], []
    for m in range(n_mod):
        if W > 1:
            sc_off, id_off, sc_stride, id_stride = lay.merge_args(m)
            mi = torch.empty((B, k_fuse[m if m < 2 else 2]), dtype=torch.int64, device=dev)
            ms = torch.empty((B, k_fuse[m if m < 2 else 2]), dtype=torch.float32, device=dev)
            nat.merge_topk_dev(g.data_ptr() + sc_off, g.data_ptr() + id_off, W, B, kp, mi.shape[1], mi.data_ptr(),
                               ms.data_ptr(), st, score_stride=sc_stride, id_stride=id_stride)
        else:
            ids_v, sc_v = lay.views(g[0])
            kk = k_fuse[m if m < 2 else 2]
            mi, ms = ids_v[m][:, :kk].contiguous(), sc_v[m][:, :kk].contiguous()
        want_m_ids.append(mi)
        want_m_sc.append(ms)
    def outs():
        return (torch.empty((B, top_k), dtype=torch.int64, device=dev), torch.empty((B, top_k), dtype=torch.float64, device=dev),
                torch.empty((B, top_k), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
                torch.empty((B, 5), dtype=torch.int64, device=dev), torch.empty((B, 5), dtype=torch.float64, device=dev),
                torch.empty((B, 5), dtype=torch.float64, device=dev))
    w_fi, w_fs, w_fm, w_fn, w_ri, w_rs, w_ro = outs()
    dom = want_m_ids[2] if use_domain else None
    nat.fuse_rrf_dev(want_m_ids[0].data_ptr(), kp, want_m_ids[1].data_ptr(), kp, dom.data_ptr() if use_domain else 0,
                     top_k if use_domain else 0, B, 0.7, 0.3, 0.2, 60, top_k, w_fi.data_ptr(), w_fs.data_ptr(),
                     w_fm.data_ptr(), w_fn.data_ptr(), st)
    nat.rerank_linear_dev(w_fi.data_ptr(), w_fs.data_ptr(), w_fm.data_ptr(), w_fn.data_ptr(), B, top_k, 1.0, 0.1, 0.0, 5,
                          w_ri.data_ptr(), w_rs.data_ptr(), w_ro.data_ptr(), st)
    # ---- one launch
    g_fi, g_fs, g_fm, g_fn, g_ri, g_rs, g_ro = outs()
    a = nat.PostArgs()
    got_m = []
    for slot in range(n_mod):
        m = slot
        a.k_fuse[slot] = k_fuse[slot]
        if W > 1:
            sc_off, id_off, sc_stride, id_stride = lay.merge_args(m)
            a.ids[slot], a.scores[slot], a.k_in[slot] = g.data_ptr() + id_off, g.data_ptr() + sc_off, kp
            a.id_stride, a.score_stride = id_stride, sc_stride
            mi = torch.full((B, k_fuse[slot]), -9, dtype=torch.int64, device=dev)
            ms = torch.full((B, k_fuse[slot]), -9.0, dtype=torch.float32, device=dev)
            a.merged_ids[slot], a.merged_scores[slot] = mi.data_ptr(), ms.data_ptr()
            got_m.append((mi, ms))
        else:
            a.ids[slot], a.k_in[slot] = want_m_ids[slot].data_ptr(), k_fuse[slot]
    a.n_lists, a.rrf_k, a.top_k = W, 60, top_k
    a.w[0], a.w[1], a.w[2] = 0.7, 0.3, 0.2
    a.fused_ids, a.fused_scores, a.fused_methods, a.fused_n = g_fi.data_ptr(), g_fs.data_ptr(), g_fm.data_ptr(), g_fn.data_ptr()
    a.rerank, a.base_w, a.method_bonus, a.recency_w, a.k_out = 1, 1.0, 0.1, 0.0, 5
    a.rr_ids, a.rr_scores, a.rr_orig = g_ri.data_ptr(), g_rs.data_ptr(), g_ro.data_ptr()
    agg = torch.full((n_mod, B), -1, dtype=torch.int32, device=dev)
    if W > 1:
        a.flags = g.data_ptr() + lay.id_bytes + lay.score_bytes
        a.flag_stride, a.n_flag_rows, a.agg_flags = lay.nbytes // 4, n_mod * B, agg.data_ptr()
    nat.post_lists_dev(a, B, st)
    torch.cuda.synchronize()
    for (mi, ms), wi, ws in zip(got_m, want_m_ids, want_m_sc):
        assert torch.equal(mi, wi) and torch.equal(ms.view(torch.int32), ws.view(torch.int32))
    assert torch.equal(g_fn, w_fn)
    for b in range(B):
        n = int(w_fn[b])
        assert torch.equal(g_fi[b, :n], w_fi[b, :n]) and torch.equal(g_fm[b, :n], w_fm[b, :n])
        assert torch.equal(g_fs[b, :n].view(torch.int64), w_fs[b, :n].view(torch.int64))
    assert torch.equal(g_ri, w_ri) and torch.equal(g_rs.view(torch.int64), w_rs.view(torch.int64))
    assert torch.equal(g_ro.view(torch.int64), w_ro.view(torch.int64))
    if W > 1:
        assert np.array_equal(agg.cpu().numpy(), flags_np.min(axis=0))


def _search_all(h, Q, SQ, kp, mask=None):
    """Dense, sparse and hybrid device forms of one batch -> dict of numpy arrays."""
    B = Q.shape[0]
    dq = torch.from_numpy(Q).cuda()
    p, i_, v_, mx = pack_sparse_queries(SQ, 0.2)
    dp, di_, dv_ = torch.from_numpy(p).cuda(), torch.from_numpy(i_).cuda(), torch.from_numpy(v_).cuda()
    dm = torch.from_numpy(mask).cuda() if mask is not None else None
    mp = dm.data_ptr() if dm is not None else 0
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    ids = torch.empty((B, kp), dtype=torch.int64, device="cuda")
    sc = torch.empty((B, kp), dtype=torch.float32, device="cuda")
    fl = torch.empty((B,), dtype=torch.int32, device="cuda")
    h.search_dense_dev(dq.data_ptr(), B, kp, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), mp, st)
    torch.cuda.synchronize()
    out["d"] = (ids.cpu().numpy().copy(), sc.cpu().numpy().copy(), fl.cpu().numpy().copy())
    h.search_sparse_dev(dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), B, len(i_), mx, kp, ids.data_ptr(), sc.data_ptr(),
                        fl.data_ptr(), mp, st)
    torch.cuda.synchronize()
    out["s"] = (ids.cpu().numpy().copy(), sc.cpu().numpy().copy(), fl.cpu().numpy().copy())
    ids2 = torch.empty((2, B, kp), dtype=torch.int64, device="cuda")
    sc2 = torch.empty((2, B, kp), dtype=torch.float32, device="cuda")
    fl2 = torch.empty((2, B), dtype=torch.int32, device="cuda")
    h.hybrid_scan_dev(dq.data_ptr(), dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), B, len(i_), mx, kp, 0, st, mp)
    h.hybrid_finish_dev(dq.data_ptr(), dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), B, mx, kp, 0, ids2.data_ptr(),
                        sc2.data_ptr(), fl2.data_ptr(), st, mp)
    torch.cuda.synchronize()
    out["h"] = (ids2.cpu().numpy().copy(), sc2.cpu().numpy().copy(), fl2.cpu().numpy().copy())
    return out


@pytest.mark.parametrize("n,d,V,nnz,B,kp,dtype,group_rows", [
    (20000, 128, 1000, 12, 9, 40, "f16", None),       # 16-row groups, two-level selection, a batch below the fused threshold
    (20000, 128, 1000, 12, 70, 40, "f16", None),      # above it
    (5000, 96, 300, 8, 4, 40, "f32", None),           # fp32 shard
    (9000, 64, 500, 10, 6, 10, "f16", 64),            # 64-row groups (the layout of shards above 3M rows)
    (700, 64, 100, 5, 3, 40, "f16", None),            # fewer groups than candidates: every group is a candidate
    (30000, 64, 600, 10, 5, 200, "f16", None),        # k' = 200: C = 304 > 256 -> the fused kernel does not apply, both modes take the chain
])
def test_fused_finish_is_bit_identical_to_the_chain(gpu, finish_mode, n, d, V, nnz, B, kp, dtype, group_rows):
    nat.debug_option(nat.HR_DEBUG_GROUP_ROWS, group_rows or 0)   # read when the handle is created
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=n + B)
    X[11] = X[n - 3]                      # a tie that straddles candidate groups
    store = nat.HR_F16 if dtype == "f16" else nat.HR_F32
    h = nat.ShardHandle(d, store, nat.HR_METRIC_COSINE, V)
    nat.debug_option(nat.HR_DEBUG_GROUP_ROWS, 0)
    h.add_dense(X if dtype == "f16" else X.astype(np.float32))
    h.add_sparse(ptr, idx, val)
    h.finalize()
    rng = np.random.default_rng(1)
    mask = np.packbits(rng.random(n) < 0.6, bitorder="little")
    for m in (None, mask):
        finish_mode(1)
        chain = _search_all(h, Q, SQ, kp, m)
        finish_mode(2)
        fused = _search_all(h, Q, SQ, kp, m)
        for key in chain:
            for a, b in zip(chain[key], fused[key]):
                assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), key
    # and both equal the oracle (unmasked)
    finish_mode(2)
    got = _search_all(h, Q, SQ, kp)
    Xo = X if dtype == "f16" else X.astype(np.float32)
    odi, ods = oracle.dense_search(Xo, Q, kp, oracle.COSINE)
    osi, oss = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
    proven = got["h"][2][0] == 1
    assert np.array_equal(got["h"][0][0][proven], odi[proven]) and np.array_equal(got["h"][1][0][proven].view(np.uint32), ods[proven].view(np.uint32))
    proven = got["h"][2][1] == 1
    assert np.array_equal(got["h"][0][1][proven], osi[proven]) and np.array_equal(got["h"][1][1][proven].view(np.uint32), oss[proven].view(np.uint32))
    h.close()


@pytest.mark.parametrize("light_cus,prep", [(0, True), (0, False), (32, True)])
def test_pipelined_engine_prep_stream_and_cu_masks(gpu, finish_mode, light_cus, prep):
    """Query prep on its own stream, and finishing / scan streams confined to disjoint compute units: same results as
    the sequential engine (and, through it, as the oracle) — including a 300-query batch that takes two scan passes."""
    n, d, V, nnz = 30000, 128, 1000, 10
    X, ptr, idx, val, _, _ = corpus(n, d, V, nnz, 4, seed=33)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=20)
    seq = HybridSearchEngine(h, cfg)
    pipe = PipelinedSearchEngine(h, cfg, depth=3, light_cus=light_cus, prep_stream=prep)
    rng = np.random.default_rng(4)
    batches = []
    for B in (16, 300, 70, 16, 128):
        Q = rng.standard_normal((B, d)).astype(np.float32)
        SQ = [(np.sort(rng.choice(V, 30, replace=False)).astype(np.int32), np.abs(rng.standard_normal(30)).astype(np.float32))
              for _ in range(B)]
        batches.append((torch.from_numpy(Q).cuda(), seq.upload_sparse(pack_sparse_queries(SQ, 0.2)), Q, SQ))
    keys = ("ids", "scores", "flags", "fused_ids", "fused_scores", "fused_methods", "rr_ids", "rr_scores")
    got = []
    for q, sq, _, _ in batches:
        o = pipe.submit(q, sq)
        with torch.cuda.stream(pipe.light):
            got.append({k: o[k].clone() for k in keys})
    pipe.synchronize()
    for (q, sq, Q, SQ), g_ in zip(batches, got):
        o = seq.search(q, sq)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(o[k], g_[k]), k
    # one batch against the oracle itself
    q, sq, Q, SQ = batches[2]
    (di, ds), (si, ss), _, _ = oracle_pipeline(X, ptr, idx, val, Q, SQ, cfg)
    assert np.array_equal(got[2]["ids"][0].cpu().numpy(), di) and np.array_equal(got[2]["ids"][1].cpu().numpy(), si)
    pipe.close()
    h.close()


def test_simulated_ranks_run_the_merge_on_the_finishing_stream(gpu):
    """bench.py --simulate-ranks: W copies of the local lists (ids shifted per copy) are merged per modality; copy 0
    carries the unshifted ids and wins every tie, so the merged lists equal the local ones."""
    n, d, V, nnz, B = 20000, 64, 500, 8, 64
    X, ptr, idx, val, Q, SQ = corpus(n, d, V, nnz, B, seed=8)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    cfg = EngineConfig(top_k=20)
    pipe = PipelinedSearchEngine(h, cfg, depth=2, simulate_ranks=8)
    sq = pipe.upload_sparse(pack_sparse_queries(SQ, 0.2))
    o = pipe.submit(torch.from_numpy(Q).cuda(), sq)
    pipe.synchronize()
    kp = 40
    # W copies of every entry with the same score: the merged top-k' holds the best k'/W entries, each W times
    # (ids shifted by copy << 40), lowest copy first
    loc_i, loc_s = o["ids"].cpu().numpy(), o["scores"].cpu().numpy()
    m_i, m_s = o["list_ids"].cpu().numpy(), o["list_scores"].cpu().numpy()
    for m in range(2):
        for b in range(B):
            live = loc_i[m, b] >= 0
            want_i = (loc_i[m, b][live][:, None] + (np.arange(8, dtype=np.int64) << 40)[None, :]).ravel()[:kp]
            want_s = np.repeat(loc_s[m, b][live], 8)[:kp]
            nw = len(want_i)
            assert np.array_equal(m_i[m, b, :nw], want_i) and np.array_equal(m_s[m, b, :nw], want_s)
    assert int(o["agg_flags"].min()) == int(o["flags"].min())
    pipe.close()
    h.close()


def test_sparse_build_restarts_after_a_failed_flush(gpu):
    """A flush that fails after the rows have reached the device CSR (in-place staging offsets, cleared
    staging vectors) must be completed by the next one: no lost rows, no double-counted entries; appends made in between
    are picked up too."""
    rng = np.random.default_rng(12)
    V, nnz, n1, n2, n3 = 400, 9, 20000, 3000, 777
    def rows(n):
        idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
        return np.arange(n + 1, dtype=np.int64) * nnz, idx, np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    parts = [rows(n1), rows(n2), rows(n3)]
    h = nat.ShardHandle(0, sparse_dim=V)
    h.add_sparse(*parts[0])
    h.finalize()
    h.add_sparse(*parts[1])
    h.debug_option(nat.HR_DEBUG_FAIL_NEXT_BUILD, 1)
    with pytest.raises(nat.HbmRagError, match="injected failure"):
        h.finalize()
    with pytest.raises((nat.HbmRagError, ValueError)):   # not searchable in between
        h.search_sparse([(np.array([1], np.int32), np.array([1.0], np.float32))], 5)
    h.add_sparse(*parts[2])                              # more rows before the retry
    h.finalize()
    ptr = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p[0]) for p in parts]))]).astype(np.int64)
    idx = np.concatenate([p[1] for p in parts])
    val = np.concatenate([p[2] for p in parts])
    assert h.num_sparse_rows == n1 + n2 + n3
    SQ = [(np.sort(rng.choice(V, 25, replace=False)).astype(np.int32), np.abs(rng.standard_normal(25)).astype(np.float32))
          for _ in range(6)]
    gi, gs = h.search_sparse(SQ, 40, 0.2)
    oi, os_ = oracle.sparse_search(ptr, idx, val, SQ, 40, 0.2)
    assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
    h.close()
