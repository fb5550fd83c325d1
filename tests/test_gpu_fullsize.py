"""BASELINE config 4 at FULL size (10M x 768 fp16) through size-independent properties — the CPU oracle needs
~10 s per query at this size, so full lists are not recomputed; instead:
  * self-query: a stored row used as query comes back first with cosine 1.0 (and its duplicate right after, by id);
  * ordering: scores non-increasing, equal scores in ascending id order, no id twice, flags all proven exact;
  * invariance: the same query gives the same list alone (B=1), inside a batch of 64, and at another batch position;
  * spot oracle: every returned (id, score) is re-scored by the CPU oracle from the regenerated row, bit for bit,
    and no sampled non-returned row beats the k-th score;
  * sharding: two half-corpus shards + the HIP merge kernel reproduce the whole-corpus lists exactly.
"""
import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, D, K, BLK = 10_000_000, 768, 40, 500_000
V, NNZ, SBLK = 10000, 100, 250_000   # the sparse side of config 4: bench.sparse_block's placeholder docs, block by block


def _block(b):
    g = torch.Generator(device="cuda")
    g.manual_seed(9000 + b)
    return torch.randn((BLK, D), device="cuda", generator=g, dtype=torch.float32).to(torch.float16)


def _rows(ids):
    """Regenerate the given rows on the host (block-wise, deterministic)."""
    out = np.empty((len(ids), D), dtype=np.float16)
    ids = np.asarray(ids)
    for b in np.unique(ids // BLK):
        blk = _block(int(b))
        sel = np.nonzero(ids // BLK == b)[0]
        out[sel] = blk[torch.from_numpy(ids[sel] % BLK).cuda()].cpu().numpy()
    return out


def _sparse_block(b):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import sparse_block
    return sparse_block(b, SBLK)


@pytest.fixture(scope="module")
def shards(gpu):
    whole = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, V)   # config 4: dense + sparse collections of one shard
    halves = [nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE) for _ in range(2)]
    whole.reserve(N)
    for i, h in enumerate(halves):
        h.reserve(N // 2)
        h.set_row_offset(i * (N // 2))
    for b in range(N // BLK):
        x = _block(b)
        if b == 3:
            x[17] = _block(0)[5]          # row 3*BLK+17 duplicates row 5: an exact tie far apart
        torch.cuda.synchronize()
        whole.add_dense_dev(x.data_ptr(), BLK)
        halves[b * BLK // (N // 2)].add_dense_dev(x.data_ptr(), BLK)
    for b in range(N // SBLK):   # 10M sparse docs, appended and flushed block by block (incremental posting build)
        whole.add_sparse(*_sparse_block(b))
        if b % 8 == 7:
            whole.finalize()
    for h in [whole] + halves:
        h.finalize()
    assert whole.num_sparse_rows == N
    yield whole, halves
    for h in [whole] + halves:
        h.close()


def test_fullsize_dense_properties(shards):
    whole, halves = shards
    assert whole.num_rows == N
    rng = np.random.default_rng(1)
    B = 64
    Q = rng.standard_normal((B, D)).astype(np.float32)
    probe = [5, 123_456, 9_999_999]
    Q[:3] = _rows(probe).astype(np.float32)
    ids, sc = whole.search_dense(Q, K)

    # self-query + duplicate tie
    assert ids[0, 0] == 5 and ids[0, 1] == 3 * BLK + 17 and sc[0, 0] == sc[0, 1] == 1.0
    assert ids[1, 0] == probe[1] and ids[2, 0] == probe[2] and sc[1, 0] == 1.0
    # ordering / uniqueness
    assert (ids >= 0).all()
    assert (np.diff(sc, axis=1) <= 0).all()
    ties = np.diff(sc, axis=1) == 0
    assert (np.diff(ids, axis=1)[ties] > 0).all()
    assert all(len(set(r)) == K for r in ids.tolist())
    # invariance: alone, and at another position inside another batch
    for j in (0, 7, 63):
        i1, s1 = whole.search_dense(Q[j:j + 1], K)
        assert np.array_equal(i1[0], ids[j]) and np.array_equal(s1[0].view(np.uint32), sc[j].view(np.uint32))
    perm = rng.permutation(B)
    ip, sp_ = whole.search_dense(Q[perm], K)
    assert np.array_equal(ip, ids[perm]) and np.array_equal(sp_.view(np.uint32), sc[perm].view(np.uint32))
    # spot oracle: returned rows re-scored from regenerated data, bit for bit; sampled outsiders do not beat the k-th
    for j in (3, 40):
        rows = _rows(ids[j])
        want = oracle.dense_scores(rows, Q[j], oracle.COSINE)
        assert np.array_equal(want.view(np.uint32), sc[j].view(np.uint32))
        outsiders = rng.integers(0, N, 20_000)
        outsiders = outsiders[~np.isin(outsiders, ids[j])]
        assert oracle.dense_scores(_rows(outsiders), Q[j], oracle.COSINE).max() <= sc[j, -1]
    # device form: flags say every list is provably exact
    dq = torch.from_numpy(Q).cuda()
    di = torch.empty((B, K), dtype=torch.int64, device="cuda")
    ds = torch.empty((B, K), dtype=torch.float32, device="cuda")
    fl = torch.zeros((B,), dtype=torch.int32, device="cuda")
    whole.search_dense_dev(dq.data_ptr(), B, K, di.data_ptr(), ds.data_ptr(), fl.data_ptr(), 0,
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.min()) == 1 and np.array_equal(di.cpu().numpy(), ids)

    # sharding: two halves + merge == whole
    G = 2
    gs = torch.empty((G, B, K), dtype=torch.float32, device="cuda")
    gi = torch.empty((G, B, K), dtype=torch.int64, device="cuda")
    for r, h in enumerate(halves):
        h.search_dense_dev(dq.data_ptr(), B, K, gi[r].data_ptr(), gs[r].data_ptr(), 0, 0,
                           torch.cuda.current_stream().cuda_stream)
    mi = torch.empty((B, K), dtype=torch.int64, device="cuda")
    ms = torch.empty((B, K), dtype=torch.float32, device="cuda")
    nat.merge_topk_dev(gs.data_ptr(), gi.data_ptr(), G, B, K, K, mi.data_ptr(), ms.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(mi.cpu().numpy(), ids) and np.array_equal(ms.cpu().numpy().view(np.uint32), sc.view(np.uint32))


def test_fullsize_batch_kernels_agree(shards):
    """The three dense scans (<= 64 queries: queries in LDS; <= 128: query tile streamed through LDS; <= 256: queries in
    registers, corpus through LDS by DMA) are different candidate generators in front of the same canonical refine:
    at 10M rows they must return identical lists, and every list must be proven exact."""
    whole, _ = shards
    rng = np.random.default_rng(77)
    Q = rng.standard_normal((256, D)).astype(np.float32)
    Q[200] = _rows([5]).astype(np.float32)[0]            # row 5 has an exact duplicate at 3*BLK+17
    i256, s256 = whole.search_dense(Q, K)
    assert i256[200, 0] == 5 and i256[200, 1] == 3 * BLK + 17
    i128, s128 = whole.search_dense(Q[:128], K)
    assert np.array_equal(i128, i256[:128]) and np.array_equal(s128.view(np.uint32), s256[:128].view(np.uint32))
    for j0 in (0, 192):
        i64, s64 = whole.search_dense(Q[j0:j0 + 64], K)
        assert np.array_equal(i64, i256[j0:j0 + 64])
        assert np.array_equal(s64.view(np.uint32), s256[j0:j0 + 64].view(np.uint32))
    dq = torch.from_numpy(Q).cuda()
    di = torch.empty((256, K), dtype=torch.int64, device="cuda")
    ds = torch.empty((256, K), dtype=torch.float32, device="cuda")
    fl = torch.zeros((256,), dtype=torch.int32, device="cuda")
    whole.search_dense_dev(dq.data_ptr(), 256, K, di.data_ptr(), ds.data_ptr(), fl.data_ptr(), 0,
                           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(fl.min()) == 1 and np.array_equal(di.cpu().numpy(), i256)


def test_fullsize_rowmask_monotone(shards):
    """Restricting the corpus can only remove results: masked lists are the unmasked ranking filtered."""
    whole, _ = shards
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((4, D)).astype(np.float32)
    ids, sc = whole.search_dense(Q, 100)
    allow = np.ones(N, dtype=bool)
    allow[ids[:, ::2].ravel()] = False              # knock out every other hit
    mids, msc = whole.search_dense(Q, 40, np.packbits(allow, bitorder="little"))
    for j in range(4):
        kept = [i for i in ids[j].tolist() if allow[i]]
        assert mids[j, :len(kept)].tolist()[:40] == kept[:40]
        assert not (~allow[mids[j]]).any()


def test_large_sparse_properties(gpu):
    """Sparse path at 2M docs (123 doc ranges): spot oracle on returned docs (bit for bit) and on sampled outsiders,
    batch-position invariance, two-shard merge == whole, every list proven exact."""
    n, V, nnz, B, K = 2_000_000, 10000, 50, 48, 40
    rng = np.random.default_rng(77)
    stride = V // nnz
    idx = ((np.arange(nnz, dtype=np.int32) * stride)[None, :] + rng.integers(0, stride, (n, nnz), dtype=np.int32)).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz, dtype=np.float32))
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    queries = [(np.sort(rng.choice(V, 100, replace=False)).astype(np.int32), np.abs(rng.standard_normal(100)).astype(np.float32))
               for _ in range(B)]
    whole = nat.ShardHandle(0, sparse_dim=V)
    whole.add_sparse(ptr, idx, val)
    whole.finalize()
    ids, sc = whole.search_sparse(queries, K, 0.2)
    assert (ids >= 0).all() and (np.diff(sc, axis=1) <= 0).all()
    ties = np.diff(sc, axis=1) == 0
    assert (np.diff(ids, axis=1)[ties] > 0).all()
    for j in (0, 17, B - 1):   # spot oracle
        qi, qv = oracle.drop_query(*queries[j], 0.2)
        rows = ids[j]
        sub_ptr = np.arange(K + 1, dtype=np.int64) * nnz
        sub_idx = np.concatenate([idx[r * nnz:(r + 1) * nnz] for r in rows])
        sub_val = np.concatenate([val[r * nnz:(r + 1) * nnz] for r in rows])
        want = oracle.sparse_scores(sub_ptr, sub_idx, sub_val, qi, qv)
        assert np.array_equal(want.view(np.uint32), sc[j].view(np.uint32))
        out = rng.integers(0, n, 200_000)
        out = out[~np.isin(out, rows)]
        o_ptr = np.arange(len(out) + 1, dtype=np.int64) * nnz
        o_idx = idx.reshape(n, nnz)[out].reshape(-1)
        o_val = val.reshape(n, nnz)[out].reshape(-1)
        assert oracle.sparse_scores(o_ptr, o_idx, o_val, qi, qv).max() <= sc[j, -1]
        i1, s1 = whole.search_sparse([queries[j]], K, 0.2)
        assert np.array_equal(i1[0], ids[j]) and np.array_equal(s1[0].view(np.uint32), sc[j].view(np.uint32))
    # two shards + merge
    from advanced_rag.engine import pack_sparse_queries
    half = n // 2
    p, i_, v_, mx = pack_sparse_queries(queries, 0.2)
    dp, di_, dv_ = (torch.from_numpy(a).cuda() for a in (p, i_, v_))
    gs = torch.empty((2, B, K), dtype=torch.float32, device="cuda")
    gi = torch.empty((2, B, K), dtype=torch.int64, device="cuda")
    fl = torch.zeros((2, B), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    shards = []
    for r in range(2):
        h = nat.ShardHandle(0, sparse_dim=V)
        h.set_row_offset(r * half)
        lo, hi = r * half, (r + 1) * half
        h.add_sparse(ptr[lo:hi + 1] - ptr[lo], idx[ptr[lo]:ptr[hi]], val[ptr[lo]:ptr[hi]])
        h.finalize()
        h.search_sparse_dev(dp.data_ptr(), di_.data_ptr(), dv_.data_ptr(), B, len(i_), mx, K, gi[r].data_ptr(), gs[r].data_ptr(),
                            fl[r].data_ptr(), 0, st)
        shards.append(h)
    mi = torch.empty((B, K), dtype=torch.int64, device="cuda")
    ms = torch.empty((B, K), dtype=torch.float32, device="cuda")
    nat.merge_topk_dev(gs.data_ptr(), gi.data_ptr(), 2, B, K, K, mi.data_ptr(), ms.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(fl.min()) == 1
    assert np.array_equal(mi.cpu().numpy(), ids) and np.array_equal(ms.cpu().numpy().view(np.uint32), sc.view(np.uint32))
    for h in shards + [whole]:
        h.close()


def test_config4_hybrid_leg_at_10m_docs(shards):
    """The sparse / hybrid leg of BASELINE config 4 at FULL size (10M docs x 100 nnz, 611 doc ranges) through the batched
    engine (B = 128: the bench's step): every list proven exact, spot oracle on the returned docs (rows regenerated block
    by block) bit for bit and on sampled outsiders, the fused lists = the oracle's RRF of the device's own lists, and the
    batch answer = the single-query host form at three batch positions."""
    from advanced_rag.engine import EngineConfig, HybridSearchEngine, pack_sparse_queries
    whole, _ = shards
    B = 128
    rng = np.random.default_rng(123)
    Q = rng.standard_normal((B, D)).astype(np.float32)
    SQ = []
    for j in range(B):
        qi = (np.arange(NNZ, dtype=np.int32) * (V // NNZ)) + rng.integers(0, V // NNZ, NNZ, dtype=np.int32)
        SQ.append((qi, np.abs(rng.standard_normal(NNZ)).astype(np.float32)))
    cfg = EngineConfig(top_k=20)
    eng = HybridSearchEngine(whole, cfg)
    out = eng.search(torch.from_numpy(Q).cuda(), eng.upload_sparse(pack_sparse_queries(SQ, 0.2, V)))
    torch.cuda.synchronize()
    assert out["flags"].min().item() == 1
    sid, ssc = out["ids"][1].cpu().numpy(), out["scores"][1].cpu().numpy()
    did = out["ids"][0].cpu().numpy()
    assert (sid >= 0).all() and (np.diff(ssc, axis=1) <= 0).all()
    assert (np.diff(sid, axis=1)[np.diff(ssc, axis=1) == 0] > 0).all()
    blocks = {}
    def doc_rows(ids):
        ptrs, idxs, vals = [0], [], []
        for r in ids:
            b = int(r) // SBLK
            if b not in blocks:
                blocks[b] = _sparse_block(b)
            p, i_, v_ = blocks[b]
            lo, hi = p[int(r) % SBLK], p[int(r) % SBLK + 1]
            idxs.append(i_[lo:hi]); vals.append(v_[lo:hi]); ptrs.append(ptrs[-1] + hi - lo)
        return np.asarray(ptrs, np.int64), np.concatenate(idxs), np.concatenate(vals)
    for j in (0, 63, B - 1):
        qi, qv = oracle.drop_query(*SQ[j], 0.2)
        want = oracle.sparse_scores(*doc_rows(sid[j]), qi, qv)
        assert np.array_equal(want.view(np.uint32), ssc[j].view(np.uint32))
        outsiders = rng.integers(0, 2 * SBLK, 20_000)          # sampled from two blocks: no returned doc may be beaten
        outsiders = outsiders[~np.isin(outsiders, sid[j])]
        assert oracle.sparse_scores(*doc_rows(outsiders), qi, qv).max() <= ssc[j, -1]
        # fusion of the device's own lists
        fi, fs, fm = oracle.rrf(did[j], sid[j], (), cfg.dense_weight, cfg.sparse_weight, 0.2, cfg.rrf_k)
        nf = int(out["fused_n"][j])
        assert np.array_equal(out["fused_ids"][j, :nf].cpu().numpy(), fi[:nf])
        assert np.array_equal(out["fused_scores"][j, :nf].cpu().numpy().view(np.uint64), fs[:nf].view(np.uint64))
        # batch position invariance against the host forms
        hi_, hs_ = whole.search_sparse([SQ[j]], K, 0.2)
        assert np.array_equal(hi_[0], sid[j]) and np.array_equal(hs_[0].view(np.uint32), ssc[j].view(np.uint32))
        di_, _ = whole.search_dense(Q[j:j + 1], K)
        assert np.array_equal(di_[0], did[j])


def test_config5_fullsize_properties(gpu):
    """BASELINE config 5 at FULL size on one GPU: 50M x 1024 fp16 (102.4 GB), B = 256 through the single-pass
    256-query kernel.  Size-independent properties: stored rows used as queries come back first with cosine 1.0,
    lists are ordered with the id tie rule, every list is proven exact, and the 256-query pass gives the same lists as
    the 128-query pass and the single-query pass for the same queries."""
    N5, D5, B5, BLK5 = 50_000_000, 1024, 256, 500_000
    h = nat.ShardHandle(D5, nat.HR_F16, nat.HR_METRIC_COSINE)
    h.reserve(N5)
    keep = {}
    for b in range(N5 // BLK5):
        g = torch.Generator(device="cuda")
        g.manual_seed(5000 + b)
        x = torch.randn((BLK5, D5), device="cuda", generator=g, dtype=torch.float32).to(torch.float16)
        if b in (0, 57, 99):
            keep[b] = x[:3].float().cpu().numpy()
        torch.cuda.synchronize()
        h.add_dense_dev(x.data_ptr(), BLK5)
        del x
    h.finalize()
    assert h.num_rows == N5
    rng = np.random.default_rng(5)
    Q = rng.standard_normal((B5, D5)).astype(np.float32)
    planted = {0: 0, 100: 57 * BLK5 + 1, 255: 99 * BLK5 + 2}       # query slot -> row it copies
    for slot, row in planted.items():
        Q[slot] = keep[row // BLK5][row % BLK5]
    dq = torch.from_numpy(Q).cuda()
    ids = torch.empty((B5, K), dtype=torch.int64, device="cuda")
    sc = torch.empty((B5, K), dtype=torch.float32, device="cuda")
    fl = torch.zeros((B5,), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    h.search_dense_dev(dq.data_ptr(), B5, K, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
    torch.cuda.synchronize()
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    assert int(fl.min()) == 1
    for slot, row in planted.items():
        assert ids[slot, 0] == row and abs(sc[slot, 0] - 1.0) < 1e-6
    assert (ids >= 0).all() and (np.diff(sc, axis=1) <= 0).all()
    assert (np.diff(ids, axis=1)[np.diff(sc, axis=1) == 0] > 0).all()
    # the same queries through the 128-query pass (B = 128) and the single-query pass (B = 1)
    i128, s128 = h.search_dense(Q[64:192], K)
    assert np.array_equal(i128, ids[64:192]) and np.array_equal(s128.view(np.uint32), sc[64:192].view(np.uint32))
    for slot in (0, 100, 255, 31):
        i1, s1 = h.search_dense(Q[slot:slot + 1], K)
        assert np.array_equal(i1[0], ids[slot]) and np.array_equal(s1[0].view(np.uint32), sc[slot].view(np.uint32))
    h.close()
