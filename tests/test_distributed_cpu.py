"""The N>1 path on CPU: two gloo ranks each hold a row shard, build their local top-k' lists
(CPU oracle standing in for the per-shard HIP search), exchange them with the engine's own
pack layout + all-gather, and merge with the offsets/strides the HIP merge kernel is given.
Every rank must end up with the global oracle's lists."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _corpus(n=5003, d=48, V=400, nnz=9, B=6):
    rng = np.random.default_rng(99)
    X = rng.standard_normal((n, d)).astype(np.float16)
    X[100] = X[4000]  # an exact cross-shard tie
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[0] = X[100].astype(np.float32)
    SQ = [(np.sort(rng.choice(V, 30, replace=False)).astype(np.int32), np.abs(rng.standard_normal(30)).astype(np.float32))
          for _ in range(B)]
    return X, ptr, idx, val, Q, SQ


def _merge_numpy(gathered, layout, world, m):
    """numpy restatement of csrc/fuse.h merge_topk_kernel on the gathered byte buffer."""
    sc_off, id_off, sc_stride, id_stride = layout.merge_args(m)
    raw = gathered.numpy().reshape(-1)
    out_ids = np.empty((layout.B, layout.kp), np.int64)
    out_sc = np.empty((layout.B, layout.kp), np.float32)
    for q in range(layout.B):
        ids, scs = [], []
        for w in range(world):
            a = id_off + w * id_stride * 8 + q * layout.kp * 8
            ids.append(raw[a:a + layout.kp * 8].view(np.int64))
            a = sc_off + w * sc_stride * 4 + q * layout.kp * 4
            scs.append(raw[a:a + layout.kp * 4].view(np.float32))
        ids, scs = np.concatenate(ids), np.concatenate(scs)
        keep = ids >= 0
        order = np.lexsort((ids[keep], -scs[keep]))[:layout.kp]
        out_ids[q], out_sc[q] = -1, 0.0
        out_ids[q, :len(order)] = ids[keep][order]
        out_sc[q, :len(order)] = scs[keep][order]
    return out_ids, out_sc


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
    import oracle
    from advanced_rag.engine import ListPack, exchange_lists, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, ptr, idx, val, Q, SQ = _corpus()
        n, kp, B = X.shape[0], 40, Q.shape[0]
        lo, hi = shard_range(n, rank, world, align=64)
        layout = ListPack(2, B, kp)
        pack = torch.zeros(layout.nbytes, dtype=torch.uint8)
        ids, scores = layout.views(pack)
        di, ds = oracle.dense_search(X[lo:hi], Q, kp, oracle.COSINE, row_offset=lo)
        lp = ptr[lo:hi + 1] - ptr[lo]
        si, ss = oracle.sparse_search(lp, idx[ptr[lo]:ptr[hi]], val[ptr[lo]:ptr[hi]], SQ, kp, 0.2, row_offset=lo)
        ids[0], scores[0] = torch.from_numpy(di), torch.from_numpy(ds)
        ids[1], scores[1] = torch.from_numpy(si), torch.from_numpy(ss)
        g = exchange_lists(pack, world, dist)
        assert g.shape == (world, layout.nbytes)
        gd = oracle.dense_search(X, Q, kp, oracle.COSINE)
        gs = oracle.sparse_search(ptr, idx, val, SQ, kp, 0.2)
        for m, (want_i, want_s) in enumerate((gd, gs)):
            got_i, got_s = _merge_numpy(g, layout, world, m)
            assert np.array_equal(got_i, want_i), f"rank {rank} modality {m}"
            assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
        assert set(gd[0][0, :2].tolist()) == {100, 4000}  # the tie pair spans both shards, lower id first
        assert gd[0][0, 0] == 100
        # global BM25 statistics: each rank observes its own documents, one all-reduce makes idf/avgdl global
        from advanced_rag.bm25 import BM25SparseEncoder
        docs = [f"doc {i} token{i % 7} shared words here {'extra ' * (i % 3)}" for i in range(41)]
        dlo, dhi = shard_range(len(docs), rank, world, align=1)
        local = BM25SparseEncoder(sparse_dim=257).fit(docs[dlo:dhi]).all_reduce_stats(dist)
        whole = BM25SparseEncoder(sparse_dim=257).fit(docs)
        assert (local.n_docs, local.total_len) == (whole.n_docs, whole.total_len)
        assert np.array_equal(local.df, whole.df)
        assert local.encode_query("token3 shared missing") == whole.encode_query("token3 shared missing")
        assert local.encode_document(docs[5]) == whole.encode_document(docs[5])
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_exchange_and_merge_matches_global_oracle():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_range_partitions_rows():
    from advanced_rag.engine import shard_range
    for n, w, a in ((10_000_000, 8, 250_000), (10, 4, 1), (1000, 3, 64), (5, 8, 1), (0, 2, 1)):
        spans = [shard_range(n, r, w, a) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert all(lo % a == 0 or lo == n for lo, _ in spans)


def test_pack_sparse_queries_matches_oracle_drop():
    import oracle
    from advanced_rag.engine import pack_sparse_queries
    rng = np.random.default_rng(4)
    qs = [(rng.permutation(500)[:n].astype(np.int32), np.round(np.abs(rng.standard_normal(n)), 1).astype(np.float32))
          for n in (0, 1, 7, 100)]
    ptr, idx, val, mx = pack_sparse_queries(qs, 0.2)
    for b, (qi, qv) in enumerate(qs):
        oi, ov = oracle.drop_query(qi, qv, 0.2)
        assert np.array_equal(idx[ptr[b]:ptr[b + 1]], oi) and np.array_equal(val[ptr[b]:ptr[b + 1]], ov)
    assert mx == 80


class _OracleShard:
    """Stands in for a ShardHandle on a box without a GPU: the oracle over one contiguous row range, with the
    handle's search signatures (global row ids through the row offset, packed local row masks)."""

    def __init__(self, X, ptr, idx, val, lo, hi, sparse_dim):
        import oracle
        self.o, self.X, self.lo, self.hi = oracle, X[lo:hi], lo, hi
        self.ptr, self.idx, self.val = ptr[lo:hi + 1] - ptr[lo], idx[ptr[lo]:ptr[hi]], val[ptr[lo]:ptr[hi]]
        self.device, self.sparse_dim = 0, sparse_dim
        self.num_rows = self.num_sparse_rows = hi - lo

    def search_dense(self, q, k, mask=None):
        return self.o.dense_search(self.X, q, k, self.o.COSINE, mask, row_offset=self.lo)

    def search_sparse(self, queries, k, drop, mask=None):
        return self.o.sparse_search(self.ptr, self.idx, self.val, queries, k, drop, mask, row_offset=self.lo)

    def fuse_rrf(self, a, b, c, wa, wb, wc, rrf_k):
        return self.o.rrf(a, b, c, wa, wb, wc, rrf_k)

    def finalize(self):
        pass

    def close(self):
        pass


def _collective_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import asyncio
    import oracle
    from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
    from advanced_rag.constants import RetrievalConstants
    from advanced_rag.engine import shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, ptr, idx, val, Q, SQ = _corpus()
        n, V = X.shape[0], 400
        lo, hi = shard_range(n, rank, world, align=64)
        mgr = MilvusIndexManager(semantic_dim=X.shape[1], sparse_dim=V, connect=False)
        mgr._native = None  # no library on this box: the shard below is the oracle
        mgr.attach_shards([_OracleShard(X, ptr, idx, val, lo, hi, V)], rows_of=[np.arange(hi - lo)], synthetic_rows=n,
                          process_group=True, first_row=lo)
        assert mgr._main.num_rows == n and mgr._main.n_shards == world
        if rank != 0:
            mgr.serve()
            ret[rank] = True
            return
        cs = mgr._main
        keep = np.random.default_rng(1).random(n) < 0.3
        packed = np.packbits(keep, bitorder="little")
        for B in (1, 6):
            gi, gs = cs.search_dense(Q[:B], 40)
            oi, os_ = oracle.dense_search(X, Q[:B], 40, oracle.COSINE)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
            gi, gs = cs.search_dense(Q[:B], 40, keep)
            oi, os_ = oracle.dense_search(X, Q[:B], 40, oracle.COSINE, packed)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
            gi, gs = cs.search_sparse(SQ[:B], 40, 0.2)
            oi, os_ = oracle.sparse_search(ptr, idx, val, SQ[:B], 40, 0.2)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
            gi, gs = cs.search_sparse(SQ[:B], 40, 0.0, keep)
            oi, os_ = oracle.sparse_search(ptr, idx, val, SQ[:B], 40, 0.0, packed)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
        assert cs.search_dense(Q[:1], 40)[0][0, 0] == 100  # the cross-rank tie: lower global row first

        # retrieve() on rank 0 over the two ranks == the single-process oracle chain
        class Gen:
            def encode_semantic(self, text):
                return Q[int(text)]

            def encode_sparse(self, text):
                qi, qv = SQ[int(text)]
                return {"indices": qi.tolist(), "values": qv.tolist()}

        mgr.embedding_generator = Gen()
        RetrievalConstants.TIMEOUT_SECONDS = 60.0
        for q in range(3):
            from advanced_rag.embedding_cache import initialize_caches
            initialize_caches()
            out = asyncio.run(HybridRetriever(mgr, RetrievalConfig(top_k=20)).retrieve(str(q), profile_hint="default"))
            di, _ = oracle.dense_search(X, Q[q:q + 1], 40, oracle.COSINE)
            si, _ = oracle.sparse_search(ptr, idx, val, SQ[q:q + 1], 40, 0.2)
            fi, fs, _ = oracle.rrf(di[0], si[0][si[0] >= 0], (), 0.7, 0.3, 0.2, 60)
            assert [o["id"] for o in out] == [MilvusIndexManager.synthetic_id(int(r)) for r in fi[:20]]
            assert [o["score"] for o in out] == [float(s) for s in fs[:20]]
        # one broadcast + one gather per retrieve(): the batching front packs the dense and the sparse search of a call
        # into one round; a filter's mask travels once (an extra broadcast in the round that first uses it)
        retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
        mgr._coalescer(mgr.collections["semantic_index"]).window_s = 0.02   # the COUNTS below are about packing, not about
        c0 = cs.n_collectives                                               # how fast this box's event loop submits
        asyncio.run(retr.retrieve("1", profile_hint="default"))
        assert cs.n_collectives - c0 == 2, cs.n_collectives - c0
        flt = {"chunk_index": {"$lt": 5}}
        keep5 = (np.arange(n) % 10) < 5
        c0 = cs.n_collectives
        out = asyncio.run(retr.retrieve("2", filters=flt, profile_hint="default"))
        assert cs.n_collectives - c0 == 3                        # + the mask, once
        c0 = cs.n_collectives
        out2 = asyncio.run(retr.retrieve("2", filters=flt, profile_hint="default"))
        assert cs.n_collectives - c0 == 2 and [o["id"] for o in out2] == [o["id"] for o in out]
        p5 = np.packbits(keep5, bitorder="little")
        di, _ = oracle.dense_search(X, Q[2:3], 40, oracle.COSINE, p5)
        si, _ = oracle.sparse_search(ptr, idx, val, SQ[2:3], 40, 0.2, p5)
        fi, fs, _ = oracle.rrf(di[0], si[0][si[0] >= 0], (), 0.7, 0.3, 0.2, 60)
        assert [o["id"] for o in out] == [MilvusIndexManager.synthetic_id(int(r)) for r in fi[:20]]
        assert all(o["metadata"]["chunk_index"] < 5 for o in out)
        # tombstones ride the same masks
        asyncio.run(mgr.delete_by_filter("semantic_index", "chunk_index == 3"))
        out3 = asyncio.run(retr.retrieve("2", filters=flt, profile_hint="default"))
        pd = np.packbits(keep5 & ((np.arange(n) % 10) != 3), bitorder="little")
        di, _ = oracle.dense_search(X, Q[2:3], 40, oracle.COSINE, pd)
        si, _ = oracle.sparse_search(ptr, idx, val, SQ[2:3], 40, 0.2, pd)
        fi, fs, _ = oracle.rrf(di[0], si[0][si[0] >= 0], (), 0.7, 0.3, 0.2, 60)
        assert [o["id"] for o in out3] == [MilvusIndexManager.synthetic_id(int(r)) for r in fi[:20]]
        # several concurrent callers share rounds
        async def many():
            return await asyncio.gather(*[retr.retrieve(str(q), profile_hint="default") for q in range(6)])
        c0 = cs.n_collectives
        outs = asyncio.run(many())
        assert cs.n_collectives - c0 < 2 * 6
        for q, o in enumerate(outs):
            di, _ = oracle.dense_search(X, Q[q:q + 1], 40, oracle.COSINE, np.packbits((np.arange(n) % 10) != 3, bitorder="little"))
            assert o[0]["id"] in {MilvusIndexManager.synthetic_id(int(r)) for r in di[0]}
        # a round that cannot be packed fails on rank 0 before any collective: the workers are not left waiting
        with pytest.raises(ValueError):
            cs.round(np.zeros((2_000_000 // X.shape[1], X.shape[1]), np.float32), None, 40)
        assert cs.search_dense(Q[:1], 40)[0][0, 0] == 100
        asyncio.run(mgr.close()) if False else None
        mgr.stop_workers()
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_collective_shard_set_serves_retrieve_across_two_ranks():
    """The torchrun form of the sharded collection (CollectiveShardSet behind MilvusIndexManager.attach_shards):
    rank 0 searches / retrieves, rank 1 serves; the per-rank search is the oracle here (no GPU), the protocol —
    header, query and mask broadcasts, the one gather, the merge — is the product's."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_collective_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


class _GrowingOracleShard:
    """A shard that can be FILLED on a box without a GPU: keeps what add_dense / add_sparse hand it (with the handle's
    conventions: fp16 storage, add_sparse reads indices / values at ABSOLUTE indptr positions) and answers searches with
    the oracle in LOCAL row numbers, like a handle whose row offset is 0."""

    def __init__(self, dim, sparse_dim):
        import oracle
        self.o, self.dim, self.sparse_dim, self.device = oracle, dim, sparse_dim, 0
        self.X = np.zeros((0, dim), np.float16)
        self.ptr, self.idx, self.val = np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32)
        self.fail_next_add = False
        self.flushes = 0

    num_rows = property(lambda self: self.X.shape[0])
    num_sparse_rows = property(lambda self: len(self.ptr) - 1)

    def add_dense(self, rows):
        if self.fail_next_add:
            self.fail_next_add = False
            raise MemoryError("this shard is full")
        self.X = np.concatenate([self.X, np.asarray(rows).astype(np.float16)])

    def add_sparse(self, ptr, idx, val):
        ptr = np.asarray(ptr, np.int64)
        self.idx = np.concatenate([self.idx, np.asarray(idx, np.int32)[ptr[0]:ptr[-1]]])
        self.val = np.concatenate([self.val, np.asarray(val, np.float32)[ptr[0]:ptr[-1]]])
        self.ptr = np.concatenate([self.ptr, self.ptr[-1] + (ptr[1:] - ptr[0])])

    def search_dense(self, q, k, mask=None):
        return self.o.dense_search(self.X, q, k, self.o.COSINE, mask)

    def search_sparse(self, queries, k, drop, mask=None):
        return self.o.sparse_search(self.ptr, self.idx, self.val, queries, k, drop, mask)

    def fuse_rrf(self, a, b, c, wa, wb, wc, rrf_k):
        return self.o.rrf(a, b, c, wa, wb, wc, rrf_k)

    def finalize(self):
        self.flushes += 1

    def save(self, path):
        np.savez(path, X=self.X, ptr=self.ptr, idx=self.idx, val=self.val)

    def close(self):
        pass


def _ingest_worker(rank, world, port, ret, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import asyncio
    import oracle
    from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
    from advanced_rag.constants import RetrievalConstants
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, ptr, idx, val, Q, SQ = _corpus()
        n, V = X.shape[0], 400
        shard = _GrowingOracleShard(X.shape[1], V)
        mgr = MilvusIndexManager(semantic_dim=X.shape[1], sparse_dim=V, connect=False)
        mgr._native = None
        mgr.attach_shards([shard], rows_of=[np.zeros(0, np.int64)], process_group=True, local_ids=True)
        cs = mgr._main
        assert cs.num_rows == 0 and cs.n_shards == world
        if rank != 0:
            real_add = shard.add_dense

            def add_dense(rows):
                if rows.shape[0] == 250:        # this rank's block of the 500-row batch below: refused
                    raise MemoryError("this shard is full")
                real_add(rows)
            shard.add_dense = add_dense
            mgr.serve()
            assert shard.flushes >= 1
            ret[rank] = (shard.num_rows, shard.num_sparse_rows)
            return

        # ---- rank 0 fills the collection; every batch is cut into one block per rank
        cuts = [0, 2001, 2002, 4100]            # uneven batches, one of a single row (rank 1's block of it is empty)
        for a, b in zip(cuts[:-1], cuts[1:]):
            mgr.add_rows(X[a:b], (ptr[a:b + 1], idx, val), chunk_index=[r % 10 for r in range(a, b)])
        assert (cs.num_rows, cs.num_sparse_rows, mgr.num_rows) == (4100, 4100, 4100)
        mgr.finalize()
        m = cuts[-1]
        keep = (np.arange(m) % 10) < 5
        packed = np.packbits(keep, bitorder="little")

        def check(m, keep=None, packed=None, holes=None):
            live = None
            if holes is not None:               # rows a failing rank never stored: the oracle masks them out
                alive = np.ones(m, bool)
                alive[holes] = False
                live = np.packbits(alive if keep is None else alive & keep, bitorder="little")
            elif packed is not None:
                live = packed
            gi, gs = cs.search_dense(Q, 40, keep)
            oi, os_ = oracle.dense_search(X[:m], Q, 40, oracle.COSINE, live)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
            gi, gs = cs.search_sparse(SQ, 40, 0.2, keep)
            oi, os_ = oracle.sparse_search(ptr[:m + 1], idx, val, SQ, 40, 0.2, live)
            assert np.array_equal(gi, oi) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))

        check(m)
        check(m, keep, packed)
        maps = cs.row_maps()
        assert len(maps) == world and sorted(np.concatenate(maps).tolist()) == list(range(m))
        assert not np.array_equal(maps[0], np.arange(len(maps[0])))      # blocks of several batches: not one contiguous range

        # retrieve() with a filter over the ingested collection == the single-process oracle chain
        class Gen:
            def encode_semantic(self, text):
                return Q[int(text)]

            def encode_sparse(self, text):
                qi, qv = SQ[int(text)]
                return {"indices": qi.tolist(), "values": qv.tolist()}

        mgr.embedding_generator = Gen()
        RetrievalConstants.TIMEOUT_SECONDS = 60.0
        retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
        flt = {"chunk_index": {"$lt": 5}}
        out = asyncio.run(retr.retrieve("2", filters=flt, profile_hint="default"))
        di, _ = oracle.dense_search(X[:m], Q[2:3], 40, oracle.COSINE, packed)
        si, _ = oracle.sparse_search(ptr[:m + 1], idx, val, SQ[2:3], 40, 0.2, packed)
        fi, fs, _ = oracle.rrf(di[0], si[0][si[0] >= 0], (), 0.7, 0.3, 0.2, 60)
        assert [o["id"] for o in out] == [MilvusIndexManager.synthetic_id(int(r)) for r in fi[:20]]
        assert [o["score"] for o in out] == [float(s) for s in fs[:20]]

        # snapshot: every rank writes its own shard file, rank 0 the payload columns and all the row maps
        mgr.save_snapshot(tmp)
        assert all(os.path.exists(os.path.join(tmp, f"main.{r}.hbmrag.npz")) or os.path.exists(os.path.join(tmp, f"main.{r}.hbmrag"))
                   for r in range(world))
        with np.load(os.path.join(tmp, "payload.npz")) as z:
            assert int(z["n_shards"]) == world
            assert sorted(np.concatenate([z[f"rows_main_{r}"] for r in range(world)]).tolist()) == list(range(m))
        with np.load(os.path.join(tmp, "main.1.hbmrag.npz")) as z1:     # rank 1's file holds exactly rank 1's rows
            assert np.array_equal(z1["X"], X[maps[1]])

        # a rank that refuses its block: rank 0 hears of it, the numbering stays common (the block is a hole)
        a, b = m, 4600
        with pytest.raises(RuntimeError, match="another rank"):
            mgr.add_rows(X[a:b], (ptr[a:b + 1], idx, val), chunk_index=[r % 10 for r in range(a, b)])
        assert cs.num_rows == b
        from advanced_rag.engine import shard_range
        lo, hi = shard_range(b - a, 1, world)
        holes = np.arange(a + lo, a + hi)
        assert mgr.num_rows == b            # the payload columns follow the numbering (PartialAppend), holes included
        # filters made before the append no longer fit: the next use sends a fresh mask (the cache was cleared everywhere)
        m = b
        keep = np.ones(m, bool)
        check(m, keep, holes=holes)
        # and the collection keeps growing afterwards
        a, b = m, n
        mgr.add_rows(X[a:b], (ptr[a:b + 1], idx, val), chunk_index=[r % 10 for r in range(a, b)])
        mgr.finalize()
        assert mgr.num_rows == cs.num_rows == n
        check(n, np.ones(n, bool), holes=holes)
        assert cs.search_dense(Q[:1], 40)[0][0, 0] == 100                # the cross-rank tie: lower global row first
        mgr.stop_workers()
        ret[rank] = (shard.num_rows, shard.num_sparse_rows)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_collective_ingest_fills_two_ranks_and_matches_the_single_shard_answer(tmp_path):
    """Ingest THROUGH the torchrun form: rank 0 calls add_rows / finalize / save_snapshot, rank 1 sits in serve(); every
    batch is broadcast and cut into one block per rank (general row maps, not one contiguous range per rank).  Searches,
    filtered searches and retrieve() over the result equal the oracle over the whole corpus; a rank that fails an append
    is reported on rank 0 and leaves a hole, not a disagreement about row numbers."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ingest_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    assert ret[0][0] + ret[1][0] == 5003 - 250          # rank 1 refused its 250-row block of the 500-row batch
    assert ret[0][1] + ret[1][1] == 5003 - 250
