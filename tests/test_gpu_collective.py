"""The torchrun form on real shards: two processes (gloo between them; on an 8-GPU node the same code runs over RCCL) each
own a libhbmrag shard on the card, the collection is INGESTED through rank 0 (collective add / finalize), searched and
retrieved from rank 0, saved (every rank writes its own shard file) and resumed rank by rank.  Every answer is checked
against the oracle over the whole corpus: bit-exact ids, scores within the fp16-row tolerance of the other parity tests."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _corpus(n=20011, d=64, V=500, nnz=9, B=5):
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d)).astype(np.float16)
    X[300] = X[15000]                       # an exact tie across the ranks
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    ptr = np.arange(n + 1, dtype=np.int64) * nnz
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[0] = X[300].astype(np.float32)
    SQ = [(np.sort(rng.choice(V, 30, replace=False)).astype(np.int32), np.abs(rng.standard_normal(30)).astype(np.float32))
          for _ in range(B)]
    return X, ptr, idx, val, Q, SQ


def _worker(rank, world, port, ret, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
    import asyncio

    import torch.distributed as dist

    import oracle
    from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig, _native
    from advanced_rag.constants import RetrievalConstants
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, ptr, idx, val, Q, SQ = _corpus()
        n, d, V = X.shape[0], X.shape[1], 500
        _native.load_library()

        def manager():
            m = MilvusIndexManager(semantic_dim=d, sparse_dim=V, connect=False, dtype="float16")
            m._connect()
            return m

        def same(got, want):
            gi, gs = got
            wi, ws = want
            assert np.array_equal(gi, wi), (gi[0, :8], wi[0, :8])
            assert np.allclose(gs, ws, rtol=2e-6, atol=2e-6)

        mgr = manager()
        h = _native.ShardHandle(d, _native.HR_F16, _native.HR_METRIC_COSINE, V, 0)
        mgr.attach_shards([h], rows_of=[np.zeros(0, np.int64)], process_group=True, local_ids=True)
        if rank != 0:
            mgr.serve()
        else:
            cs = mgr._main
            cuts = [0, 7001, 7002, 16000, n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                mgr.add_rows(X[a:b], (ptr[a:b + 1], idx, val), chunk_index=[r % 10 for r in range(a, b)])
            mgr.finalize()
            assert (cs.num_rows, cs.num_sparse_rows, mgr.num_rows) == (n, n, n)
            keep = (np.arange(n) % 10) < 5
            packed = np.packbits(keep, bitorder="little")
            same(cs.search_dense(Q, 40), oracle.dense_search(X, Q, 40, oracle.COSINE))
            same(cs.search_dense(Q, 40, keep), oracle.dense_search(X, Q, 40, oracle.COSINE, packed))
            same(cs.search_sparse(SQ, 40, 0.2), oracle.sparse_search(ptr, idx, val, SQ, 40, 0.2))
            same(cs.search_sparse(SQ, 40, 0.0, keep), oracle.sparse_search(ptr, idx, val, SQ, 40, 0.0, packed))
            assert cs.search_dense(Q[:1], 40)[0][0, :2].tolist() == [300, 15000]

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
            di, _ = oracle.dense_search(X, Q[2:3], 40, oracle.COSINE, packed)
            si, _ = oracle.sparse_search(ptr, idx, val, SQ[2:3], 40, 0.2, packed)
            fi, fs, _ = oracle.rrf(di[0], si[0][si[0] >= 0], (), 0.7, 0.3, 0.2, 60)
            assert [o["id"] for o in out] == [MilvusIndexManager.synthetic_id(int(r)) for r in fi[:20]]
            assert np.allclose([o["score"] for o in out], fs[:20], rtol=1e-12)
            mgr.save_snapshot(tmp)
            maps = cs.row_maps()
            assert sorted(np.concatenate(maps).tolist()) == list(range(n))
            mgr.stop_workers()
        dist.barrier()

        # ---- resume: every rank loads its own shard file and row map
        m2 = manager()
        m2.load_snapshot_rank(tmp, process_group=True, device=0)
        if rank != 0:
            m2.serve()
        else:
            cs2 = m2._main
            assert cs2.num_rows == n and m2.num_rows == n
            same(cs2.search_dense(Q, 40), oracle.dense_search(X, Q, 40, oracle.COSINE))
            same(cs2.search_sparse(SQ, 40, 0.2, keep), oracle.sparse_search(ptr, idx, val, SQ, 40, 0.2, packed))
            m2.embedding_generator = mgr.embedding_generator
            out2 = asyncio.run(HybridRetriever(m2, RetrievalConfig(top_k=20)).retrieve("2", filters=flt, profile_hint="default"))
            assert [o["id"] for o in out2] == [o["id"] for o in out]
            m2.stop_workers()
        ret[rank] = True
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_collective_ingest_search_snapshot_on_two_gpu_processes(gpu, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret, str(tmp_path)), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
