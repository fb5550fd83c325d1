"""Ingest, resume and concurrency of the shard store (reference indexing.py:377-431 inserts and flushes
per index_chunks call): incremental sparse appends rebuild only the tail ranges, the host keeps no copy of the corpus,
snapshots are validated on load, `*_dev` searches that share a stream are serialised."""
import os
import struct
import threading
import time

import numpy as np
import pytest

import oracle
from advanced_rag import _native as nat

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _sparse(rng, n, V, nnz):
    idx = np.sort(np.argpartition(rng.random((n, V)), nnz - 1, axis=1)[:, :nnz], axis=1).astype(np.int32).reshape(-1)
    val = np.abs(rng.standard_normal(n * nnz)).astype(np.float32)
    return np.arange(n + 1, dtype=np.int64) * nnz, idx, val


def test_fifty_ragged_appends_stay_bit_exact(gpu):
    """50 appends of 1 .. 3 000 docs (several crossing the 16 384-doc range boundary), a flush after every one, and
    a search against the oracle on the prefix after every fifth: the tail-range rebuild must leave exactly the
    index a from-scratch build gives."""
    rng = np.random.default_rng(17)
    V, nnz = 700, 9
    sizes = [int(x) for x in rng.integers(1, 3000, size=50)]
    sizes[3], sizes[10], sizes[11] = 1, 16384 - sum(sizes[:10]) % 16384, 5  # land exactly on a boundary, then step over
    n = sum(sizes)
    ptr, idx, val = _sparse(rng, n, V, nnz)
    queries = [(np.sort(rng.choice(V, 25, replace=False)).astype(np.int32), np.abs(rng.standard_normal(25)).astype(np.float32))
               for _ in range(4)]
    h = nat.ShardHandle(0, sparse_dim=V)
    lo = 0
    for i, sz in enumerate(sizes):
        h.add_sparse(ptr[lo:lo + sz + 1], idx, val)
        lo += sz
        h.finalize()
        assert h.num_sparse_rows == lo
        if i % 5 == 4 or i == len(sizes) - 1:
            ids, sc = h.search_sparse(queries, 30, 0.2)
            oi, os_ = oracle.sparse_search(ptr[:lo + 1], idx, val, queries, 30, 0.2)
            assert np.array_equal(ids, oi), i
            assert np.array_equal(_bits(sc), _bits(os_))
    fresh = nat.ShardHandle(0, sparse_dim=V)
    fresh.add_sparse(ptr, idx, val)
    fresh.finalize()
    a, b = h.search_sparse(queries, 50, 0.0), fresh.search_sparse(queries, 50, 0.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1]))
    h.close()
    fresh.close()


def test_small_append_costs_one_range_and_host_keeps_no_corpus(gpu):
    psutil = pytest.importorskip("psutil")
    rng = np.random.default_rng(5)
    V, nnz, blk = 10000, 100, 250_000
    h = nat.ShardHandle(0, sparse_dim=V)
    proc = psutil.Process()
    rss0 = proc.memory_info().rss
    t_build = 0.0
    for b in range(8):  # 2M docs, 200M entries: 1.6 GB of CSR + 0.8 GB of postings on the device
        ptr = np.arange(blk + 1, dtype=np.int64) * nnz
        idx = ((np.arange(nnz, dtype=np.int32) * (V // nnz))[None, :] +
               rng.integers(0, V // nnz, size=(blk, nnz), dtype=np.int32)).reshape(-1)
        val = np.abs(rng.standard_normal(blk * nnz, dtype=np.float32))
        h.add_sparse(ptr, idx, val)
        t0 = time.perf_counter()
        h.finalize()
        t_build += time.perf_counter() - t0
        del ptr, idx, val
    rss1 = proc.memory_info().rss
    assert rss1 - rss0 < 600e6, f"host RSS grew by {(rss1 - rss0) / 1e6:.0f} MB for a 1.6 GB CSR: the corpus must live on the device"
    ptr, idx, val = _sparse(rng, 1000, V, nnz)
    h.add_sparse(ptr, idx, val)
    t0 = time.perf_counter()
    h.finalize()
    t_small = time.perf_counter() - t0
    assert h.num_sparse_rows == 8 * blk + 1000
    print(f"2M-doc build in 8 flushes: {t_build * 1e3:.0f} ms; flush after a 1 000-doc append: {t_small * 1e3:.1f} ms")
    assert t_small < 0.1 * t_build and t_small < 0.25
    # the appended docs are searchable: a query that is one of them finds it first
    q = [(idx[:nnz], val[:nnz])]
    ids, _ = h.search_sparse(q, 5, 0.0)
    assert ids[0, 0] == 8 * blk
    h.close()


def test_snapshot_is_validated_on_load(gpu, tmp_path):
    rng = np.random.default_rng(3)
    n, d, V = 3000, 64, 500
    X = rng.standard_normal((n, d)).astype(np.float16)
    ptr, idx, val = _sparse(rng, n, V, 7)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    h.add_dense(X)
    h.add_sparse(ptr, idx, val)
    h.finalize()
    path = str(tmp_path / "s.hbmrag")
    h.save(path)
    assert not os.path.exists(path + ".tmp")
    good = open(path, "rb").read()
    q = rng.standard_normal((2, d)).astype(np.float32)
    back = nat.ShardHandle.load(path, d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
    a, b = h.search_dense(q, 10), back.search_dense(q, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1]))
    sq = [(idx[:7], val[:7])]
    a, b = h.search_sparse(sq, 10, 0.0), back.search_sparse(sq, 10, 0.0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1]))
    back.close()
    # what the caller expects is checked against what the file holds
    for args in ((d * 2, nat.HR_F16, nat.HR_METRIC_COSINE, V), (d, nat.HR_F32, nat.HR_METRIC_COSINE, V),
                 (d, nat.HR_F16, nat.HR_METRIC_IP, V), (d, nat.HR_F16, nat.HR_METRIC_COSINE, V + 1)):
        with pytest.raises(ValueError):
            nat.ShardHandle.load(path, *args)

    def refused(blob, name):
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        with pytest.raises(ValueError):
            nat.ShardHandle.load(p, d, nat.HR_F16, nat.HR_METRIC_COSINE, V)

    refused(good[:len(good) - 100], "truncated")                      # shorter than the header says
    refused(good + b"\0" * 8, "padded")
    refused(b"NOTASNAP" + good[8:], "magic")
    sparse_at = len(good) - (n + 1) * 8 - len(idx) * 8                # [indptr][idx][val] close the file
    idx_at = sparse_at + (n + 1) * 8
    bad = bytearray(good)
    bad[idx_at + 40:idx_at + 44] = struct.pack("<i", V + 123)         # an index outside [0, sparse_dim)
    refused(bytes(bad), "index_out_of_range")
    bad = bytearray(good)
    bad[idx_at + 4:idx_at + 8] = bad[idx_at:idx_at + 4]               # a duplicate (not strictly ascending) index
    refused(bytes(bad), "not_ascending")
    bad = bytearray(good)
    bad[sparse_at + 16:sparse_at + 24] = struct.pack("<q", 10 ** 12)  # a row pointer beyond the entries
    refused(bytes(bad), "row_pointer")
    bad = bytearray(good)
    val_at = idx_at + len(idx) * 4
    bad[val_at:val_at + 4] = struct.pack("<f", float("nan"))
    refused(bytes(bad), "nan_weight")
    bad = bytearray(good)
    bad[48:56] = struct.pack("<q", 1 << 62)                           # absurd n_rows in the header
    refused(bytes(bad), "header")
    h.close()


def test_dev_searches_sharing_a_stream_are_serialised(gpu):
    """The reference's service admits 64 concurrent retrieve() calls; with the device-resident query cache they all
    reach hr_search_dense_dev from worker threads on ONE stream.  Every thread must get its own query's answer."""
    rng = np.random.default_rng(8)
    n, d, k, T, reps = 20000, 128, 20, 16, 12
    X = rng.standard_normal((n, d)).astype(np.float16)
    Q = rng.standard_normal((T, d)).astype(np.float32)
    h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE)
    h.add_dense(X)
    h.finalize()
    want_i, want_s = h.search_dense(Q, k)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    dq = [torch.from_numpy(Q[t:t + 1]).to(dev) for t in range(T)]
    errors = []
    start = threading.Barrier(T)

    def worker(t):
        try:
            ids = torch.empty((1, k), dtype=torch.int64, device=dev)
            sc = torch.empty((1, k), dtype=torch.float32, device=dev)
            fl = torch.zeros((1,), dtype=torch.int32, device=dev)
            start.wait()
            for _ in range(reps):
                h.search_dense_dev(dq[t].data_ptr(), 1, k, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, stream.cuda_stream)
                stream.synchronize()
                if not (np.array_equal(ids.cpu().numpy()[0], want_i[t]) and np.array_equal(_bits(sc.cpu().numpy()[0]), _bits(want_s[t]))):
                    errors.append(t)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    h.close()


def test_device_to_device_ingest_equals_host_ingest(gpu):
    """index_chunks with an encoder that offers encode_to_device: the dense (and domain) rows go from the encoder's
    output tensor into the shard without a host hop (hr_add_dense_raw_dev).  The shard must be the one the host path
    builds — same rows, same fp16 rounding — so searches return identical lists, and both equal the oracle on the
    encoder's own output."""
    import asyncio
    from advanced_rag import AdvancedRAGPipeline, BM25SparseEncoder, PipelineConfig
    from advanced_rag.embedding_cache import initialize_caches
    from advanced_rag.encoders import EncoderConfig, SentenceEncoder

    docs = [{"id": f"d{i}", "text": " ".join(f"Sentence {j} of document {i} mentions topic{(i * j) % 11} and item{j % 5}." for j in range(12)),
             "metadata": {"source": "unit"}} for i in range(40)]
    bm25 = BM25SparseEncoder(sparse_dim=512).fit(d["text"] for d in docs)
    enc = SentenceEncoder(EncoderConfig(hidden=64, layers=2, heads=4, intermediate=128), device="cuda:0", sparse_encoder=bm25,
                          domain_dim=96, batch_size=16)

    class HostHop:
        encode_semantic = staticmethod(enc.encode_semantic)
        encode_semantic_batch = staticmethod(enc.encode_semantic_batch)
        encode_domain = staticmethod(enc.encode_domain)
        encode_sparse = staticmethod(enc.encode_sparse)
        encode_sparse_query = staticmethod(enc.encode_sparse_query)

    mgrs, calls = {}, []
    for name, gen in (("device", enc), ("host", HostHop())):
        initialize_caches()
        p = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False), semantic_dim=64, sparse_dim=512,
                                domain_dim=96, dtype="float16")
        p.index_manager.embedding_generator = gen
        calls = []
        for part in (docs[:25], docs[25:]):  # two calls: an append behind a flushed batch
            rep = asyncio.run(p.ingest_documents(part, domain="manuals"))
            s = rep["indexing_summary"]
            assert not s["errors"] and s["indexed_semantic"] == s["total_chunks"] == s["indexed_sparse"] == s["indexed_domain"]
            assert set(s["timing_ms"]) == {"encode", "append", "flush"}
            calls.append(s["total_chunks"])
        mgrs[name] = p
    md, mh = mgrs["device"].index_manager, mgrs["host"].index_manager
    assert md._cols["id"] == mh._cols["id"] and md.num_rows == mh.num_rows == sum(calls) >= 40
    texts = md._cols["content"]
    # the encoder's own output for the same call pattern (a forward pass is only bit-reproducible for the same batch
    # composition): semantic rows are encoded per ingest call on both paths; domain rows per call on the device path
    # and one text at a time on the host path
    per_call = [texts[:calls[0]], texts[calls[0]:]]
    X = np.concatenate([np.stack(enc.encode_semantic_batch(t)) for t in per_call]).astype(np.float16)
    Xd_dev = np.concatenate([enc.encode_domain_to_device(t, "manuals").cpu().numpy() for t in per_call]).astype(np.float16)
    Xd_host = np.stack([enc.encode_domain(t, "manuals") for t in texts]).astype(np.float16)
    rng = np.random.default_rng(2)
    for coll, refs, dim in (("semantic_index", (X, X), 64), ("domain_index", (Xd_dev, Xd_host), 96)):
        Q = rng.standard_normal((9, dim)).astype(np.float32)
        Q[0] = refs[0][3].astype(np.float32)
        for mgr, ref in zip((md, mh), refs):
            got = mgr.collections[coll].handle.search_dense(Q, 30)
            oi, os_ = oracle.dense_search(ref, Q, 30, oracle.COSINE)
            assert np.array_equal(got[0], oi) and np.array_equal(got[1].view(np.uint32), os_.view(np.uint32)), coll
    # sparse rows travelled as one CSR per call on both paths
    sq = [(np.asarray(bm25.encode_query(texts[5])["indices"], np.int32), np.asarray(bm25.encode_query(texts[5])["values"], np.float32))]
    assert all(np.array_equal(a, b) for a, b in zip(md._main.first.search_sparse(sq, 20, 0.2), mh._main.first.search_sparse(sq, 20, 0.2)))
    for p in mgrs.values():
        asyncio.run(p.close())


def test_bm25_payloads_on_the_device_equal_the_host_encoder(gpu):
    """hr_bm25_encode_dev (csrc/text.h) against BM25SparseEncoder.encode_document — the host restatement of the same
    arithmetic and the fallback of the batch form — bit for bit: indices, float32 weight bits, row lengths.  Mixed case,
    digits, underscores, punctuation runs, tokens at both ends of a document, an empty document, a single very long
    token, a document of one byte; documents the kernel must hand back (a non-ASCII byte; longer than 65 535 bytes) come
    out identical through the fallback; vocabularies of 257 / 10 000 / 65 536 slots on the device and 70 000 on the host."""
    from advanced_rag import BM25SparseEncoder
    rng = np.random.default_rng(12)
    words = [f"w{i}" for i in range(4000)] + ["Alpha", "BETA", "gamma_9", "_x_", "42", "a", "Z"]
    seps = [" ", ", ", ". ", "\n", " - ", "!!", "\t", "(", ")", "'s "]

    def doc(n):
        return "".join(str(rng.choice(words)) + str(rng.choice(seps)) for _ in range(n))

    texts = [doc(int(n)) for n in rng.integers(1, 700, size=200)]
    texts += ["", "x", "x" * 5000, "...---...", "end", "Start middle END", "under_score and 123 and CamelCase",
              "naïve café déjà vu", "plain then ünïcode", doc(20000)]
    assert len(texts[-1].encode()) > 65535
    for dim in (257, 10000, 65536, 70000):
        enc = BM25SparseEncoder(sparse_dim=dim).fit(texts[:150])
        ptr, idx, val = enc.encode_documents_csr(texts, device="cuda:0")
        assert ptr.shape == (len(texts) + 1,) and ptr[-1] == idx.shape[0] == val.shape[0]
        for i, t in enumerate(texts):
            want = enc.encode_document(t)
            assert idx[ptr[i]:ptr[i + 1]].tolist() == want["indices"], (dim, i, t[:40])
            assert np.array_equal(_bits(val[ptr[i]:ptr[i + 1]]), _bits(np.asarray(want["values"], np.float32))), (dim, i)
    # the flags themselves, straight from the entry point
    enc = BM25SparseEncoder(sparse_dim=1000).fit(texts[:50])
    raw = [t.encode() for t in ("ascii only", "café", "y" * 70000, "a b c d e f g h")]
    off = np.zeros(len(raw) + 1, np.int64)
    np.cumsum([len(r) for r in raw], out=off[1:])
    d_text = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).cuda()
    d_off = torch.from_numpy(off).cuda()
    cap = 4
    d_idx = torch.empty((4, cap), dtype=torch.int32, device="cuda")
    d_val = torch.empty((4, cap), dtype=torch.float32, device="cuda")
    d_nnz = torch.empty(4, dtype=torch.int32, device="cuda")
    d_fl = torch.empty(4, dtype=torch.int32, device="cuda")
    nat.bm25_encode_dev(d_text.data_ptr(), d_off.data_ptr(), 4, 1000, enc.k1, enc.b, enc.avgdl, cap, d_idx.data_ptr(),
                        d_val.data_ptr(), d_nnz.data_ptr(), d_fl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_fl.tolist() == [0, 1, 2, 2] and d_nnz.tolist() == [2, 0, 0, 0]    # the last one has 8 slots, cap is 4
    with pytest.raises(nat.HbmRagError):
        nat.bm25_encode_dev(d_text.data_ptr(), d_off.data_ptr(), 4, 70000, enc.k1, enc.b, enc.avgdl, cap, d_idx.data_ptr(),
                            d_val.data_ptr(), d_nnz.data_ptr(), d_fl.data_ptr(), 0)
