"""Encoder / cross-encoder plugins (PyTorch): structural checks on CPU with a tiny config, and on the GPU the full
reference-shaped flow: encoder -> index_chunks -> retrieve -> cross-encoder rerank."""
import asyncio

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from advanced_rag.encoders import CrossEncoderModel, EncoderConfig, HashTokenizer, SentenceEncoder  # noqa: E402

TINY = EncoderConfig(vocab_size=5000, hidden=64, layers=2, heads=4, intermediate=128, max_len=64)


def test_tokenizer_pairs_padding_and_truncation():
    tok = HashTokenizer(5000, 16)
    ids, types = tok.encode("Hello, world!", "second text here")
    assert ids[0] == 101 and ids.count(102) == 2 and len(ids) == len(types) and types[-1] == 1 and types[0] == 0
    ids, _ = tok.encode("word " * 100)
    assert len(ids) == 16
    ids, types, mask = tok.batch(["a b c", "d"], device="cpu")
    assert ids.shape == types.shape == mask.shape and ids.shape[1] % 8 == 0 and mask[1].sum() == 3


def test_sentence_encoder_cpu_shapes_determinism_and_batching():
    enc = SentenceEncoder(TINY, device="cpu", seed=7)
    texts = ["alpha beta gamma", "delta", "alpha beta gamma", "a much longer sentence with many more tokens in it"]
    vs = enc.encode_semantic_batch(texts)
    assert len(vs) == 4 and vs[0].shape == (64,) and vs[0].dtype == np.float32
    assert np.allclose(np.linalg.norm(np.stack(vs), axis=1), 1.0, atol=1e-5)
    assert np.allclose(vs[0], vs[2], atol=1e-6)                       # padding does not leak into the result
    assert np.allclose(enc.encode_semantic(texts[3]), vs[3], atol=1e-5)   # batch == single
    again = SentenceEncoder(TINY, device="cpu", seed=7).encode_semantic_batch(texts)
    assert all(np.array_equal(a, b) for a, b in zip(again, vs))        # same seed, same weights, same call -> same bits
    assert not np.allclose(SentenceEncoder(TINY, device="cpu", seed=8).encode_semantic(texts[1]), vs[1])
    assert enc.encode_domain("x", "law").shape == (64,)
    assert SentenceEncoder(TINY, device="cpu", domain_dim=100).encode_domain("x").shape == (100,)


def test_cross_encoder_cpu_predict_and_local_weights(tmp_path):
    ce = CrossEncoderModel(TINY, device="cpu", seed=3)
    pairs = [("what is rag", "retrieval augmented generation"), ("what is rag", "a piece of cloth"), ("q", "d")]
    s = ce.predict(pairs)
    assert s.shape == (3,) and s.dtype == np.float32 and np.isfinite(s).all()
    assert np.allclose(ce.predict(pairs[:1]), s[:1], atol=1e-5)
    assert ce.flops_per_pair(64) > 0
    # round-trip through a LOCAL safetensors file with HuggingFace names
    from safetensors.torch import save_file
    sd = ce.module.state_dict()
    hf = {"bert.embeddings.word_embeddings.weight": sd["encoder.word.weight"] * 0 + 0.01,
          "classifier.weight": sd["head.weight"] * 0 + 0.5, "classifier.bias": sd["head.bias"] * 0}
    path = str(tmp_path / "m.safetensors")
    save_file({k: v.contiguous() for k, v in hf.items()}, path)
    ce.load_local(path)
    assert torch.allclose(ce.module.state_dict()["head.weight"], torch.full_like(sd["head.weight"], 0.5))


def test_cross_encoder_last_layer_for_token_0_only_gives_the_same_logits():
    """The relevance head reads token 0: the last layer computes keys / values for every token and everything else for
    token 0 alone (encoders._Layer.forward_first_token).  Same logits as the layer run over every token — fp32 on the
    CPU here (ragged lengths), fp16 on the GPU below."""
    ce = CrossEncoderModel(EncoderConfig(vocab_size=2000, hidden=64, layers=3, heads=2, intermediate=128, max_len=64),
                           device="cpu", max_len=64)
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(1, 2000, (9, 37), generator=g)
    ids[2, 12:] = 0
    ids[5, 1:] = 0
    types = (torch.arange(37)[None, :] >= 9).long().expand(9, -1).contiguous()
    mask = ids != 0
    with torch.inference_mode():
        a = ce.module(ids, types, mask)
        b = ce.module(ids, types, mask, all_tokens_last_layer=True)
    assert a.shape == (9,) and torch.allclose(a, b, atol=1e-6, rtol=1e-5)
    full, run = ce.flops_per_pair(128, executed=False), ce.flops_per_pair(128)
    assert 0.5 < run / full < 1.0


@pytest.mark.gpu
def test_cross_encoder_last_layer_for_token_0_only_on_the_gpu(gpu):
    ce = CrossEncoderModel(EncoderConfig(gelu="tanh"), device="cuda:0", max_len=512)
    g = torch.Generator(device="cuda").manual_seed(8)
    ids = torch.randint(1000, 30000, (64, 128), device="cuda", generator=g)
    ids[:, 0] = 101
    ids[3, 70:] = 0
    ids[9, 5:] = 0
    types = (torch.arange(128, device="cuda")[None, :] >= 32).long().expand(64, -1).contiguous()
    mask = ids != 0
    with torch.inference_mode():
        a = ce.module(ids, types, mask)
        b = ce.module(ids, types, mask, all_tokens_last_layer=True)
    assert torch.isfinite(a).all() and torch.allclose(a, b, atol=2e-3, rtol=2e-2), (a - b).abs().max()


@pytest.mark.gpu
def test_encoders_drive_the_pipeline_on_gpu(gpu):
    from advanced_rag import AdvancedRAGPipeline, BM25SparseEncoder, CrossEncoderReranker, PipelineConfig
    from advanced_rag.constants import RetrievalConstants
    from advanced_rag.embedding_cache import initialize_caches

    initialize_caches()
    docs = [{"id": f"doc{i}", "text": f"Passage {i} about subject{i % 5}. It mentions item{i % 3} several times. item{i % 3} again."}
            for i in range(40)]
    bm25 = BM25SparseEncoder(sparse_dim=4096).fit(d["text"] for d in docs)
    enc = SentenceEncoder(device="cuda:0", sparse_encoder=bm25, domain_dim=96, max_len=64)   # MiniLM-L6 shape, fp16
    assert enc.dim == 384 and next(enc.module.parameters()).dtype == torch.float16
    dev = enc.encode_to_device(["one", "two words"])
    assert dev.is_cuda and dev.shape == (2, 384)
    p = AdvancedRAGPipeline(config=PipelineConfig(enable_audit_logging=False, top_k=10, rerank_top_k=3),
                            semantic_dim=384, sparse_dim=4096, domain_dim=96)
    p.index_manager.embedding_generator = enc
    ce = CrossEncoderModel(device="cuda:0", max_len=96)
    p.retriever.reranker = CrossEncoderReranker(model=ce)
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    try:
        rep = asyncio.run(p.ingest_documents(docs))
        assert rep["indexing_summary"]["indexed_semantic"] == rep["chunks_created"] and not rep["indexing_summary"]["errors"]
        query = "subject2 item1"
        raw = asyncio.run(p.retriever.retrieve(query, profile_hint="default"))
        assert len(raw) == 10
        want = ce.predict([(query, r["content"]) for r in raw])
        results, _ = asyncio.run(p.retrieve(query, context={"retrieval_profile": "default"}))
        order = np.argsort(-want, kind="stable")[:3]
        assert [r.chunk_id for r in results] == [raw[i]["id"] for i in order]
        assert np.allclose([r.score for r in results], want[order], atol=2e-3)
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old
        asyncio.run(p.close())


@pytest.mark.gpu
def test_device_resident_query_embedding_cache(gpu):
    """north star: 'embedding cache -> device-resident tensor'.  Query embeddings live in one HBM table; a hit is a
    device pointer handed to the search kernel, and results equal the host-embedding path."""
    from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
    from advanced_rag.constants import RetrievalConstants
    from advanced_rag.bm25 import BM25SparseEncoder
    texts = [f"passage number {i} about thing{i % 11}" for i in range(500)]
    bm25 = BM25SparseEncoder(sparse_dim=512).fit(texts)
    enc = SentenceEncoder(TINY, device="cuda:0", dtype=torch.float32, seed=5, sparse_encoder=bm25)
    X = np.stack(enc.encode_semantic_batch(texts))
    rows = [bm25.encode_document(t) for t in texts]
    indptr = np.cumsum([0] + [len(r["indices"]) for r in rows]).astype(np.int64)
    docs_sparse = (indptr, np.concatenate([r["indices"] for r in rows]).astype(np.int32),
                   np.concatenate([r["values"] for r in rows]).astype(np.float32))
    plain = MilvusIndexManager(semantic_dim=64, sparse_dim=512, enable_domain=False)
    cached = MilvusIndexManager(semantic_dim=64, sparse_dim=512, enable_domain=False, device_embedding_cache=8)
    for m in (plain, cached):
        m.embedding_generator = enc
        m.add_rows(X, docs_sparse, ids=[f"t{i}" for i in range(500)], contents=texts)
        m.finalize()
    old = RetrievalConstants.TIMEOUT_SECONDS
    RetrievalConstants.TIMEOUT_SECONDS = 60.0
    try:
        q = "passage about thing3"
        emb = asyncio.run(cached._generate_semantic_embedding(q))
        assert emb.is_cuda and emb.shape == (64,)
        assert asyncio.run(cached._generate_semantic_embedding(q)).data_ptr() == emb.data_ptr()   # same HBM slot
        assert cached.device_cache_stats == {"hits": 1, "misses": 1}
        a = asyncio.run(plain.search(asyncio.run(plain._generate_semantic_embedding(q)), "semantic_index", 10))
        b = asyncio.run(cached.search(emb, "semantic_index", 10))
        assert [h["id"] for h in a] == [h["id"] for h in b]
        assert np.allclose([h["score"] for h in a], [h["score"] for h in b], atol=1e-6)
        for i in range(12):                                   # FIFO eviction beyond 8 slots
            asyncio.run(cached._generate_semantic_embedding(f"query {i}"))
        assert len(cached._dev_cache._slots) == 8
        out = asyncio.run(HybridRetriever(cached, RetrievalConfig(top_k=5)).retrieve(q, profile_hint="default"))
        ref = asyncio.run(HybridRetriever(plain, RetrievalConfig(top_k=5)).retrieve(q, profile_hint="default"))
        assert [h["id"] for h in out] == [h["id"] for h in ref] and len(out) == 5
    finally:
        RetrievalConstants.TIMEOUT_SECONDS = old
        asyncio.run(plain.close())
        asyncio.run(cached.close())


@pytest.mark.gpu
@pytest.mark.parametrize("rows,hidden", [(1, 8), (37, 136), (1000, 384), (513, 768), (100, 1024), (16, 64)])
def test_fused_add_layernorm_matches_torch_fp32(gpu, rows, hidden):
    """hr_add_layernorm_f16_dev (residual add + LayerNorm in one pass, fp32 statistics) against the plain PyTorch
    fp32 LayerNorm of the fp16-rounded sum: within fp16 output rounding (2e-3 absolute on unit-scale outputs); with and
    without a residual, in place, rows not a multiple of the 16 a block takes."""
    from advanced_rag import _native as nat
    g = torch.Generator(device="cuda").manual_seed(rows * 1000 + hidden)
    x = (torch.randn((rows, hidden), device="cuda", generator=g) * 2).half()
    r = torch.randn((rows, hidden), device="cuda", generator=g).half()
    gamma = (1 + 0.1 * torch.randn(hidden, device="cuda", generator=g)).half()
    beta = (0.1 * torch.randn(hidden, device="cuda", generator=g)).half()
    st = torch.cuda.current_stream().cuda_stream
    for res in (r, None):
        s = (x + res) if res is not None else x
        want = torch.nn.functional.layer_norm(s.float(), (hidden,), gamma.float(), beta.float(), 1e-12)
        out = torch.empty_like(x)
        nat.add_layernorm_f16_dev(x.data_ptr(), res.data_ptr() if res is not None else 0, gamma.data_ptr(), beta.data_ptr(),
                                  out.data_ptr(), rows, hidden, 1e-12, st)
        torch.cuda.synchronize()
        assert torch.allclose(out.float(), want, atol=2e-3, rtol=2e-3), (out.float() - want).abs().max()
    xin = x.clone()
    nat.add_layernorm_f16_dev(xin.data_ptr(), r.data_ptr(), gamma.data_ptr(), beta.data_ptr(), xin.data_ptr(), rows, hidden,
                              1e-12, st)   # in place
    torch.cuda.synchronize()
    want = torch.nn.functional.layer_norm((x + r).float(), (hidden,), gamma.float(), beta.float(), 1e-12)
    assert torch.allclose(xin.float(), want, atol=2e-3, rtol=2e-3)
    with pytest.raises(ValueError):
        nat.add_layernorm_f16_dev(x.data_ptr(), 0, gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), rows, 12, 1e-12, st)
    with pytest.raises(nat.HbmRagError):
        nat.add_layernorm_f16_dev(x.data_ptr(), 0, gamma.data_ptr(), beta.data_ptr(), x.data_ptr(), rows, 2048, 1e-12, st)


@pytest.mark.gpu
@pytest.mark.parametrize("n_seq,T,hidden", [(3, 7, 64), (40, 128, 384), (5, 33, 768), (2, 512, 1024)])
def test_fused_embedding_layernorm_matches_the_unfused_fp16_ops(gpu, n_seq, T, hidden):
    """hr_embed_layernorm_f16_dev against the ops it replaces, run the way the unfused fp16 module runs them (gather, fp16
    add of the position rows, fp16 add of the segment rows, LayerNorm with fp32 statistics): the same roundings, so only
    the last LayerNorm's arithmetic differs — within one fp16 ulp of the output; ids outside the tables are clamped."""
    from advanced_rag import _native as nat
    g = torch.Generator(device="cuda").manual_seed(hidden + T)
    V = 1000
    word = (torch.randn((V, hidden), device="cuda", generator=g) * 0.5).half()
    pos = (torch.randn((512, hidden), device="cuda", generator=g) * 0.5).half()
    seg = (torch.randn((2, hidden), device="cuda", generator=g) * 0.5).half()
    gamma = (1 + 0.1 * torch.randn(hidden, device="cuda", generator=g)).half()
    beta = (0.1 * torch.randn(hidden, device="cuda", generator=g)).half()
    ids = torch.randint(0, V, (n_seq, T), device="cuda", generator=g)
    types = torch.randint(0, 2, (n_seq, T), device="cuda", generator=g)
    out = torch.empty((n_seq, T, hidden), dtype=torch.float16, device="cuda")

    def run(i, t):
        nat.embed_layernorm_f16_dev(i.data_ptr(), t.data_ptr(), word.data_ptr(), pos.data_ptr(), seg.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), out.data_ptr(), n_seq, T, hidden, 1e-12, V, 2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return out.clone()

    got = run(ids, types)
    h = (word[ids] + pos[:T][None]) + seg[types]                      # fp16, rounded after each add
    want = torch.nn.functional.layer_norm(h.float(), (hidden,), gamma.float(), beta.float(), 1e-12)
    assert torch.allclose(got.float(), want, atol=2e-3, rtol=2e-3), (got.float() - want).abs().max()
    wild_i, wild_t = ids.clone(), types.clone()
    wild_i[0, 0], wild_i[-1, -1], wild_t[0, 0] = -5, V + 9, 7
    ref_i, ref_t = wild_i.clamp(0, V - 1), wild_t.clamp(0, 1)
    assert torch.equal(run(wild_i, wild_t), run(ref_i, ref_t))


@pytest.mark.gpu
def test_encoder_forward_with_the_fused_layernorm_matches_the_unfused_module(gpu):
    """The fp16 GPU forward (fused add + LayerNorm) against the same module run unfused in fp32 on the same weights."""
    from advanced_rag.encoders import BertEncoder
    torch.manual_seed(3)
    c = EncoderConfig(vocab_size=2000, hidden=128, layers=2, heads=4, intermediate=256, max_len=64)
    m32 = BertEncoder(c).cuda().eval()
    m16 = BertEncoder(c).cuda().eval()
    m16.load_state_dict(m32.state_dict())
    m16 = m16.half()
    ids = torch.randint(1, 2000, (5, 33), device="cuda")
    types = torch.zeros_like(ids)
    mask = torch.ones_like(ids, dtype=torch.bool)
    mask[2, 20:] = False
    with torch.no_grad():
        a = m16(ids, types, mask).float()
        b = m32(ids, types, mask)
    assert torch.allclose(a, b, atol=3e-2, rtol=3e-2), (a - b).abs().max()


@pytest.mark.gpu
def test_device_embedding_table_evicts_fifo_and_keeps_values(gpu):
    """FIFO eviction like the host cache; a row is overwritten only after the device has drained (searches enqueued with
    the evicted key's view read the old bytes); live keys keep their slot pointer and their values."""
    from advanced_rag.embedding_cache import DeviceEmbeddingTable
    t = DeviceEmbeddingTable(capacity=4, dim=16, device="cuda:0")
    vecs = {f"k{i}": torch.full((16,), float(i)) for i in range(7)}
    views = {k: t.store(k, v) for k, v in list(vecs.items())[:4]}
    ptrs = {k: v.data_ptr() for k, v in views.items()}
    old_k0 = views["k0"].clone()
    for k in ("k4", "k5", "k6"):          # evicts k0, k1, k2 in that order
        t.store(k, vecs[k])
    torch.cuda.synchronize()
    assert t.lookup("k0") is None and t.lookup("k1") is None and t.lookup("k2") is None
    assert t.lookup("k3").data_ptr() == ptrs["k3"] and torch.equal(t.lookup("k3").cpu(), vecs["k3"])
    for k in ("k4", "k5", "k6"):
        assert torch.equal(t.lookup(k).cpu(), vecs[k])
    assert torch.equal(old_k0.cpu(), vecs["k0"])   # the copy taken before the eviction still holds k0's bytes
    assert t.store("k5", vecs["k5"]).data_ptr() == t.lookup("k5").data_ptr()   # re-store of a live key: same slot


@pytest.mark.gpu
@pytest.mark.parametrize("grow", [0.0, 1.0])
@pytest.mark.parametrize("n_seq,T,heads", [(3, 16, 1), (5, 40, 4), (7, 128, 12), (2, 200, 3), (2, 512, 12), (1, 8, 2), (9, 129, 5),
                                           (2, 1024, 2), (11, 300, 7)])
def test_attention_hd32_kernel_matches_sdpa_fp32(gpu, n_seq, T, heads, grow):
    """hr_attention_f16_dev (head dimension 32: QK^T, masked online softmax and PV on the MFMA units, from the fused
    QKV buffer to the [tokens, hidden] layout) against PyTorch's scaled_dot_product_attention in fp32 on the same
    fp16-rounded inputs, ragged sequence lengths (keys at or beyond the length masked): within fp16 output rounding."""
    from advanced_rag import _native as nat
    g = torch.Generator(device="cuda").manual_seed(n_seq * 1000 + T)
    H = heads * 32
    qkv = (torch.randn((n_seq, T, 3, heads, 32), device="cuda", generator=g) * 1.5)
    # grow = 1: keys that double along the sequence — the scores of later chunks exceed the reference maximum of the
    # earlier ones by more than 2^8, so the rescale branch of the kernel's lazy softmax runs in the middle of sequences
    # too.  Scores then reach ~30 (log2 units) and the fp16 rounding of the pre-scaled Q (2^-11 relative) is worth 1 - 2 %
    # of a probability: the tolerance of that variant is the fp16 arithmetic's, not the kernel's.
    qkv[:, :, 1] *= (1.0 + grow * torch.arange(T, device="cuda")[None, :, None, None] / T)
    qkv = qkv.half()
    atol, rtol = (4e-3, 1e-2) if grow == 0.0 else (1.5e-2, 3e-2)
    lengths = torch.randint(1, T + 1, (n_seq,), device="cuda", generator=g).to(torch.int32)
    lengths[0] = T
    out = torch.full((n_seq, T, H), float("nan"), dtype=torch.float16, device="cuda")
    nat.attention_f16_dev(qkv.data_ptr(), lengths.data_ptr(), out.data_ptr(), n_seq, T, heads, 32, 32 ** -0.5,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4)
    key_ok = torch.arange(T, device="cuda")[None, :] < lengths[:, None]
    bias = torch.zeros((n_seq, 1, 1, T), device="cuda").masked_fill(~key_ok[:, None, None, :], float("-inf"))
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias).transpose(1, 2).reshape(n_seq, T, H)
    assert torch.isfinite(out).all()
    assert torch.allclose(out.float(), want, atol=atol, rtol=rtol), (out.float() - want).abs().max()
    # no lengths = every key is valid
    nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), n_seq, T, heads, 32, 32 ** -0.5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n_seq, T, H)
    assert torch.allclose(out.float(), want, atol=atol, rtol=rtol)
    with pytest.raises(nat.HbmRagError):
        nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), n_seq, T, heads, 64, 0.125, 0)
    # operands by pointer and stride: the first nq tokens as queries over a separate [n_seq, T, 2, heads, 32] K / V buffer
    kvbuf = qkv[:, :, 1:].contiguous()
    for nq in sorted({1, min(T, 33), min(T, 160)}):
        qrows = qkv[:, :nq, 0].contiguous()                                   # [n_seq, nq, heads, 32]
        o2 = torch.full((n_seq, nq, H), float("nan"), dtype=torch.float16, device="cuda")
        nat.attention_rows_f16_dev(qrows.data_ptr(), nq * H, H, kvbuf.data_ptr(), kvbuf.data_ptr() + 2 * H, T * 2 * H, 2 * H,
                                   lengths.data_ptr(), o2.data_ptr(), n_seq, T, nq, heads, 32, 32 ** -0.5,
                                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        w2 = torch.nn.functional.scaled_dot_product_attention(q[:, :, :nq], k, v, attn_mask=bias).transpose(1, 2).reshape(n_seq, nq, H)
        assert torch.allclose(o2.float(), w2, atol=atol, rtol=rtol), (nq, (o2.float() - w2).abs().max())
    with pytest.raises(nat.HbmRagError):   # K and V of a (sequence, head) must fit LDS
        nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), 1, 1025, heads, 32, 0.125, 0)
