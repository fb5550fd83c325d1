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
    # and against plain fp32 PyTorch on the fp16-rounded tables (no fp16 rounding between the adds): the kernel's two fp16
    # roundings of the sum (|h| < 4: half an ulp = 1e-3 each) pass through the LayerNorm's gain of ~1.5
    h32 = word.float()[ids] + pos.float()[:T][None] + seg.float()[types]
    want32 = torch.nn.functional.layer_norm(h32, (hidden,), gamma.float(), beta.float(), 1e-12)
    assert torch.allclose(got.float(), want32, atol=6e-3, rtol=2e-3), (got.float() - want32).abs().max()
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
@pytest.mark.parametrize("n_seq,T,heads,hd", [(3, 16, 1, 32), (5, 40, 4, 32), (7, 128, 12, 32), (2, 200, 3, 32), (2, 512, 12, 32),
                                              (1, 8, 2, 32), (9, 129, 5, 32), (2, 1024, 2, 32), (11, 300, 7, 32),
                                              (3, 16, 1, 64), (5, 40, 4, 64), (7, 128, 12, 64), (2, 200, 3, 64), (2, 512, 12, 64),
                                              (1, 8, 2, 64), (9, 129, 5, 64), (3, 300, 16, 64)])
def test_attention_kernel_matches_sdpa_fp32(gpu, n_seq, T, heads, hd, grow):
    """hr_attention_f16_dev (head dimensions 32 and 64: QK^T, masked online softmax and PV on the MFMA units, from the fused
    QKV buffer to the [tokens, hidden] layout) against PyTorch's scaled_dot_product_attention in fp32 on the same
    fp16-rounded inputs, ragged sequence lengths (keys at or beyond the length masked): within fp16 output rounding."""
    from advanced_rag import _native as nat
    g = torch.Generator(device="cuda").manual_seed(n_seq * 1000 + T + hd)
    H = heads * hd
    qkv = (torch.randn((n_seq, T, 3, heads, hd), device="cuda", generator=g) * (1.5 if hd == 32 else 1.25))
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
    nat.attention_f16_dev(qkv.data_ptr(), lengths.data_ptr(), out.data_ptr(), n_seq, T, heads, hd, hd ** -0.5,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    q, k, v = qkv.float().permute(2, 0, 3, 1, 4)
    key_ok = torch.arange(T, device="cuda")[None, :] < lengths[:, None]
    bias = torch.zeros((n_seq, 1, 1, T), device="cuda").masked_fill(~key_ok[:, None, None, :], float("-inf"))
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=bias).transpose(1, 2).reshape(n_seq, T, H)
    assert torch.isfinite(out).all()
    assert torch.allclose(out.float(), want, atol=atol, rtol=rtol), (out.float() - want).abs().max()
    # no lengths = every key is valid
    nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), n_seq, T, heads, hd, hd ** -0.5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n_seq, T, H)
    assert torch.allclose(out.float(), want, atol=atol, rtol=rtol)
    # fragment-order output (what the fused layer tail reads): the same values, the other layout
    from advanced_rag.encoder_kernels import fr_rows, from_fragment_order
    ofr = torch.zeros((fr_rows(n_seq * T), H), dtype=torch.float16, device="cuda")
    nat.attention_fr_f16_dev(qkv.data_ptr(), 0, ofr.data_ptr(), n_seq, T, heads, hd, hd ** -0.5, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    # (the two store paths round the normalised output through different instruction pairs: a last-place difference here and there)
    assert torch.allclose(from_fragment_order(ofr, n_seq * T).float(), out.reshape(n_seq * T, H).float(), atol=2.5e-4, rtol=1e-3)
    with pytest.raises(nat.HbmRagError):   # head dimensions other than 32 / 64 are refused, not approximated
        nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), n_seq, T, heads, 48, 0.125, 0)
    # operands by pointer and stride: the first nq tokens as queries over a separate [n_seq, T, 2, heads, hd] K / V buffer
    kvbuf = qkv[:, :, 1:].contiguous()
    for nq in sorted({1, min(T, 33), min(T, 160)}):
        qrows = qkv[:, :nq, 0].contiguous()                                   # [n_seq, nq, heads, hd]
        o2 = torch.full((n_seq, nq, H), float("nan"), dtype=torch.float16, device="cuda")
        nat.attention_rows_f16_dev(qrows.data_ptr(), nq * H, H, kvbuf.data_ptr(), kvbuf.data_ptr() + 2 * H, T * 2 * H, 2 * H,
                                   lengths.data_ptr(), o2.data_ptr(), n_seq, T, nq, heads, hd, hd ** -0.5,
                                   torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        w2 = torch.nn.functional.scaled_dot_product_attention(q[:, :, :nq], k, v, attn_mask=bias).transpose(1, 2).reshape(n_seq, nq, H)
        assert torch.allclose(o2.float(), w2, atol=atol, rtol=rtol), (nq, (o2.float() - w2).abs().max())
    with pytest.raises(nat.HbmRagError):   # K and V of a (sequence, head) must fit LDS
        nat.attention_f16_dev(qkv.data_ptr(), 0, out.data_ptr(), 1, 1025 if hd == 32 else 513, heads, hd, 0.125, 0)


# --------------------------------------------------------------------------- hand-written layer kernels (round 4)
def test_weight_packers_follow_the_documented_index_formulas():
    """encoder_kernels.pack_natural / pack_accumulator_order / pack_tail_stream against the element-by-element definitions
    of include/hbmrag.h (CPU: pure reshapes)."""
    from advanced_rag.encoder_kernels import pack_accumulator_order, pack_natural, pack_tail_stream
    N, K = 48, 96
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 2039          # exactly representable in fp16
    nat_p = pack_natural(w).reshape(N // 16, K // 32, 4, 16, 8)
    acc_p = pack_accumulator_order(w).reshape(N // 16, K // 32, 4, 16, 8)
    for t in range(N // 16):
        for s in range(K // 32):
            for g in range(4):
                for r in (0, 7, 15):
                    for j in range(8):
                        assert nat_p[t, s, g, r, j] == w[16 * t + r, 32 * s + 8 * g + j]
                        kk = 32 * s + (4 * g + j if j < 4 else 16 + 4 * g + j - 4)
                        assert acc_p[t, s, g, r, j] == w[16 * t + r, kk]
    H, I = 64, 128
    gen = torch.Generator().manual_seed(5)
    w_out, w_up, w_down = (torch.randn(shape, generator=gen).half().float() for shape in ((H, H), (I, H), (H, I)))
    stream = pack_tail_stream(w_out, w_up, w_down).reshape(-1, H // 16, 512)   # [stage][piece][halves]
    n = I // 32
    assert stream.shape[0] == H // 32 + 2 * n
    out_p = pack_accumulator_order(w_out).reshape(H // 16, H // 32, 512)    # the attention output arrives in fragment order
    up_p = pack_accumulator_order(w_up).reshape(I // 16, H // 32, 512)
    down_p = pack_accumulator_order(w_down).reshape(H // 16, I // 32, 512)
    hs = H // 32
    for so in range(hs):                                                        # two output tiles of W_out per stage
        assert torch.equal(stream[so].reshape(2, hs, 512), out_p[2 * so: 2 * so + 2])
    order = [("up", 0)] + [x for c in range(n - 1) for x in (("up", c + 1), ("down", c))] + [("down", n - 1)]
    for i, (kind, c) in enumerate(order):
        got = stream[hs + i]
        if kind == "up":
            assert torch.equal(got.reshape(2, hs, 512), up_p[2 * c: 2 * c + 2])
        else:
            assert torch.equal(got, down_p[:, c])
    from advanced_rag.encoder_kernels import store_order_rows
    perm = store_order_rows(64)
    assert sorted(perm.tolist()) == list(range(64))
    for so in range(2):          # lane group g of a stage holds rows 4 g .. 4 g + 3 of both tiles = features 32 so + 8 g .. + 7
        for g_ in range(4):
            got = [int(perm[32 * so + 16 * u + 4 * g_ + r]) for u in range(2) for r in range(4)]
            assert got == list(range(32 * so + 8 * g_, 32 * so + 8 * g_ + 8))
    # fragment order of activations: X_fr[tile][s][16 g + c][j] = X[16 tile + c][32 s + 16 (j >> 2) + 4 g + (j & 3)]
    from advanced_rag.encoder_kernels import fr_rows, from_fragment_order, to_fragment_order
    M, Hx = 37, 96
    x = (torch.arange(M * Hx, dtype=torch.float32).reshape(M, Hx) % 2039).half()
    fr = to_fragment_order(x)
    assert fr.shape == (fr_rows(M), Hx) == (48, 96)
    v = fr.reshape(3, Hx // 32, 4, 16, 8)
    for tile, s_, g_, c_, j in ((0, 0, 0, 0, 0), (1, 2, 3, 5, 6), (2, 1, 2, 4, 3), (0, 2, 1, 15, 7)):
        row = 16 * tile + c_
        assert v[tile, s_, g_, c_, j] == (x[row, 32 * s_ + 16 * (j >> 2) + 4 * g_ + (j & 3)] if row < M else 0)
    assert torch.equal(from_fragment_order(fr, M), x)


def _ref_layer_tail(a, x, layer, gelu):
    """fp32 arithmetic of everything after the attention on fp16-rounded operands."""
    f = lambda t: t.float()   # noqa: E731
    x1 = torch.nn.functional.layer_norm(f(x) + f(a) @ f(layer.out.weight).t() + f(layer.out.bias), (x.shape[-1],),
                                        f(layer.ln1.weight), f(layer.ln1.bias), layer.ln1.eps)
    x1 = x1.half().float()                                     # the kernel hands x1 to the FFN (and the residual) as fp16
    h = x1 @ f(layer.up.weight).t() + f(layer.up.bias)
    h = torch.nn.functional.gelu(h, approximate="tanh" if gelu == "tanh" else "none").half().float()
    y = x1 + h @ f(layer.down.weight).t() + f(layer.down.bias)
    return torch.nn.functional.layer_norm(y, (x.shape[-1],), f(layer.ln2.weight), f(layer.ln2.bias), layer.ln2.eps)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 127, 128, 129, 1000, 4096 + 77, 65536, 65536 + 300 + 13])   # from 65 536 rows: 4 token tiles per wave
def test_linear_rows_kernel_matches_fp32(gpu, rows):
    """hr_linear_rows_f16_dev (K = 384; N = 1152 the QKV projection, 768 keys + values, 32 the smallest) against an fp32
    matmul of the same fp16-rounded operands; ragged row counts, a strided output."""
    from advanced_rag.encoder_kernels import linear_rows, pack_linear, to_fragment_order
    g = torch.Generator(device="cuda").manual_seed(rows)
    x = torch.randn((rows, 384), device="cuda", generator=g).half()
    for N in (1152, 768, 32):
        w = (torch.randn((N, 384), device="cuda", generator=g) * 0.05).half()
        b = torch.randn((N,), device="cuda", generator=g)
        want = x.float() @ w.float().t() + b
        out = torch.full((rows, N + 8), float("nan"), dtype=torch.float16, device="cuda")
        linear_rows(x, *pack_linear(w, b, False), N, out=out[:, :N])
        torch.cuda.synchronize()
        assert torch.isnan(out[:, N:]).all()                 # nothing written beyond the N columns of a strided row
        assert torch.allclose(out[:, :N].float(), want, atol=4e-3, rtol=4e-3), (N, (out[:, :N].float() - want).abs().max())
        # the same rows handed over in fragment order, the weights in accumulator k order (another summation order inside
        # the MFMA: equal up to fp32 rounding)
        out2 = linear_rows(to_fragment_order(x), *pack_linear(w, b, True), N, rows=rows, x_fr=True)
        torch.cuda.synchronize()
        assert torch.allclose(out2.float(), want, atol=4e-3, rtol=4e-3)
        assert torch.allclose(out2.float(), out[:, :N].float(), atol=4e-3, rtol=4e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("gelu", ["tanh", "erf"])
@pytest.mark.parametrize("rows", [1, 128, 200, 2560 + 33])
def test_encoder_tail_kernel_matches_fp32(gpu, rows, gelu):
    """hr_encoder_tail_f16_dev — output projection + residual + LayerNorm + FFN (tanh / erf GELU) + residual + LayerNorm in
    one launch, the FFN intermediate kept on chip — against fp32 PyTorch on the same fp16-rounded operands, with
    non-trivial LayerNorm parameters and biases (random-init models have gamma = 1, beta = 0, bias = 0)."""
    from advanced_rag.encoder_kernels import LayerKernels, encoder_tail, from_fragment_order, to_fragment_order
    from advanced_rag.encoders import _Layer
    torch.manual_seed(rows)
    layer = _Layer(EncoderConfig(gelu=gelu))
    with torch.no_grad():
        for lin in (layer.qkv, layer.out, layer.up, layer.down):
            lin.weight.normal_(0, 0.05)
            lin.bias.normal_(0, 0.1)
        for ln in (layer.ln1, layer.ln2):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.normal_(0, 0.2)
    layer = layer.to("cuda", torch.float16)
    g = torch.Generator(device="cuda").manual_seed(rows + 1)
    a = torch.randn((rows, 384), device="cuda", generator=g).half()
    x = torch.randn((rows, 384), device="cuda", generator=g).half()
    k = LayerKernels().ensure(layer)
    out = encoder_tail(to_fragment_order(a), x, k, 1536, layer.ln1.eps, gelu == "erf")
    torch.cuda.synchronize()
    want = _ref_layer_tail(a, x, layer, gelu)
    assert torch.isfinite(out).all()
    err = (out.float() - want).abs()
    assert torch.allclose(out.float(), want, atol=1.5e-2, rtol=1.5e-2), (err.max(), err.mean())
    assert err.mean() < 1.5e-3                                # fp16 output rounding of O(1) values is ~2e-4 on average
    # residual and output in fragment order: the same bits as row-major
    out_fr = encoder_tail(to_fragment_order(a), to_fragment_order(x), k, 1536, layer.ln1.eps, gelu == "erf", rows=rows, x_fr=True,
                          out_fr=True)
    torch.cuda.synchronize()
    assert torch.equal(from_fragment_order(out_fr, rows), out)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [16, 128, 200])
def test_layer_kernels_equal_the_unfused_forward(gpu, T):
    """A whole MiniLM-shaped model with the three hand-written launches per layer against the same weights through the
    PyTorch GEMMs + fused elementwise kernels (use_layer_kernels = False) and against fp32: cross-encoder logits (the last
    layer for token 0 only) and sentence embeddings, ragged lengths."""
    ce = CrossEncoderModel(EncoderConfig(), device="cuda:0", seed=5)
    n = 37
    g = torch.Generator(device="cuda").manual_seed(T)
    ids = torch.randint(1000, 30000, (n, T), device="cuda", generator=g)
    ids[:, 0] = 101
    types = (torch.arange(T, device="cuda")[None, :] >= T // 3).long().expand(n, T).contiguous()
    mask = torch.arange(T, device="cuda")[None, :] < torch.randint(2, T + 1, (n, 1), device="cuda", generator=g)
    mask[0] = True
    ids = ids.masked_fill(~mask, 0)
    with torch.inference_mode():
        fused = ce.module(ids, types, mask)
        for layer in ce.module.encoder.layers:
            layer.use_layer_kernels = False
        plain = ce.module(ids, types, mask)
        ref = CrossEncoderModel(EncoderConfig(), device="cuda:0", dtype=torch.float32, seed=5)
        want = ref.module(ids, types, mask)
    assert torch.allclose(fused, want, atol=3e-2, rtol=3e-2), (fused - want).abs().max()
    assert (fused - want).abs().max() <= 2.0 * (plain - want).abs().max() + 5e-3   # no worse than the fp16 GEMM path
    enc = SentenceEncoder(EncoderConfig(), device="cuda:0", seed=6)
    with torch.inference_mode():
        e1 = enc.module(ids, types, mask)
        for layer in enc.module.encoder.layers:
            layer.use_layer_kernels = False
        e2 = enc.module(ids, types, mask)
    assert torch.allclose(e1, e2, atol=5e-3), (e1 - e2).abs().max()


@pytest.mark.gpu
def test_hash_tokenizer_on_the_device_equals_the_host_tokenizer(gpu):
    """hr_hash_tokenize_dev (csrc/text.h) against HashTokenizer.encode / .batch on the host: ids, row lengths, padding,
    the batch width, the mask — identical tensors.  Words and digits, every ASCII punctuation and control character as a
    token of its own, all the characters Python's \\s knows below 0x80 (incl. 0x1c - 0x1f), truncation at max_len, empty and
    one-character texts, a word that spans many threads' slices; a batch with a non-ASCII text takes the host path."""
    rng = np.random.default_rng(5)
    words = [f"tok{i}" for i in range(500)] + ["MiXeD", "under_score", "42", "x"]
    seps = [" ", ", ", ". ", "\n", "\t", " - ", "!?", "(", ")", "'s ", "\x1c", "\x1f ", "\x0b", "\x01", "~", "@#$%"]
    texts = ["".join(str(rng.choice(words)) + str(rng.choice(seps)) for _ in range(int(n))) for n in rng.integers(1, 200, size=120)]
    texts += ["", " ", "x", "!", "a" * 3000, "...", " lead and trail ", "".join(chr(c) for c in range(1, 128))]
    for vocab, max_len in ((30522, 64), (5000, 16), (30522, 250), (1001 + 7, 512)):
        tok = HashTokenizer(vocab, max_len)
        for chunk in (texts, texts[:1], texts[120:123]):
            got = tok.batch(chunk, device="cuda:0")
            want = HashTokenizer(vocab, max_len).batch(chunk, device="cpu")
            assert got[0].is_cuda
            for g, w in zip(got, want):
                assert g.shape == w.shape and torch.equal(g.cpu(), w), (vocab, max_len, len(chunk))
    tok = HashTokenizer(30522, 64)
    assert tok._batch_on_device(["plain", "ünïcode"], torch.device("cuda:0")) is None
    got, want = tok.batch(["plain", "ünïcode"], device="cuda:0"), tok.batch(["plain", "ünïcode"], device="cpu")
    assert all(torch.equal(g.cpu(), w) for g, w in zip(got, want))


@pytest.mark.gpu
def test_small_batches_replay_a_captured_graph_and_match_the_eager_forward(gpu):
    """SentenceEncoder.encode_to_device for a lone query or a handful (the query side of retrieve()): the forward is
    captured once per quantised shape as a HIP graph and replayed.  Same embeddings as the eager forward (fp16 noise:
    the padded width differs), one graph per shape, larger batches stay eager, and two threads may share the encoder."""
    import threading
    import time
    for cfg in (EncoderConfig(), EncoderConfig(hidden=768, layers=12, heads=12, intermediate=3072)):
        enc = SentenceEncoder(cfg, device="cuda:0", max_len=64, seed=3)
        eager = SentenceEncoder(cfg, device="cuda:0", max_len=64, seed=3)
        eager.use_graphs = False
        texts = ["what is retrieval augmented generation", "short", "a b c d e f g h i j k l m n o p q r s t u v w x y z " * 2,
                 "hybrid dense sparse fusion rerank", "x"]
        for group in ([texts[0]], texts[:2], texts[:3], texts, [texts[2]], [texts[0]]):
            got, want = enc.encode_to_device(group), eager.encode_to_device(group)
            assert got.shape == want.shape and torch.allclose(got, want, atol=4e-3), (got - want).abs().max()
        assert set(enc._graphs) <= {(b, w) for b in (1, 2, 4, 8, 16, 32, 64) for w in (16, 32, 64)} and len(enc._graphs) >= 3
        assert not getattr(eager, "_graphs", {})
        n_graphs = len(enc._graphs)
        big = enc.encode_to_device(texts * 20, batch_size=100)     # 100 texts in one batch: eager
        assert big.shape[0] == 100 and len(enc._graphs) == n_graphs and torch.allclose(big[:5], enc.encode_to_device(texts), atol=4e-3)
        mid = enc.encode_to_device(texts * 8)                      # 40 texts: the 64-text graph
        assert torch.allclose(mid, eager.encode_to_device(texts * 8), atol=4e-3) and len(enc._graphs) == n_graphs + 1
        # replay is what makes the lone query cheap: time both (reported, not asserted beyond "not slower by much")
        for e in (enc, eager):
            e.encode_to_device([texts[0]])
        torch.cuda.synchronize()
        t = {}
        for name, e in (("graph", enc), ("eager", eager)):
            t0 = time.perf_counter()
            for _ in range(20):
                e.encode_to_device([texts[0]])
            torch.cuda.synchronize()
            t[name] = (time.perf_counter() - t0) / 20 * 1e3
        print(f"hidden {cfg.hidden} x {cfg.layers}: lone-query encode {t['graph']:.3f} ms replayed, {t['eager']:.3f} ms eager")
        assert t["graph"] < 1.5 * t["eager"]
        if cfg.hidden == 384:   # the 20 pairs of one rerank through the cross-encoder: replayed too, same logits
            ce, ce_eager = CrossEncoderModel(cfg, device="cuda:0", max_len=128, seed=5), CrossEncoderModel(cfg, device="cuda:0", max_len=128, seed=5)
            ce_eager.use_graphs = False
            pairs = [(texts[0], texts[i % 5] + f" document {i} " + "filler words " * (i % 7)) for i in range(20)]
            for group in (pairs, pairs[:1], pairs[:7]):
                a_, b_ = ce.predict_to_device(group), ce_eager.predict_to_device(group)
                assert a_.shape == b_.shape and torch.allclose(a_.float(), b_.float(), atol=3e-3), (a_ - b_).abs().max()
            assert ce._graphs and not getattr(ce_eager, "_graphs", {})
            ce.predict(pairs); ce_eager.predict(pairs)
            torch.cuda.synchronize()
            for name, m in (("replayed", ce), ("eager", ce_eager)):
                t0 = time.perf_counter()
                for _ in range(20):
                    m.predict_to_device(pairs)
                torch.cuda.synchronize()
                print(f"cross-encoder, 20 pairs: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms {name}")
        out = {}

        def worker(k):
            out[k] = [enc.encode_to_device([texts[k % 5]]).clone() for _ in range(10)]
        th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        [x.start() for x in th]
        [x.join() for x in th]
        torch.cuda.synchronize()
        for k in range(4):
            ref = eager.encode_to_device([texts[k % 5]])
            assert all(torch.allclose(v, ref, atol=4e-3) for v in out[k])
