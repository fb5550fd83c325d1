"""BASELINE config 4 end to end on one GPU at 1M rows: hybrid dense + sparse -> RRF -> cross-encoder rerank 20 -> 5 as the
pipelined engine runs it in bench.py (the cross-encoder forward is the post-hook of the finishing stream), checked
against (a) the sequential engine for the fused candidates and (b) torch.topk over an fp32 forward of the same
cross-encoder weights for the 20 -> 5 cut (reference retrieval.py:518-563 with a CrossEncoderReranker, :651-681)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from advanced_rag import _native as nat
from advanced_rag.engine import EngineConfig, HybridSearchEngine, PipelinedSearchEngine, pack_sparse_queries

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_config4_hybrid_plus_cross_encoder_cut_at_1m_rows(gpu):
    import bench
    from advanced_rag.encoders import CrossEncoderModel
    N, D, B, T, top_k, keep = 1_000_000, 768, 32, 64, 20, 5
    dev = torch.device("cuda:0")
    h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, bench.SPARSE_DIM, 0)
    h.reserve(N)
    g = torch.Generator(device=dev).manual_seed(4)
    for b in range(4):
        x = torch.randn((N // 4, D), device=dev, generator=g, dtype=torch.float32).to(torch.float16)
        h.add_dense_dev(x.data_ptr(), N // 4)
        h.add_sparse(*bench.sparse_block(b, N // 4))
    h.finalize()
    cfg = EngineConfig(top_k=top_k, rerank_top_k=keep)
    ce16 = CrossEncoderModel(device=str(dev), max_len=T, seed=11)
    ce32 = CrossEncoderModel(device=str(dev), max_len=T, seed=11, dtype=torch.float32)
    vocab = ce16.config.vocab_size
    pos = torch.arange(T, device=dev, dtype=torch.int64)[None, None, :]
    types = torch.zeros((B * top_k, T), dtype=torch.long, device=dev)
    types[:, T // 4:] = 1
    mask = torch.ones((B * top_k, T), dtype=torch.bool, device=dev)
    mask[::3, T - 10:] = False                      # ragged pairs: padding at the tail
    qslot = torch.arange(B, device=dev)[:, None, None]

    def tokens(fused):
        toks = (1000 + (fused.clamp_min(0)[:, :, None] * 7919 + pos * 104729 + qslot * 31) % (vocab - 1000)).view(B * top_k, T)
        toks[:, 0] = 101
        return toks.masked_fill(~mask, 0)

    def hook(b):
        fused = b["fused_ids"]
        with torch.inference_mode():
            scores = ce16.module(tokens(fused), types, mask).view(B, top_k).float().masked_fill(fused < 0, float("-inf"))
        top = torch.topk(scores, keep, dim=1)
        b["ce_ids"], b["ce_scores"], b["ce_all"] = torch.gather(fused, 1, top.indices), top.values, scores

    Q, SQ = bench.make_queries(3, B, D)
    seq = HybridSearchEngine(h, cfg, device=str(dev))
    pipe = PipelinedSearchEngine(h, cfg, device=str(dev), depth=2)
    pipe.post_hook = hook
    got = []
    for i in range(3):
        q, sq = torch.from_numpy(Q[i]).to(dev), pipe.upload_sparse(pack_sparse_queries(SQ[i], 0.2))
        o = pipe.submit(q, sq)
        with torch.cuda.stream(pipe.light):
            got.append({k: o[k].clone() for k in ("fused_ids", "fused_scores", "ce_ids", "ce_scores", "ce_all", "flags")})
    pipe.synchronize()
    for i in range(3):
        q, sq = torch.from_numpy(Q[i]).to(dev), seq.upload_sparse(pack_sparse_queries(SQ[i], 0.2))
        want = seq.search(q, sq)
        torch.cuda.synchronize()
        assert int(got[i]["flags"].min()) == 1
        assert torch.equal(got[i]["fused_ids"], want["fused_ids"]) and torch.equal(got[i]["fused_scores"], want["fused_scores"])
        fused = got[i]["fused_ids"]
        assert int((fused >= 0).sum()) == B * top_k
        with torch.inference_mode():
            ref = ce32.module(tokens(fused), types, mask).view(B, top_k)
        assert torch.allclose(got[i]["ce_all"], ref, atol=3e-2, rtol=3e-2), (got[i]["ce_all"] - ref).abs().max()
        # the cut: every kept candidate is within fp16 noise of the fp32 top-5, and where the fp32 scores are separated
        # by more than that noise the kept sets are identical
        ref_top = torch.topk(ref, keep, dim=1)
        kept_ref_scores = torch.gather(ref, 1, torch.stack([(fused[b][:, None] == got[i]["ce_ids"][b][None, :]).float().argmax(0)
                                                             for b in range(B)]))
        assert bool((kept_ref_scores >= ref_top.values[:, -1:] - 3e-2).all())
        ref_sorted = torch.sort(ref, dim=1, descending=True).values
        clear = (ref_sorted[:, keep - 1] - ref_sorted[:, keep]) > 6e-2
        same = torch.tensor([set(got[i]["ce_ids"][b].tolist()) == set(torch.gather(fused, 1, ref_top.indices)[b].tolist())
                             for b in range(B)], device=dev)
        assert bool(same[clear].all())
    # the same three batches with the forward given the chip to itself (engine.post_hook_exclusive: the hook of batch i is
    # enqueued behind the scans of batch i + 1, the last one by synchronize()): same candidates, same cut
    recorded = []

    def recording_hook(b):
        hook(b)
        recorded.append({k: b[k].clone() for k in ("fused_ids", "ce_ids", "ce_scores", "ce_all")})

    pipe.post_hook, pipe.post_hook_exclusive = recording_hook, True
    for i in range(3):
        q, sq = torch.from_numpy(Q[i]).to(dev), pipe.upload_sparse(pack_sparse_queries(SQ[i], 0.2))
        pipe.submit(q, sq)
        assert len(recorded) == i          # batch i's hook waits for the next submit
    pipe.synchronize()
    assert len(recorded) == 3
    for i in range(3):
        assert torch.equal(recorded[i]["fused_ids"], got[i]["fused_ids"])
        assert torch.equal(recorded[i]["ce_ids"], got[i]["ce_ids"])
        assert torch.allclose(recorded[i]["ce_all"], got[i]["ce_all"], atol=1e-3, rtol=0)
    pipe.close()
    h.close()
