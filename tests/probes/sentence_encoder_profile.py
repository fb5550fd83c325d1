"""Ad-hoc probe: forward of the bge-base-shaped SentenceEncoder (hidden 768, 12 heads x 64, 12 layers) on 128 sequences x 256
tokens — what `bench.py --ingest` runs per call — with the top kernels."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag.encoders import EncoderConfig, SentenceEncoder
B, T = 128, 256
cfg = EncoderConfig(hidden=768, layers=12, intermediate=3072, heads=12)
enc = SentenceEncoder(cfg, device="cuda:0", max_len=T, batch_size=B)
ids = torch.randint(1000, 30000, (B, T), device="cuda"); types = torch.zeros_like(ids); mask = torch.ones_like(ids, dtype=torch.bool)
with torch.inference_mode():
    for _ in range(3): enc.module(ids, types, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): enc.module(ids, types, mask)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    print(f"forward {B} x {T}: {ms:.2f} ms", flush=True)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        for _ in range(3): enc.module(ids, types, mask)
        torch.cuda.synchronize()
    print(p.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=80))
texts = [" ".join(f"w{(i * 31 + j) % 20000}" for j in range(400)) for i in range(B)]
t0 = time.perf_counter(); enc.encode_to_device(texts); torch.cuda.synchronize(); print(f"encode_to_device of {B} texts (tokenizer + forward): {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); enc.encode_to_device(texts); torch.cuda.synchronize(); print(f"again: {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); enc.tokenizer.batch(texts, device="cuda:0"); print(f"tokenizer.batch alone: {(time.perf_counter() - t0) * 1e3:.1f} ms")
