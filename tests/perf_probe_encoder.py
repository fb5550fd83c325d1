"""Ad-hoc probe (not a test): forward pass of the sentence encoder (bge-base shape by default: hidden 768, 12 layers, 12 heads x 64,
intermediate 3072) on `n` sequences of T tokens; prints ms per forward, TFLOP/s and the share of the fp16 MFMA peak, plus the top kernels.
    python tests/perf_probe_encoder.py [T=512] [n=1024] [hidden=768] [layers=12]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag.encoders import EncoderConfig, SentenceEncoder  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 768
L = int(sys.argv[4]) if len(sys.argv) > 4 else 12
cfg = EncoderConfig(hidden=H, layers=L, heads=H // 64 if H >= 768 else 12, intermediate=4 * H, max_len=512)
enc = SentenceEncoder(cfg, device="cuda:0", max_len=512)
dev = torch.device("cuda:0")
ids = torch.randint(1000, 30000, (n, T), device=dev)
ids[:, 0] = 101
types = torch.zeros((n, T), dtype=torch.long, device=dev)
mask = torch.ones((n, T), dtype=torch.bool, device=dev)
flops = n * L * (2 * T * (4 * H * H + 2 * H * 4 * H) + 4 * T * T * H)
for use_kernels in (True, False):
    for layer in enc.module.encoder.layers:
        layer.use_layer_kernels = use_kernels
    with torch.inference_mode():
        for _ in range(2):
            enc.module(ids, types, mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            enc.module(ids, types, mask)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = flops / (ms * 1e-3) / 1e12
        print(f"hidden {H} x {L} layers, {n} x {T} tokens, layer kernels {use_kernels}: {ms:8.3f} ms/forward  {tf:7.1f} TFLOP/s = {tf / 2500:.3f} of the fp16 MFMA peak", flush=True)
        if os.environ.get("PROBE_PROFILE") == "1":
            from torch.profiler import ProfilerActivity, profile as prof
            with prof(activities=[ProfilerActivity.CUDA]) as p:
                enc.module(ids, types, mask)
                torch.cuda.synchronize()
            print(p.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=80), flush=True)
