"""Ad-hoc probe: MiniLM-L6-H384 forward, hand-written layer kernels vs the PyTorch GEMM path, by number of rows
(python tests/probes/fused_crossover.py).  Prints ms per forward for both, eager and (small shapes) replayed as a graph."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag.encoders import CrossEncoderModel

dev = "cuda:0"
ce = CrossEncoderModel(device=dev, max_len=512)
for B, T in ((1, 16), (1, 64), (1, 128), (4, 128), (20, 128), (64, 128), (256, 128), (1024, 128), (20, 512)):
    ids = torch.randint(1000, 30000, (B, T), device=dev); ids[:, 0] = 101
    types = torch.zeros_like(ids); mask = torch.ones((B, T), dtype=torch.bool, device=dev)
    res = {}
    for fused in (True, False):
        for l in ce.module.encoder.layers:
            l.use_layer_kernels = fused
        with torch.inference_mode():
            for _ in range(3):
                ce.module(ids, types, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ce.module(ids, types, mask)
            torch.cuda.synchronize()
            res[("eager", fused)] = (time.perf_counter() - t0) / 20 * 1e3
            if B * T <= 4096:
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    ce.module(ids, types, mask)
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(g):
                    out = ce.module(ids, types, mask)
                g.replay(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                res[("graph", fused)] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"rows {B * T:7d} (B {B}, T {T}): eager fused {res[('eager', True)]:.3f} / gemm {res[('eager', False)]:.3f} ms;"
          + (f" graph fused {res[('graph', True)]:.3f} / gemm {res[('graph', False)]:.3f} ms" if ("graph", True) in res else ""))
