"""Ad-hoc probe (not a test): forward pass of the cross-encoder leg (2560 pairs = 128 queries x 20 candidates) at a given
sequence length, in the arrangements under study; prints ms per forward, share of the fp16 MFMA peak and the top kernels.
    python tests/perf_probe_ce.py [T=128] [pairs=2560]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag.encoders import CrossEncoderModel, EncoderConfig  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2560
dev = torch.device("cuda:0")
ids = torch.randint(1000, 30000, (P, T), device=dev)
ids[:, 0] = 101
types = torch.zeros((P, T), dtype=torch.long, device=dev)
types[:, T // 4:] = 1
mask = torch.ones((P, T), dtype=torch.bool, device=dev)


def run(name, ce, tunable=False, profile=False):
    if tunable:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("PROBE_TUNE_MS", "200")))
        torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("PROBE_TUNE_ITERS", "100")))
        torch.cuda.tunable.set_filename(os.path.join(ROOT, "gpurun_out", "tunableop_ce.csv"))
    with torch.inference_mode():
        t0 = time.time()
        for _ in range(3):
            ce.module(ids, types, mask)
        torch.cuda.synchronize()
        warm = time.time() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ce.module(ids, types, mask)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = P * ce.flops_per_pair(T) / (ms * 1e-3) / 1e12
        print(f"{name:34s} {ms:8.3f} ms/forward  {tf:7.1f} TFLOP/s executed = {tf / 2500:.3f} of the fp16 MFMA peak (warm-up {warm:.1f}s)", flush=True)
        if profile:
            from torch.profiler import ProfilerActivity, profile as prof
            with prof(activities=[ProfilerActivity.CUDA]) as p:
                for _ in range(3):
                    ce.module(ids, types, mask)
                torch.cuda.synchronize()
            print(p.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=90), flush=True)
    if tunable:
        torch.cuda.tunable.enable(False)


if os.environ.get("PROBE_NO_RECORDED") == "1":   # tune every shape from scratch: the shipped solutions are not loaded
    import advanced_rag.encoders as _enc
    _enc._TUNED_GEMMS = False
ce = CrossEncoderModel(EncoderConfig(gelu="tanh"), device=str(dev), max_len=512)
print("recorded hipBLASLt solutions in use:", ce.tuned_gemms, flush=True)
run("tanh gelu epilogue + HIP attention + recorded GEMM solutions", ce, profile=True)
if os.environ.get("PROBE_TUNE") == "1":   # re-tune online (writes gpurun_out/tunableop_ce.csv): how the shipped file was made
    torch.cuda.tunable.tuning_enable(True)
    run("... re-tuned online", ce, tunable=True)


def run_chunked(chunk):
    """The same forward, `chunk` sequences at a time through ALL layers: the activations of a chunk (chunk x T x (384 + 1152
    + 1536) halves) stay in the 256 MiB Infinity Cache between the kernels that write and read them."""
    with torch.inference_mode():
        parts = [(ids[a:a + chunk], types[a:a + chunk], mask[a:a + chunk]) for a in range(0, P, chunk)]
        t0 = time.time()
        for _ in range(2):
            for p in parts:
                ce.module(*p)
        torch.cuda.synchronize()
        warm = time.time() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            for p in parts:
                ce.module(*p)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = P * ce.flops_per_pair(T) / (ms * 1e-3) / 1e12
        mb = chunk * T * (384 + 1152 + 1536 + 384) * 2 / 1e6
        print(f"chunks of {chunk:5d} sequences ({mb:6.1f} MB live) {ms:8.3f} ms/forward  {tf:7.1f} TFLOP/s = {tf / 2500:.3f} (warm-up {warm:.1f}s)", flush=True)


if os.environ.get("PROBE_CHUNKS"):   # e.g. PROBE_CHUNKS=160,320,640,1280 ; tunes the chunk-sized GEMM shapes online first
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(150)
    torch.cuda.tunable.set_filename(os.path.join(ROOT, "gpurun_out", "tunableop_ce_chunks.csv"))
    for c in os.environ["PROBE_CHUNKS"].split(","):
        run_chunked(int(c))
