"""Ad-hoc probe (not a test): dense search at B in {64,128,256}, big-batch pass vs multi-pass, + parity on a small corpus."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, "advanced-rag-milvus_amd"); sys.path.insert(0, ".")
from advanced_rag import _native as nat
import oracle
nat.debug_option(nat.HR_DEBUG_DENSE_KERNELS, int(os.environ.get('PROBE_DENSE_MASK', '0')))
nat.debug_option(nat.HR_DEBUG_GROUP_ROWS, int(os.environ.get('PROBE_GROUP_ROWS', '0')))   # 16 / 64 rows per candidate group (0 = by shard size)
   # e.g. 16 = the 4 x 64-query register form

# parity first (small corpus, B=200 -> GQ=16; B=100 -> GQ=8)
rng = np.random.default_rng(0)
for dt, npdt in (() if os.environ.get('SKIP_PARITY') else ((nat.HR_F16, np.float16), (nat.HR_F32, np.float32))):
    for n, D in (((3333, 768), (70001, 768)) if os.environ.get('PARITY_768') else ((5000, 256), (70000, 256), (3333, 768), (70001, 768), (40000, 1024), (250000, 768))):
        X = rng.standard_normal((n, D)).astype(np.float32).astype(npdt)
        X[n // 2] = X[7]  # a tie
        h = nat.ShardHandle(D, dt, nat.HR_METRIC_COSINE); h.add_dense(X); h.finalize()
        mask = np.packbits(rng.random(n) < 0.7, bitorder="little")
        for B in (100, 128, 200, 300):
            Q = rng.standard_normal((B, D)).astype(np.float32)
            Q[1] = X[7].astype(np.float32)
            ok = True
            for m in (None, mask):
                ids, sc = h.search_dense(Q, 40, m)
                sel = [0, 1, 2, B // 2, B - 2, B - 1]
                oids, osc = oracle.dense_search(X, Q[sel], 40, oracle.COSINE, mask=m)
                ok = ok and np.array_equal(ids[sel], oids) and np.array_equal(sc[sel].view(np.uint32), osc.view(np.uint32))
            print("parity", "f16" if dt == nat.HR_F16 else "f32", n, D, B, ok, flush=True)
        h.close()

N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda:0")
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE); h.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for r0 in range(0, N, 500_000):
    n = min(500_000, N - r0)
    x = torch.randn((n, D), device=dev, generator=g, dtype=torch.float32).to(torch.float16); torch.cuda.synchronize()
    if os.environ.get("PROBE_ZERO"):   # power experiment: one nonzero element per row, the MFMAs multiply zeros
        x.zero_(); x[:, 0] = 1
    h.add_dense_dev(x.data_ptr(), n)
h.finalize(); h.set_profiling(2)
st = torch.cuda.current_stream().cuda_stream
for B in (128, 256):
    q = torch.randn((B, D), device=dev, generator=g)
    ids = torch.empty((B, 40), dtype=torch.int64, device=dev); sc = torch.empty((B, 40), dtype=torch.float32, device=dev)
    fl = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(3): h.search_dense_dev(q.data_ptr(), B, 40, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
    torch.cuda.synchronize(); h.kernel_ms()
    t0 = time.time()
    for _ in range(10): h.search_dense_dev(q.data_ptr(), B, 40, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
    torch.cuda.synchronize(); dt_ = (time.time() - t0) / 10
    ms = h.kernel_ms(); scan, launches = ms["dense_scan"]
    print(f"B={B:3d} total {dt_*1e3:7.3f} ms QPS {B/dt_:9.0f} scan {scan:7.3f} ms x{launches//10} launches = {h.dense_scan_bytes/scan/1e6:7.1f} GB/s/launch | "
          f"gsel {ms['group_select'][0]:.3f} refine {ms['refine'][0]:.3f} topk {ms['topk'][0]:.3f} exact {int(fl.sum())}/{B}", flush=True)
