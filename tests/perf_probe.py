"""Ad-hoc perf probe (not a test): dense scan at a given shape with device-generated data."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "advanced-rag-milvus_amd")
from advanced_rag import _native as nat

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda:0")
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE)
h.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
t0 = time.time()
blk = 500_000
for r0 in range(0, N, blk):
    n = min(blk, N - r0)
    x = torch.randn((n, D), device=dev, generator=g, dtype=torch.float32).to(torch.float16)
    torch.cuda.synchronize()
    h.add_dense_dev(x.data_ptr(), n)
h.finalize()
print(f"ingest {time.time()-t0:.1f}s rows={h.num_rows} bytes={h.device_bytes/1e9:.2f}GB", flush=True)
h.set_profiling(2)
st = torch.cuda.current_stream().cuda_stream
for B in (1, 8, 16, 32, 64):
    q = torch.randn((B, D), device=dev, generator=g)
    ids = torch.empty((B, 40), dtype=torch.int64, device=dev)
    sc = torch.empty((B, 40), dtype=torch.float32, device=dev)
    fl = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(3):
        h.search_dense_dev(q.data_ptr(), B, 40, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
    torch.cuda.synchronize(); h.kernel_ms()
    t0 = time.time()
    it = 10
    for _ in range(it):
        h.search_dense_dev(q.data_ptr(), B, 40, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / it
    ms = h.kernel_ms()
    scan = ms["dense_scan"][0]
    print(f"B={B:3d} total {dt*1e3:7.3f} ms  QPS {B/dt:9.0f}  scan {scan:7.3f} ms = {h.dense_scan_bytes/scan/1e9:7.1f} GB/s "
          f"| prep {ms['prep'][0]:.3f} gsel {ms['group_select'][0]:.3f} refine {ms['refine'][0]:.3f} topk {ms['topk'][0]:.3f} "
          f"| exact {int(fl.sum())}/{B}", flush=True)
