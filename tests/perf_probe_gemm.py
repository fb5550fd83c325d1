"""Ad-hoc probe (not a test): where a k-step of dense_scan_gemm_kernel spends its cycles.
    make -C advanced-rag-milvus_amd stamp
    HBMRAG_LIB=advanced-rag-milvus_amd/lib/libhbmrag_stamp.so python tests/perf_probe_gemm.py [rows] [dim]
Shares only: the stamps forbid overlaps the real kernel has (its run time is not a measurement)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, "advanced-rag-milvus_amd"); sys.path.insert(0, ".")
from advanced_rag import _native as nat

N, D, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 256
dev = torch.device("cuda:0")
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE); h.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for r0 in range(0, N, 500_000):
    n = min(500_000, N - r0)
    x = torch.randn((n, D), device=dev, generator=g, dtype=torch.float32).to(torch.float16); torch.cuda.synchronize()
    h.add_dense_dev(x.data_ptr(), n)
h.finalize(); h.set_profiling(2)
q = torch.randn((B, D), device=dev, generator=g)
ids = torch.empty((B, 40), dtype=torch.int64, device=dev); sc = torch.empty((B, 40), dtype=torch.float32, device=dev)
fl = torch.empty((B,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    h.search_dense_dev(q.data_ptr(), B, 40, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
torch.cuda.synchronize()
print("scan ms (stamped build: not a measurement)", h.kernel_ms()["dense_scan"])
L = nat.load_library()
fn = getattr(L, "hr_debug_gemm_stamps", None)
if fn is None:
    sys.exit("not a stamp build")
buf = (ctypes.c_ulonglong * 16)()
fn(buf)
v = np.array(list(buf), dtype=np.float64).reshape(2, 8)
names = ["reads of this step returned (settle)", "refill issue (+ leading half: own loads of next step)",
         "barrier before the MFMAs", "epilogue", "32 MFMAs + 12 reads of the next step (issue)",
         "trailing half: own loads of the step after next", "barrier after the MFMAs", "bookkeeping after the MFMAs"]
steps = -(-(N // 256) // 256) * (D // 32)
for w, label in ((0, "wave 0 (corpus loader)"), (1, "wave 5 (query loader)")):
    tot = v[w].sum()
    print(label, f"total {tot / steps:.0f} cycles/step over ~{steps} steps")
    for n_, x in zip(names, v[w]):
        print(f"   {n_:34s} {x / steps:8.1f} cycles/step {100 * x / tot:5.1f}%")
