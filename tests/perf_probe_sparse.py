"""Ad-hoc probe (not a test): phase breakdown of sparse_scan_kernel from a -DHR_TRACE build
(HBMRAG_LIB=advanced-rag-milvus_amd/lib/libhbmrag_trace.so python tests/perf_probe_sparse.py [docs] [B])."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, "advanced-rag-milvus_amd")
sys.path.insert(0, ".")
from advanced_rag import _native as nat
from advanced_rag.engine import pack_sparse_queries
from bench import sparse_block, sparse_block_zipf, zipf_queries, SPARSE_DIM, SPARSE_NNZ

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ZIPF = len(sys.argv) > 3 and sys.argv[3] == "zipf"   # Zipf(1.1) postings + 20-term queries (bench.py --sparse-dist zipf)
dev = torch.device("cuda:0")
h = nat.ShardHandle(64, nat.HR_F16, nat.HR_METRIC_COSINE, SPARSE_DIM)
blk = 250_000
for b in range(N // blk):
    h.add_dense(np.zeros((blk, 64), np.float16) + 1)
    h.add_sparse(*(sparse_block_zipf(b, blk) if ZIPF else sparse_block(b, blk)))
h.finalize()
L = nat.load_library()
trace = getattr(L, "hr_debug_trace", None)
rng = np.random.default_rng(3)
_, idx, val = sparse_block(0, B, seed=77)
sq = [(idx[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ], val[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ]) for i in range(B)]
if ZIPF:
    sq = zipf_queries(rng, B)
ptr, qi, qv, mx = pack_sparse_queries(sq, 0.2)
d_ptr, d_i, d_v = (torch.from_numpy(a).to(dev) for a in (ptr, qi, qv))
ids = torch.empty((B, 40), dtype=torch.int64, device=dev)
sc = torch.empty((B, 40), dtype=torch.float32, device=dev)
fl = torch.empty((B,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
h.set_profiling(2)
def run():
    h.search_sparse_dev(d_ptr.data_ptr(), d_i.data_ptr(), d_v.data_ptr(), B, int(qi.shape[0]), int(mx), 40,
                        ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), 0, st)
for _ in range(3):
    run()
torch.cuda.synchronize(); h.kernel_ms()
if trace is not None:
    buf = (ctypes.c_ulonglong * 16)()
    trace(buf, 1)
it = 10
for _ in range(it):
    run()
torch.cuda.synchronize()
ms = h.kernel_ms()
print(f"docs {N} B {B}: sparse_scan {ms['sparse_scan'][0]:.3f} ms  exact {int(fl.sum())}/{B}")
if trace is not None:
    trace(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    nb = v[15]
    names = ["zero acc", "hop1 q terms", "hop2 run bounds", "item table (2 barriers)", "postings issue+arrive (all sweeps)",
             "atomics drain", "barrier", "group max + store"]
    tot = v[:8].sum()
    for n_, x in zip(names, v[:8]):
        print(f"  {n_:36s} {x / nb:9.1f} ticks/block  {100 * x / tot:5.1f}%")
    print(f"  total {tot / nb:.1f} ticks/block over {int(nb)} blocks")
