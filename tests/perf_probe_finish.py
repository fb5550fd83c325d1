"""Ad-hoc probe (not a test): where a block of the fused finishing kernel spends its time, alone on an idle chip.
    make -C advanced-rag-milvus_amd stamp
    HBMRAG_LIB=advanced-rag-milvus_amd/lib/libhbmrag_stamp.so python tests/perf_probe_finish.py [rows=1250000]
s_memtime ticks are shader cycles here (~2.1 GHz under load).  Block (query 0, modality m) of the last launch."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from advanced_rag import _native as nat  # noqa: E402
from advanced_rag.engine import pack_sparse_queries  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
D, B, BLK, kp = 768, 128, 250_000, 40
dev = torch.device("cuda:0")
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, bench.SPARSE_DIM, 0)
h.reserve(N)
for b in range(-(-N // BLK)):
    n = min(BLK, N - b * BLK)
    h.add_dense(bench.dense_block(b, n, D))
    h.add_sparse(*bench.sparse_block(b, n))
h.finalize()
Q, SQ = bench.make_queries(1, B, D)
q = torch.from_numpy(Q[0]).to(dev)
ptr, idx, val, mx = pack_sparse_queries(SQ[0], 0.2)
dp, di, dv = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev)
ids = torch.empty((2, B, kp), dtype=torch.int64, device=dev)
sc = torch.empty((2, B, kp), dtype=torch.float32, device=dev)
fl = torch.empty((2, B), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
h.hybrid_scan_dev(q.data_ptr(), dp.data_ptr(), di.data_ptr(), dv.data_ptr(), B, len(idx), mx, kp, 0, st)
L = nat.load_library()
fn = getattr(L, "hr_debug_finish_stamps", None)


def stamps():
    if fn is None:
        print("not a stamp build: no phase shares")
        return
    buf = (ctypes.c_ulonglong * 16)()
    fn(buf)
    v = np.array(list(buf), dtype=np.int64).reshape(2, 8)
    for m, label in ((0, "dense block"), (1, "sparse block")):
        d = np.diff(v[m][:5]) * 1e-3
        print(f"  {label}: {v[m][5]} candidate rows after the trim, {v[m][6]} through the canonical chain")
        print(f"  {label}: bucket maxima {d[0]:.1f}, group select {d[1]:.1f}, refine {d[2]:.1f}, top-k {d[3]:.1f}, total {d.sum():.1f} (k shader cycles)")


for mode, name in ((1, "chain"), (2, "fused")):
    nat.debug_option(nat.HR_DEBUG_FINISH_MODE, mode)

    for _ in range(3):
        h.hybrid_finish_dev(q.data_ptr(), dp.data_ptr(), di.data_ptr(), dv.data_ptr(), B, mx, kp, 0, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        h.hybrid_finish_dev(q.data_ptr(), dp.data_ptr(), di.data_ptr(), dv.data_ptr(), B, mx, kp, 0, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), st)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per finish alone (B = {B}, rows = {N}); all proven: {int(fl.min()) == 1}")
    if mode == 2:
        stamps()
