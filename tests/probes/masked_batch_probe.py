"""Diagnostic: device-form searches with a row mask at many batch sizes against the host forms (which the oracle tests pin)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
import torch
from advanced_rag import _native as nat
from advanced_rag.engine import EngineConfig, HybridSearchEngine, pack_sparse_queries
rng = np.random.default_rng(3)
n, d, V, nq = 50000, 96, 2000, 128
X = rng.standard_normal((n, d)).astype(np.float32)
X[40000] = X[17]
idx = np.sort(np.argpartition(rng.random((n, V)), 19, axis=1)[:, :20], axis=1).astype(np.int32).reshape(-1)
val = np.abs(rng.standard_normal(n * 20)).astype(np.float32)
ptr = np.arange(n + 1, dtype=np.int64) * 20
Q = rng.standard_normal((nq, d)).astype(np.float32)
Q[5] = X[17]
SQ = [(np.sort(rng.choice(V, 40, replace=False)).astype(np.int32), np.abs(rng.standard_normal(40)).astype(np.float32)) for _ in range(nq)]
h = nat.ShardHandle(d, nat.HR_F16, nat.HR_METRIC_COSINE, V)
h.add_dense(X); h.add_sparse(ptr, idx, val); h.finalize()
keep = (np.arange(n) % 10) < 5
mask_bits = np.packbits(keep, bitorder="little")
pad = np.zeros(((n + 63) // 64) * 8, np.uint8); pad[:mask_bits.size] = mask_bits
dmask = torch.from_numpy(pad).cuda()
kp = 40
hd_ids, hd_sc = h.search_dense(Q, kp, pad)
hs_ids, hs_sc = h.search_sparse(SQ, kp, 0.2, pad)
s = torch.cuda.current_stream().cuda_stream
bad = 0
eng = HybridSearchEngine(h, EngineConfig(top_k=20, enable_reranking=False))
for B in (1, 2, 3, 5, 8, 13, 16, 17, 31, 32, 33, 40, 48, 63, 64, 65, 96, 100, 127, 128):
    for rep in range(3):
        sel = rng.permutation(nq)[:B]
        q = torch.from_numpy(Q[sel]).cuda()
        ids = torch.empty((B, kp), dtype=torch.int64, device="cuda"); sc = torch.empty((B, kp), dtype=torch.float32, device="cuda")
        fl = torch.zeros((B,), dtype=torch.int32, device="cuda")
        h.search_dense_dev(q.data_ptr(), B, kp, ids.data_ptr(), sc.data_ptr(), fl.data_ptr(), dmask.data_ptr(), s)
        p, si, sv, mx = pack_sparse_queries([SQ[i] for i in sel], 0.2, V)
        dp, dsi, dsv = torch.from_numpy(p).cuda(), torch.from_numpy(si).cuda(), torch.from_numpy(sv).cuda()
        ids2 = torch.empty((B, kp), dtype=torch.int64, device="cuda"); sc2 = torch.empty((B, kp), dtype=torch.float32, device="cuda")
        fl2 = torch.zeros((B,), dtype=torch.int32, device="cuda")
        h.search_sparse_dev(dp.data_ptr(), dsi.data_ptr(), dsv.data_ptr(), B, int(si.shape[0]), int(mx), kp, ids2.data_ptr(), sc2.data_ptr(), fl2.data_ptr(), dmask.data_ptr(), s)
        out = eng.search(q, (dp, dsi, dsv, int(mx)), rowmask=dmask)
        torch.cuda.synchronize()
        f1, f2 = fl.cpu().numpy(), fl2.cpu().numpy()
        e_ids, e_fl = out["ids"].cpu().numpy(), out["flags"].cpu().numpy()
        for j, qi in enumerate(sel):
            if f1[j] == 1 and not (np.array_equal(ids[j].cpu().numpy(), hd_ids[qi]) and np.array_equal(sc[j].cpu().numpy(), hd_sc[qi])):
                bad += 1; print("DENSE mismatch B", B, "query", qi, "proven", f1[j])
            if f2[j] == 1 and not (np.array_equal(ids2[j].cpu().numpy(), hs_ids[qi]) and np.array_equal(sc2[j].cpu().numpy(), hs_sc[qi])):
                bad += 1; print("SPARSE mismatch B", B, "query", qi, "proven", f2[j])
            if e_fl[0][j] == 1 and not np.array_equal(e_ids[0][j], hd_ids[qi]):
                bad += 1; print("HYBRID dense-list mismatch B", B, "query", qi)
            if e_fl[1][j] == 1 and not np.array_equal(e_ids[1][j], hs_ids[qi]):
                bad += 1; print("HYBRID sparse-list mismatch B", B, "query", qi)
    print("B", B, "done; unproven dense/sparse", int((f1 != 1).sum()), int((f2 != 1).sum()), flush=True)
print("mismatches:", bad)
