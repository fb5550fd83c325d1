import asyncio, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
import g5_data, oracle
from advanced_rag import MilvusIndexManager
g, X, csr, Q, SQ = g5_data.inputs()
mgr = MilvusIndexManager(semantic_dim=384, sparse_dim=g5_data.SPARSE_DIM, dtype="float16", enable_domain=False)
mgr.add_rows(X, csr, ids=[g5_data.row_id(r) for r in range(1000)], contents=["x"] * 1000)
mgr.finalize()
X16 = X.astype(np.float16)
for q in range(8):
    for kp in (60, 80):
        hits = asyncio.run(mgr.search(Q[q], "semantic_index", kp))
        dev = [int(h["id"].rsplit("::", 1)[1], 16) for h in hits]
        di, ds = oracle.dense_search(X16, Q[q:q + 1], kp, oracle.COSINE)
        if dev != di[0].tolist():
            bad = [(i, a, b, hits[i]["score"], float(ds[0][i])) for i, (a, b) in enumerate(zip(dev, di[0].tolist())) if a != b]
            print("q", q, "kp", kp, "DIFF", bad[:6])
        else:
            same_sc = all(np.float32(h["score"]) == s for h, s in zip(hits, ds[0]))
            print("q", q, "kp", kp, "same ids; scores equal:", same_sc)
asyncio.run(mgr.close())

# ---- the failing case: retrieve under the troubleshooting profile, query 0, fp16 rows
from advanced_rag import HybridRetriever, RetrievalConfig
from advanced_rag.constants import RetrievalConstants
from advanced_rag.embedding_cache import initialize_caches
RetrievalConstants.TIMEOUT_SECONDS = 60.0
mgr = MilvusIndexManager(semantic_dim=384, sparse_dim=g5_data.SPARSE_DIM, dtype="float16", enable_domain=False)
mgr.add_rows(X, csr, ids=[g5_data.row_id(r) for r in range(1000)], contents=[g5_data.mmr_content(r) for r in range(1000)])
mgr.finalize()
class Gen:
    def encode_semantic(self, t): return Q[0]
    def encode_sparse(self, t): return {"indices": SQ[0][0].tolist(), "values": SQ[0][1].astype(float).tolist()}
mgr.embedding_generator = Gen()
initialize_caches()
retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
out = asyncio.run(retr.retrieve("plain statement", profile_hint="troubleshooting"))
kp, top_k = 60, 30
di, _ = oracle.dense_search(X16, Q[0:1], kp, oracle.COSINE)
si, _ = oracle.sparse_search(csr[0], csr[1], csr[2], [SQ[0]], kp, 0.2)
sl = si[0][si[0] >= 0]
ids, scores, methods = oracle.rrf(di[0], sl, (), 0.7, 0.3, 0.2, 60)
sel = oracle.mmr(ids, [float(x) for x in scores], [g5_data.mmr_content(int(r)) for r in ids], top_k, 0.5)[:top_k]
dev_rows = [int(o["id"].rsplit("::", 1)[1], 16) for o in out]
print("device rows", dev_rows)
print("oracle rows", [int(ids[i]) for i in sel])
for pos, (o, i) in enumerate(zip(out, sel)):
    if float(o["score"]) != float(scores[i]):
        r = dev_rows[pos]
        print("pos", pos, "row", r, "device", o["score"], o["retrieval_methods"], "oracle", float(scores[i]), "dense rank", list(di[0]).index(r) + 1 if r in di[0] else None,
              "sparse rank", list(sl).index(r) + 1 if r in sl else None)
dl = asyncio.run(mgr.search(Q[0], "semantic_index", 60))
spl = asyncio.run(mgr.search({"indices": SQ[0][0].tolist(), "values": SQ[0][1].astype(float).tolist()}, "sparse_index", 60, search_params={"metric_type": "IP", "params": {"drop_ratio_search": 0.2}}))
print("dense equal", [int(h["id"].rsplit("::", 1)[1], 16) for h in dl] == di[0].tolist(), "sparse equal", [int(h["id"].rsplit("::", 1)[1], 16) for h in spl] == sl.tolist(), len(spl), len(sl))
asyncio.run(mgr.close())
