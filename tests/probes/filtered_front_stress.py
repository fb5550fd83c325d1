"""Diagnostic: concurrent filtered + plain retrieve() through the two-round front, many times; reports which list of which query
differs from the sequential answer."""
import asyncio, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag import HybridRetriever, MilvusIndexManager, RetrievalConfig
from advanced_rag.constants import RetrievalConstants
from advanced_rag.embedding_cache import initialize_caches
RetrievalConstants.TIMEOUT_SECONDS = 120.0
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
class TableGen:
    def encode_semantic(self, text): return Q[int(text[1:])]
    def encode_sparse(self, text):
        qi, qv = SQ[int(text[1:])]
        return {"indices": qi.tolist(), "values": qv.tolist()}
    def encode_domain(self, text, domain=None): return np.zeros(8, np.float32)
F = {"chunk_index": {"$lt": 5}}
expr = "chunk_index < 5"
def mk(coalesce):
    initialize_caches()
    m = MilvusIndexManager(semantic_dim=d, sparse_dim=V, dtype="float16", enable_domain=False, coalesce=coalesce)
    m.add_rows(X, (ptr, idx, val), chunk_index=(np.arange(n) % 10).tolist())
    m.finalize()
    m.embedding_generator = TableGen()
    m.hybrid_search = None        # two rounds: searches, then fusion
    return m
ref_m = mk(False)
sparse_params = {"metric_type": "IP", "params": {"drop_ratio_search": 0.2}}
async def lists(m, i, flt):
    dl = await m.search(Q[i], "semantic_index", 40, flt)
    sl = await m.search({"indices": SQ[i][0].tolist(), "values": SQ[i][1].tolist()}, "sparse_index", 40, flt, sparse_params)
    return [(h["id"], h["score"]) for h in dl], [(h["id"], h["score"]) for h in sl]
ref = {(i, f): asyncio.run(lists(ref_m, i, expr if f else None)) for i in range(nq) for f in (0, 1)}
m = mk(True)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    async def go():
        jobs = [(i, 0) for i in range(nq)] + [(i, 1) for i in range(0, nq, 4)]
        outs = await asyncio.gather(*[lists(m, i, expr if f else None) for i, f in jobs])
        return jobs, outs
    jobs, outs = asyncio.run(go())
    for (i, f), (dl, sl) in zip(jobs, outs):
        rd, rs = ref[(i, f)]
        if dl != rd or sl != rs:
            bad += 1
            which = "dense" if dl != rd else "sparse"
            a, b = (dl, rd) if dl != rd else (sl, rs)
            pos = next((k for k, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
            print(f"iter {it}: query {i} filtered={f} {which} list differs at {pos}: got {a[pos:pos+2]} want {b[pos:pos+2]} (len {len(a)}/{len(b)})", flush=True)
    st = m._front.stats
# ---- the same through retrieve() (searches + the fusion round), as the test does
def strip(hits):
    return [(h["id"], float(h["score"]).hex(), tuple(h["retrieval_methods"]), h["method"], float(h["original_score"]).hex()) for h in hits]
retr_ref = HybridRetriever(ref_m, RetrievalConfig(top_k=20))
retr = HybridRetriever(m, RetrievalConfig(top_k=20))
want = {(i, f): strip(asyncio.run(retr_ref.retrieve(f"q{i}", filters=F if f else None, profile_hint="default"))) for i in range(nq) for f in (0, 1)}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    async def go2():
        jobs = [(i, 0) for i in range(nq)] + [(i, 1) for i in range(0, nq, 4)]
        outs = await asyncio.gather(*[retr.retrieve(f"q{i}", filters=F if f else None, profile_hint="default") for i, f in jobs])
        return jobs, outs
    jobs, outs = asyncio.run(go2())
    for (i, f), o in zip(jobs, outs):
        got = strip(o)
        if got != want[(i, f)]:
            bad += 1
            pos = next((k for k, (x, y) in enumerate(zip(got, want[(i, f)])) if x != y), -1)
            print(f"retrieve iter {it}: query {i} filtered={f} differs at {pos}: got {got[pos:pos+1]} want {want[(i, f)][pos:pos+1]} (len {len(got)}/{len(want[(i, f)])})", flush=True)
st = m._front.stats
print("iterations done; mismatching lists:", bad, "front stats", {k: v for k, v in st.items() if k != "busy_s"})
asyncio.run(m.close()); asyncio.run(ref_m.close())
