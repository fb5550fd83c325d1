"""Ad-hoc probe (not a test): where a single-query retrieve() spends its time at 10M x 768."""
import asyncio, sys, time
import numpy as np
import torch
sys.path.insert(0, "advanced-rag-milvus_amd"); sys.path.insert(0, ".")
from advanced_rag import _native as nat
from advanced_rag.constants import RetrievalConstants
from advanced_rag.indexing import MilvusIndexManager, ShardCollection
from advanced_rag.retrieval import HybridRetriever, RetrievalConfig
from bench import sparse_block, SPARSE_DIM, SPARSE_NNZ

N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 768
dev = torch.device("cuda:0")
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, SPARSE_DIM); h.reserve(N)
g = torch.Generator(device=dev); g.manual_seed(1)
for b in range(N // 500_000):
    x = torch.randn((500_000, D), device=dev, generator=g, dtype=torch.float32).to(torch.float16); torch.cuda.synchronize()
    h.add_dense_dev(x.data_ptr(), 500_000)
    h.add_sparse(*sparse_block(b, 500_000))
h.finalize()
rng = np.random.default_rng(0)
Q = rng.standard_normal((64, D)).astype(np.float32)
_, idx, val = sparse_block(0, 64, seed=9)
SQ = [(idx[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ], val[i * SPARSE_NNZ:(i + 1) * SPARSE_NNZ]) for i in range(64)]

def t(fn, n=30):
    for _ in range(5): fn()
    xs = []
    for i in range(n):
        t0 = time.perf_counter(); fn(); xs.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(xs))

print("host-form dense  B=1 k=40: %.3f ms" % t(lambda: h.search_dense(Q[:1], 40)))
print("host-form sparse B=1 k=40: %.3f ms" % t(lambda: h.search_sparse(SQ[:1], 40, 0.2)))
mgr = MilvusIndexManager(semantic_dim=D, sparse_dim=SPARSE_DIM, connect=False)
mgr.attach_shards([h], synthetic_rows=h.num_rows)
class Gen:
    def encode_semantic(self, text): return Q[int(text[1:])]
    def encode_sparse(self, text):
        qi, qv = SQ[int(text[1:])]; return {"indices": qi.tolist(), "values": qv.tolist()}
mgr.embedding_generator = Gen()
RetrievalConstants.TIMEOUT_SECONDS = 60.0
retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))
loop = asyncio.new_event_loop()
run = lambda coro: loop.run_until_complete(coro)
i = [0]
def one():
    i[0] += 1
    return run(retr.retrieve(f"q{i[0] % 64}", profile_hint="default"))
print("retrieve(): %.3f ms" % t(one, 60))
print("  mgr.search dense : %.3f ms" % t(lambda: run(mgr.search(Q[0], "semantic_index", 40))))
print("  mgr.search sparse: %.3f ms" % t(lambda: run(mgr.search({"indices": SQ[0][0].tolist(), "values": SQ[0][1].tolist()}, "sparse_index", 40))))
print("  embeddings       : %.3f ms" % t(lambda: (run(mgr._generate_semantic_embedding("q1")), run(mgr._generate_sparse_embedding("q1")))))
a = run(mgr.search(Q[0], "semantic_index", 40)); b = run(mgr.search({"indices": SQ[0][0].tolist(), "values": SQ[0][1].tolist()}, "sparse_index", 40))
print("  fuse             : %.3f ms" % t(lambda: retr._fuse_results([dict(x) for x in a], [dict(x) for x in b], [])))
