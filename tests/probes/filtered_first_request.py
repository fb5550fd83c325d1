"""Ad-hoc probe: what the FIRST filtered retrieve() of an expression costs, piece by piece (2M-row synthetic shard)."""
import asyncio
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from advanced_rag import _native as nat  # noqa: E402
from advanced_rag.constants import RetrievalConstants  # noqa: E402
from advanced_rag.indexing import MilvusIndexManager  # noqa: E402
from advanced_rag.retrieval import HybridRetriever, RetrievalConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
D, BLK = 768, 250_000
h = nat.ShardHandle(D, nat.HR_F16, nat.HR_METRIC_COSINE, bench.SPARSE_DIM, 0)
h.reserve(N)
for b in range(-(-N // BLK)):
    n = min(BLK, N - b * BLK)
    h.add_dense(bench.dense_block(b, n, D))
    h.add_sparse(*bench.sparse_block(b, n))
h.finalize()
Q, SQ = bench.make_queries(1, 64, D)
mgr = MilvusIndexManager(semantic_dim=D, sparse_dim=bench.SPARSE_DIM, connect=False)
mgr.attach_shards([h], synthetic_rows=N)


class Gen:
    def encode_semantic(self, text):
        return Q[0][int(text[1:])]

    def encode_sparse(self, text):
        qi, qv = SQ[0][int(text[1:])]
        return {"indices": qi.tolist(), "values": qv.tolist()}


mgr.embedding_generator = Gen()
RetrievalConstants.TIMEOUT_SECONDS = 60.0
retr = HybridRetriever(mgr, RetrievalConfig(top_k=20))


async def timed(q, **kw):
    t0 = time.perf_counter()
    out = await retr.retrieve(q, profile_hint="default", **kw)
    return (time.perf_counter() - t0) * 1e3, len(out)


async def main():
    for i in range(20):
        await timed(f"q{i}")
    print("unfiltered:", [round((await timed(f"q{i}"))[0], 2) for i in range(5)], flush=True)
    for lim in (5, 3, 7):
        st0 = dict(mgr._front.stats)
        first = await timed("q0", filters={"chunk_index": {"$lt": lim}})
        st1 = dict(mgr._front.stats)
        rest = [round((await timed(f"q{i}", filters={"chunk_index": {"$lt": lim}}))[0], 2) for i in range(1, 5)]
        print(f"chunk_index < {lim}: first {first[0]:.2f} ms ({first[1]} hits), then {rest}; front stats delta",
              {k: st1[k] - st0[k] for k in st1 if st1[k] != st0[k]}, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mgr._global_device_mask("chunk_index < 2")
    torch.cuda.synchronize()
    print("mask eval of a new expression alone:", round((time.perf_counter() - t0) * 1e3, 3), "ms")

asyncio.run(main())
asyncio.run(mgr.close())
h.close()
