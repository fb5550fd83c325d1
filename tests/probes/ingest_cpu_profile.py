"""Ad-hoc probe (not a test, no GPU): where the HOST time of `AdvancedRAGPipeline.ingest_documents()` goes.  The shard is a
stand-in that counts rows, the encoder is the real SentenceEncoder host side (hash tokenizer, BM25 payloads) around a
one-layer CPU model, the documents are bench.py --ingest's (~512 tokens, Zipfian vocabulary).

    python tests/probes/ingest_cpu_profile.py [docs=512] [profile=1]
"""
import asyncio
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "advanced-rag-milvus_amd"))
from advanced_rag import AdvancedRAGPipeline, BM25SparseEncoder, PipelineConfig  # noqa: E402
from advanced_rag.embedding_cache import initialize_caches  # noqa: E402
from advanced_rag.encoders import EncoderConfig, SentenceEncoder  # noqa: E402

N_DOCS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PROFILE = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
DIM, SPARSE_DIM = 32, 10000


class CountingShard:
    dim, sparse_dim, device, dtype = DIM, SPARSE_DIM, 0, 0
    num_rows = num_sparse_rows = 0

    def add_dense(self, rows):
        self.num_rows += rows.shape[0]

    def add_sparse(self, ptr, idx, val):
        self.num_sparse_rows += len(ptr) - 1

    def finalize(self):
        pass

    def close(self):
        pass


rng = np.random.default_rng(99)
vocab = [f"w{i}" for i in range(20000)]
zipf = 1.0 / np.arange(1, len(vocab) + 1) ** 1.05
zipf /= zipf.sum()


def make_doc(i):
    words = rng.choice(len(vocab), size=512, p=zipf)
    sents = [" ".join(vocab[w] for w in words[j:j + 16]).capitalize() + "." for j in range(0, 512, 16)]
    return {"id": f"doc{i}", "text": " ".join(sents), "metadata": {"source": "bench"}}


docs = [make_doc(i) for i in range(N_DOCS)]
bm25 = BM25SparseEncoder(sparse_dim=SPARSE_DIM).fit(d["text"] for d in docs)
enc = SentenceEncoder(EncoderConfig(hidden=DIM, layers=1, heads=2, intermediate=64), device="cpu", sparse_encoder=bm25, max_len=256,
                      batch_size=128)


class HostHop:
    encode_semantic = lambda self, t: enc.encode_semantic(t)
    encode_semantic_batch = lambda self, ts: enc.encode_semantic_batch(ts)
    encode_sparse = lambda self, t: enc.encode_sparse(t)
    encode_sparse_query = lambda self, t: enc.encode_sparse_query(t)


initialize_caches()
pipe = AdvancedRAGPipeline(connect_to_milvus=False, config=PipelineConfig(enable_audit_logging=False), semantic_dim=DIM,
                           sparse_dim=SPARSE_DIM, enable_domain=False, dtype="float16")
mgr = pipe.index_manager
mgr._native = None
mgr.attach_shards([CountingShard()], rows_of=[np.zeros(0, np.int64)])
mgr.embedding_generator = HostHop()


async def run():
    tm = {}
    for b0 in range(0, len(docs), 128):
        rep = await pipe.ingest_documents(docs[b0:b0 + 128])
        assert not rep["indexing_summary"]["errors"], rep["indexing_summary"]["errors"][:2]
        for k, v in rep["indexing_summary"]["timing_ms"].items():
            tm[k] = tm.get(k, 0.0) + v
    return tm

asyncio.run(pipe.ingest_documents(docs[:16]))
prof = cProfile.Profile() if PROFILE else None
t0, c0 = time.perf_counter(), time.process_time()
if prof:
    prof.enable()
tm = asyncio.run(run())
if prof:
    prof.disable()
wall, cpu = time.perf_counter() - t0, time.process_time() - c0
print(f"{N_DOCS} docs: {N_DOCS / wall:.0f} docs/s, {cpu / N_DOCS * 1e6:.0f} us of process CPU per doc; timing_ms {({k: round(v) for k, v in tm.items()})}")
if prof:
    pstats.Stats(prof).sort_stats("cumulative").print_stats(40)
