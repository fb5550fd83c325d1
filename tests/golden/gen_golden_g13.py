"""Golden fixture G13 from the REFERENCE implementation (container-only): HybridRetriever.retrieve end to end at the size of
BASELINE config 2 — 100 000 x 384 fp32 — dense-only and hybrid, 8 queries, over an exact numpy FLAT stand-in for the Milvus
collections (the same stand-in as g5, gen_golden.py:159-197).  g5 pins the chain at 1 000 rows; this one pins it where
candidate groups, range boundaries of the sparse postings (16 384 docs) and several scan blocks per query are in play.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_g13.py

Imports the reference's logic modules through an empty parent package (as its own tests do); writes
tests/golden/g13_retrieve_c2.json: seeds, shapes and the reference's outputs (ids, float64 fused scores as hex, method tags)."""
import asyncio
import importlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/advanced_rag"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def ref_module(name):
    if "advanced_rag" not in sys.modules or getattr(sys.modules["advanced_rag"], "__path__", None) != [REF]:
        pkg = types.ModuleType("advanced_rag")
        pkg.__path__ = [REF]
        sys.modules["advanced_rag"] = pkg
    return importlib.import_module(f"advanced_rag.{name}")


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference is not present: this generator runs in the build container only")
    R, C = ref_module("retrieval"), ref_module("constants")
    N, D, NQ = 100_000, 384, 8
    X = np.random.default_rng(2234).standard_normal((N, D)).astype(np.float32)
    srng = np.random.default_rng(6678)
    sp_idx = np.empty((N, 100), np.int32)
    sp_val = np.empty((N, 100), np.float32)
    for r in range(N):                     # the call order of tests/g5_data.inputs()
        sp_idx[r] = np.arange(100) * 100 + srng.integers(0, 100, size=100)
        sp_val[r] = np.abs(srng.standard_normal(100)).astype(np.float32)
    qrng = np.random.default_rng(5321)
    Q = qrng.standard_normal((NQ, D)).astype(np.float32)
    SQ = [((np.arange(100) * 100 + qrng.integers(0, 100, size=100)).astype(np.int32),
           np.abs(qrng.standard_normal(100)).astype(np.float32)) for _ in range(NQ)]

    class NumpyManager:
        def __init__(self, with_sparse):
            self.Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
            self.collections = {"semantic_index": 1, **({"sparse_index": 1} if with_sparse else {})}
            self.q = None

        async def _generate_semantic_embedding(self, text):
            return self.q[0]

        async def _generate_sparse_embedding(self, text):
            return {"indices": self.q[1][0].tolist(), "values": self.q[1][1].tolist()}

        async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
            if collection_name == "semantic_index":
                s = self.Xn @ (query_embedding / np.linalg.norm(query_embedding))
            else:
                idx = np.asarray(query_embedding["indices"])
                val = np.asarray(query_embedding["values"], dtype=np.float32)
                order = np.argsort(np.abs(val), kind="stable")
                keep = np.sort(order[int(np.floor(0.2 * len(val))):])
                qd = np.zeros(10000, dtype=np.float64)
                qd[idx[keep]] = val[keep]
                prod = qd[sp_idx] * sp_val.astype(np.float64)          # [N, 100] exact products
                acc = np.zeros(N, dtype=np.float64)
                for j in range(100):                                   # entry order, one row's sum per lane
                    acc += prod[:, j]
                s = acc.astype(np.float32)
            order = np.lexsort((np.arange(len(s)), -s))[:top_k]
            order = [int(i) for i in order if collection_name == "semantic_index" or s[i] > 0]
            return [{"id": f"doc{r // 10}::{r % 10}::{r:08x}", "content": f"row {r}", "score": float(s[r]),
                     "metadata": {"doc_id": f"doc{r // 10}", "chunk_index": r % 10}} for r in order]

    C.RetrievalConstants.TIMEOUT_SECONDS = 600.0
    g = {"N": N, "D": D, "corpus_seed": 2234, "sparse_seed": 6678, "query_seed": 5321, "n_queries": NQ, "runs": []}
    for with_sparse in (False, True):
        mgr = NumpyManager(with_sparse)
        hr = R.HybridRetriever(index_manager=mgr, config=R.RetrievalConfig(top_k=20))
        for qi in range(NQ):
            mgr.q = (Q[qi], SQ[qi])
            out = asyncio.run(hr.retrieve("plain statement", profile_hint="default"))
            g["runs"].append({"with_sparse": with_sparse, "query": qi, "ids": [o["id"] for o in out],
                              "scores": [float(o["score"]).hex() for o in out],
                              "methods": [sorted(o["retrieval_methods"]) for o in out]})
    with open(os.path.join(HERE, "g13_retrieve_c2.json"), "w") as f:
        json.dump(g, f, indent=0)
    print("wrote g13_retrieve_c2.json:", len(g["runs"]), "runs")


if __name__ == "__main__":
    main()
