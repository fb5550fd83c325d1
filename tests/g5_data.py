"""Rebuilds the inputs of golden fixture g5 (tests/golden/g5_retrieve_c1.json) from its seeds.

g5 holds the outputs of the REFERENCE's HybridRetriever.retrieve (reference
src/advanced_rag/retrieval.py:215-491, imported by tests/golden/gen_golden.py:159-219) on
BASELINE config 1 (1k x 384 fp32) over an exact numpy FLAT collection: 8 queries, dense-only and
hybrid, top-20 ids + float64 fused scores + method tags.  The generator's corpus, sparse rows and
queries are restated here bit for bit (same numpy Generators, same call order), so the oracle and
the HIP path can be run on exactly the inputs the reference saw.
"""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SPARSE_DIM = 10000


def load():
    with open(os.path.join(GOLD, "g5_retrieve_c1.json")) as f:
        return json.load(f)


def row_id(r: int) -> str:
    return f"doc{r // 10}::{r % 10}::{r:08x}"


def inputs(g=None):
    """-> (g, X float32 [N,D], (indptr, idx, val) CSR of the sparse rows, Q float32 [8,D], SQ [(idx, val)] * 8)."""
    g = g or load()
    N, D = g["N"], g["D"]
    X = np.random.default_rng(g["corpus_seed"]).standard_normal((N, D)).astype(np.float32)
    srng = np.random.default_rng(g["sparse_seed"])
    idx_rows, val_rows = [], []
    for _ in range(N):
        idx_rows.append((np.arange(100) * 100 + srng.integers(0, 100, size=100)).astype(np.int32))
        val_rows.append(np.abs(srng.standard_normal(100)).astype(np.float32))
    qrng = np.random.default_rng(g["query_seed"])
    Q = qrng.standard_normal((g["n_queries"], D)).astype(np.float32)
    SQ = [((np.arange(100) * 100 + qrng.integers(0, 100, size=100)).astype(np.int32),
           np.abs(qrng.standard_normal(100)).astype(np.float32)) for _ in range(g["n_queries"])]
    indptr = np.arange(N + 1, dtype=np.int64) * 100
    return g, X, (indptr, np.concatenate(idx_rows), np.concatenate(val_rows)), Q, SQ
