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


def load_g13():
    """G13 = the same chain at BASELINE config 2's size (100 000 x 384 fp32; tests/golden/gen_golden_g13.py)."""
    with open(os.path.join(GOLD, "g13_retrieve_c2.json")) as f:
        return json.load(f)


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


def mmr_content(r: int) -> str:
    """Row contents for the MMR fixtures (G11 / G12): five tokens, two of them shared by every row and two by the rows
    of the same residue classes, so that token-Jaccard similarities take the values 2/8, 3/7 and 4/6."""
    return f"row {r} topic{r % 5} alpha{r % 3} common"


def payload_row(r: int) -> dict:
    """Scalar fields of corpus row r in the G12 fixture (every float exactly representable in float32)."""
    return {"chunk_id": row_id(r), "doc_id": f"doc{r // 10}", "content": mmr_content(r), "chunk_index": r % 10,
            "token_count": 5, "entropy": (r % 8) / 8.0, "redundancy": (r % 4) / 4.0, "domain_density": (r % 16) / 16.0,
            "metadata_json": "{}", "timestamp": f"2024-0{1 + r % 9}-1{r % 9}T00:00:00"}


class NumpyFlatManager:
    """Exact FLAT stand-in for the index manager over the g5 corpus (the stand-in of tests/golden/gen_golden.py, restated
    for the CPU tests): cosine over fp32-normalised rows, sparse IP in float64 after the 20 % drop, ties by lower row,
    sparse hits only with score > 0; hits carry the G11 / G12 payload (mmr_content, payload_row)."""

    def __init__(self, X, csr, Q, SQ, with_sparse=True, fixed_query=None):
        ptr, idx, val = csr
        self.Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        self.rows = [(idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]]) for r in range(X.shape[0])]
        self.Q, self.SQ, self.fixed = Q, SQ, fixed_query
        self.collections = {"semantic_index": 1, **({"sparse_index": 1} if with_sparse else {})}
        self.seen_top_k = []

    def _i(self, text):
        return self.fixed if self.fixed is not None else int(text.rsplit("q", 1)[1])

    async def _generate_semantic_embedding(self, text):
        return self.Q[self._i(text)]

    async def _generate_sparse_embedding(self, text):
        qi, qv = self.SQ[self._i(text)]
        return {"indices": qi.tolist(), "values": qv.astype(float).tolist()}

    async def search(self, query_embedding, collection_name, top_k=20, filters=None, search_params=None):
        self.seen_top_k.append(top_k)
        if collection_name == "semantic_index":
            s = self.Xn @ (query_embedding / np.linalg.norm(query_embedding))
        else:
            qi = np.asarray(query_embedding["indices"])
            qv = np.asarray(query_embedding["values"], dtype=np.float32)
            keep = np.sort(np.argsort(np.abs(qv), kind="stable")[int(np.floor(0.2 * len(qv))):])
            qd = np.zeros(SPARSE_DIM, dtype=np.float64)
            qd[qi[keep]] = qv[keep]
            s = np.array([float(np.sum(qd[ri] * rv.astype(np.float64))) for ri, rv in self.rows], dtype=np.float32)
        order = [int(i) for i in np.lexsort((np.arange(len(s)), -s))[:top_k] if collection_name == "semantic_index" or s[i] > 0]
        out = []
        for r in order:
            row = payload_row(r)
            out.append({"id": row["chunk_id"], "content": row["content"], "score": float(s[r]),
                        "metadata": {"doc_id": row["doc_id"], "chunk_index": row["chunk_index"], "entropy": row["entropy"],
                                     "redundancy": row["redundancy"], "domain_density": row["domain_density"],
                                     "timestamp": row["timestamp"]}})
        return out

    async def close(self):
        pass
