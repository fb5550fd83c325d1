"""Generate golden fixture G6 from the REFERENCE's MilvusIndexManager.search (container-only).

Run here, never on the GPU box:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_g6.py

G6 pins what reference src/advanced_rag/indexing.py:439-551 does AROUND the Milvus RPC: which
arguments `Collection.search` receives (query data format, anns field, default search params per
collection, limit, expr, output fields) and how the returned hits are formatted into
{id := chunk_id, content, score, metadata{6 keys}} dicts, in the order Milvus returned them.

indexing.py needs `pymilvus` and `tenacity`, which are not installed here.  As SURVEY.md App. B.2
describes, the two are replaced by INERT stand-ins: module objects that only carry the imported
names (pymilvus: connections, Collection, CollectionSchema, FieldSchema, DataType, utility;
tenacity: retry as an identity decorator factory + three no-op helpers).  They implement nothing of
Milvus: the manager is built with connect=False and a recording fake object — exact numpy FLAT
search over a small payload table, returning hit objects shaped like pymilvus' (hit.score,
hit.entity.get(field)) — is placed in `manager.collections[name]`.  What the fixture pins is the
reference's own code between its `search()` signature and that call; the distance arithmetic is the
fake's (numpy fp32) and is compared only to 1e-4.

Output: tests/golden/g6_search_format.json (inputs + expected outputs only).
"""
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

SPARSE_DIM = 64
OUTPUT_FIELDS_SCALAR = ("chunk_id", "doc_id", "content", "chunk_index", "entropy", "redundancy", "domain_density",
                        "metadata_json", "timestamp")


def install_stand_ins():
    pm = types.ModuleType("pymilvus")
    for name in ("connections", "Collection", "CollectionSchema", "FieldSchema", "DataType", "utility"):
        setattr(pm, name, type(name, (), {}))
    sys.modules["pymilvus"] = pm
    tn = types.ModuleType("tenacity")
    tn.retry = lambda *a, **k: (lambda fn: fn)
    tn.stop_after_attempt = tn.wait_exponential = tn.retry_if_exception_type = lambda *a, **k: None
    sys.modules["tenacity"] = tn
    pkg = types.ModuleType("advanced_rag")
    pkg.__path__ = [REF]
    sys.modules["advanced_rag"] = pkg


def payload_table(n=24, dim=16, seed=77):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        rows.append({
            "chunk_id": f"doc{r // 4}::{r % 4}::{r:08x}", "doc_id": f"doc{r // 4}",
            "content": f"chunk {r} of doc{r // 4}: lorem ipsum {r * 7 % 11}", "chunk_index": r % 4, "token_count": 10 + r,
            "entropy": round(float(rng.random()), 3), "redundancy": round(float(rng.random()), 3),
            "domain_density": round(float(rng.random()), 3), "metadata_json": "{}",
            "timestamp": f"2024-0{1 + r % 9}-1{r % 9}T00:00:00",
        })
    dense = rng.standard_normal((n, dim)).astype(np.float32)
    sp_idx = [np.sort(rng.choice(SPARSE_DIM, 6, replace=False)).astype(np.int32) for _ in range(n)]
    sp_val = [np.abs(rng.standard_normal(6)).astype(np.float32) for _ in range(n)]
    return rows, dense, sp_idx, sp_val


class Entity:
    def __init__(self, row):
        self.row = row

    def get(self, field):
        v = self.row[field]
        # Milvus FLOAT fields are float32: pymilvus hands back the float32-rounded value as a Python float
        return float(np.float32(v)) if field in ("entropy", "redundancy", "domain_density") else v


class Hit:
    def __init__(self, row, score):
        self.entity, self.score = Entity(row), float(score)


class FakeCollection:
    """Exact FLAT stand-in for one Milvus collection; records how the reference called it."""

    def __init__(self, kind, rows, dense, sp_idx, sp_val):
        self.kind, self.rows, self.dense, self.sp_idx, self.sp_val = kind, rows, dense, sp_idx, sp_val
        self.calls = []

    def _passes(self, row, expr):
        if not expr:
            return True
        ns = {k: row[k] for k in row}
        return bool(eval(expr, {"__builtins__": {}}, ns))  # the generator's own literal exprs only ("and" grammar)

    def search(self, data, anns_field, param, limit, expr=None, output_fields=None):
        if self.kind == "dense":
            assert isinstance(data, list) and isinstance(data[0], list) and isinstance(data[0][0], float)
            shape = ["list", len(data), "list", len(data[0]), type(data[0][0]).__name__]
            q = np.asarray(data[0], dtype=np.float32)
            Xn = self.dense / np.linalg.norm(self.dense, axis=1, keepdims=True)
            s = Xn @ (q / np.linalg.norm(q))
        else:
            shape = [type(data).__name__, list(data.shape), str(data.dtype), data.indices.dtype.name]
            qd = np.asarray(data.todense(), dtype=np.float64).reshape(-1)
            s = np.array([float(np.sum(qd[i] * v.astype(np.float64))) for i, v in zip(self.sp_idx, self.sp_val)],
                         dtype=np.float32)
        self.calls.append({"data": shape, "anns_field": anns_field, "param": param, "limit": limit, "expr": expr,
                           "output_fields": list(output_fields)})
        order = [int(i) for i in np.lexsort((np.arange(len(s)), -s))
                 if self._passes(self.rows[int(i)], expr) and (self.kind == "dense" or s[int(i)] > 0)][:limit]
        return [[Hit(self.rows[i], s[i]) for i in order]]


def main():
    install_stand_ins()
    I = importlib.import_module("advanced_rag.indexing")
    rows, dense, sp_idx, sp_val = payload_table()
    mgr = I.MilvusIndexManager(semantic_dim=dense.shape[1], sparse_dim=SPARSE_DIM, domain_dim=dense.shape[1], connect=False)
    mgr.collections["semantic_index"] = FakeCollection("dense", rows, dense, sp_idx, sp_val)
    mgr.collections["sparse_index"] = FakeCollection("sparse", rows, dense, sp_idx, sp_val)
    rng = np.random.default_rng(78)
    cases = []

    def run(label, collection, query, top_k, filters=None, params="default"):
        coll = mgr.collections[collection]
        coll.calls.clear()
        kw = {} if params == "default" else {"search_params": params}
        q = np.asarray(query, np.float32) if collection != "sparse_index" else query
        out = asyncio.run(mgr.search(q, collection, top_k=top_k, filters=filters, **kw))
        cases.append({"label": label, "collection": collection,
                      "query": [float(x) for x in query] if collection != "sparse_index" else query,
                      "top_k": top_k, "filters": filters,
                      "search_params": None if params == "default" else params,
                      "collection_search_call": coll.calls[0], "results": out})

    q0 = rng.standard_normal(dense.shape[1]).astype(np.float32).tolist()
    q1 = rng.standard_normal(dense.shape[1]).astype(np.float32).tolist()
    run("dense-default-params", "semantic_index", q0, 5)
    run("dense-explicit-params", "semantic_index", q1, 8, None, {"metric_type": "COSINE", "params": {"ef": 64}})
    run("dense-filter", "semantic_index", q0, 6, 'doc_id == "doc2" and chunk_index >= 1')
    run("dense-more-than-rows", "semantic_index", q1, 40)
    sq = {"indices": [int(i) for i in sorted(rng.choice(SPARSE_DIM, 10, replace=False))],
          "values": [float(v) for v in np.abs(rng.standard_normal(10)).astype(np.float32)]}
    run("sparse-default-params", "sparse_index", sq, 5)
    run("sparse-explicit-params", "sparse_index", sq, 7, None, {"metric_type": "IP", "params": {"drop_ratio_search": 0.0}})
    run("sparse-filter", "sparse_index", sq, 24, "entropy >= 0.5")
    errors = []
    for label, fn in (("unknown-collection", lambda: mgr.search(np.zeros(4, np.float32), "nope")),
                      ("sparse-bad-payload", lambda: mgr.search(np.zeros(4, np.float32), "sparse_index"))):
        try:
            asyncio.run(fn())
            errors.append({"label": label, "error": None})
        except Exception as e:
            errors.append({"label": label, "error": type(e).__name__, "message": str(e)})
    g6 = {"sparse_dim": SPARSE_DIM, "dim": int(dense.shape[1]), "payload_seed": 77,
          "rows": rows, "dense": [[float(x) for x in r] for r in dense],
          "sparse": [{"indices": i.tolist(), "values": [float(x) for x in v]} for i, v in zip(sp_idx, sp_val)],
          "cases": cases, "errors": errors}
    with open(os.path.join(HERE, "g6_search_format.json"), "w") as f:
        json.dump(g6, f, indent=None)
    print("wrote g6_search_format.json:", len(cases), "cases")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    main()
