"""CPU oracle for the hybrid-search hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (advanced-rag-milvus_amd/) must never do so.

Two layers:
  * the C restatement (oracle.c, loaded with ctypes) — canonical scores and
    ranking, bit-exact with the HIP refine kernels;
  * numpy restatements — `dense_scores_np` (same canonical arithmetic, vectorised
    over rows, used to cross-check the C code) and the `cpu_*` functions that
    restate "the reference's Milvus path with Milvus replaced by exact numpy"
     for the CPU baseline: fp32 BLAS matmul + argpartition,
    scipy CSR·q, and the reference's RRF in Python.

Parity status: the reference's dense/sparse arithmetic runs inside the Milvus
server (pymilvus>=2.3.0 / milvusdb/milvus:v2.3.3), which is not in the tree and
which no reference test exercises -> "parity unpinned" for raw distances.  RRF
fusion, rerank, profiles and filter expressions ARE pinned: tests/golden/*.json
were generated from the imported reference by tests/golden/gen_golden.py.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import build as _build

F32, F16 = 0, 1
IP, COSINE = 0, 1

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = _build.build()
        L = ctypes.CDLL(path)
        c = ctypes
        L.oracle_dense_scores.argtypes = [c.c_void_p, c.c_int, c.c_int64, c.c_int, c.c_void_p, c.c_int, c.c_void_p]
        L.oracle_dense_scores.restype = None
        L.oracle_sparse_scores.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p,
                                           c.c_int, c.c_void_p]
        L.oracle_sparse_scores.restype = None
        L.oracle_topk.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_int, c.c_int, c.c_int64, c.c_void_p,
                                  c.c_void_p]
        L.oracle_topk.restype = c.c_int
        L.oracle_drop_query.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_double, c.c_void_p, c.c_void_p]
        L.oracle_drop_query.restype = c.c_int
        L.oracle_rrf.argtypes = [c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_void_p, c.c_int, c.c_double,
                                 c.c_double, c.c_double, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]
        L.oracle_rrf.restype = c.c_int
        L.oracle_float_to_half.argtypes = [c.c_float]
        L.oracle_float_to_half.restype = c.c_uint16
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dtype_code(X: np.ndarray) -> int:
    if X.dtype == np.float16:
        return F16
    if X.dtype == np.float32:
        return F32
    raise TypeError(f"dense shard must be float16 or float32, got {X.dtype}")


# --------------------------------------------------------------------------- dense
def dense_scores(X: np.ndarray, q: np.ndarray, metric: int) -> np.ndarray:
    """Canonical fp32 scores of every row of X (float16|float32 [n,d]) vs q (float32 [d])."""
    X = np.ascontiguousarray(X)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = X.shape
    out = np.empty(n, dtype=np.float32)
    lib().oracle_dense_scores(_ptr(X), _dtype_code(X), n, d, _ptr(q), metric, _ptr(out))
    return out


def dense_scores_np(X: np.ndarray, q: np.ndarray, metric: int) -> np.ndarray:
    """Same arithmetic in numpy: k-ordered fp64 accumulation, vectorised over rows."""
    n, d = X.shape
    q64 = q.astype(np.float64)
    s = np.zeros(n, dtype=np.float64)
    xn2 = np.zeros(n, dtype=np.float64)
    for k in range(d):
        col = X[:, k].astype(np.float64)
        s += col * q64[k]
        xn2 += col * col
    if metric == COSINE:
        qn2 = 0.0
        for k in range(d):
            qn2 += float(q64[k]) * float(q64[k])
        den = xn2 * qn2
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(den > 0.0, s / np.sqrt(den), 0.0)
    return s.astype(np.float32)


def topk(scores: np.ndarray, k: int, mask: Optional[np.ndarray] = None, only_positive: bool = False,
         row_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(ids int64[k], scores float32[k]) ranked by (score desc, row asc); -1/0 padded."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    ids = np.empty(k, dtype=np.int64)
    out = np.empty(k, dtype=np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    rc = lib().oracle_topk(_ptr(scores), scores.shape[0], _ptr(m), int(only_positive), k, row_offset, _ptr(ids),
                           _ptr(out))
    if rc < 0:
        raise MemoryError("oracle_topk")
    return ids, out


def dense_search(X: np.ndarray, Q: np.ndarray, k: int, metric: int, mask: Optional[np.ndarray] = None,
                 row_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Oracle for hr_search_dense: ids [B,k], scores [B,k]."""
    Q = np.atleast_2d(Q)
    ids = np.empty((Q.shape[0], k), dtype=np.int64)
    sc = np.empty((Q.shape[0], k), dtype=np.float32)
    for b in range(Q.shape[0]):
        ids[b], sc[b] = topk(dense_scores(X, Q[b], metric), k, mask, False, row_offset)
    return ids, sc


# --------------------------------------------------------------------------- sparse
def drop_query(idx: Sequence[int], val: Sequence[float], ratio: float) -> Tuple[np.ndarray, np.ndarray]:
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    oi = np.empty(max(len(idx), 1), dtype=np.int32)
    ov = np.empty(max(len(idx), 1), dtype=np.float32)
    kept = lib().oracle_drop_query(_ptr(idx), _ptr(val), len(idx), float(ratio), _ptr(oi), _ptr(ov))
    if kept < 0:
        raise MemoryError("oracle_drop_query")
    return oi[:kept].copy(), ov[:kept].copy()


def sparse_scores(indptr: np.ndarray, idx: np.ndarray, val: np.ndarray, q_idx: np.ndarray,
                  q_val: np.ndarray) -> np.ndarray:
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    q_idx = np.ascontiguousarray(q_idx, dtype=np.int32)
    q_val = np.ascontiguousarray(q_val, dtype=np.float32)
    n = indptr.shape[0] - 1
    out = np.empty(n, dtype=np.float32)
    lib().oracle_sparse_scores(_ptr(indptr), _ptr(idx), _ptr(val), n, _ptr(q_idx), _ptr(q_val), len(q_idx), _ptr(out))
    return out


def sparse_search(indptr, idx, val, queries, k: int, drop_ratio: float = 0.0, mask: Optional[np.ndarray] = None,
                  row_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Oracle for hr_search_sparse.  queries: list of (indices, values)."""
    ids = np.empty((len(queries), k), dtype=np.int64)
    sc = np.empty((len(queries), k), dtype=np.float32)
    for b, (qi, qv) in enumerate(queries):
        di, dv = drop_query(qi, qv, drop_ratio)
        ids[b], sc[b] = topk(sparse_scores(indptr, idx, val, di, dv), k, mask, True, row_offset)
    return ids, sc


# --------------------------------------------------------------------------- fusion
def rrf(ids_a, ids_b, ids_c=(), wa: float = 0.7, wb: float = 0.3, wc: float = 0.2, rrf_k: int = 60):
    """C restatement of reference retrieval.py:432-487.  Returns (ids, float64 scores, method masks)."""
    a = np.ascontiguousarray(ids_a, dtype=np.int64)
    b = np.ascontiguousarray(ids_b, dtype=np.int64)
    c = np.ascontiguousarray(ids_c, dtype=np.int64)
    cap = max(len(a) + len(b) + len(c), 1)
    oi = np.empty(cap, dtype=np.int64)
    os_ = np.empty(cap, dtype=np.float64)
    om = np.empty(cap, dtype=np.int32)
    n = lib().oracle_rrf(_ptr(a), len(a), _ptr(b), len(b), _ptr(c), len(c), wa, wb, wc, rrf_k, _ptr(oi), _ptr(os_),
                         _ptr(om))
    if n < 0:
        raise MemoryError("oracle_rrf")
    return oi[:n].copy(), os_[:n].copy(), om[:n].copy()


def rrf_py(ids_a, ids_b, ids_c=(), wa: float = 0.7, wb: float = 0.3, wc: float = 0.2, rrf_k: int = 60):
    """Python restatement of the same lines (dict insertion order + stable sort)."""
    fused = {}
    for bit, (lst, w) in enumerate(((ids_a, wa), (ids_b, wb), (ids_c, wc))):
        for rank, i in enumerate(lst, start=1):
            i = int(i)
            if i < 0:
                break
            ent = fused.setdefault(i, [0.0, 0])
            ent[0] += (1.0 / (rrf_k + rank)) * w
            ent[1] |= 1 << bit
    items = list(fused.items())
    items.sort(key=lambda kv: kv[1][0], reverse=True)
    return (np.array([k for k, _ in items], dtype=np.int64), np.array([v[0] for _, v in items], dtype=np.float64),
            np.array([v[1] for _, v in items], dtype=np.int32))


def mmr(ids, scores, contents, k: int, mmr_lambda: float):
    """Greedy MMR of reference retrieval.py:493-516 (`_mmr_diversify`) on a fused list given as parallel sequences:
    relevance = the fused score, similarity = token Jaccard of the lower-cased, whitespace-split contents (None = ""),
    value = lambda*rel - (1-lambda)*max similarity to the selected; the FIRST candidate (in fused order) with a strictly
    larger value wins; stops at k.  -> positions into the input, in selection order.  Pure Python: lists are <= 3k'."""
    toks = [set((c or "").lower().split()) for c in contents]
    cand = list(range(len(ids)))
    sel = []
    while cand and len(sel) < k:
        best, best_val = None, -1e9
        for i in cand:
            if not sel:
                val = scores[i]
            else:
                sim = max((len(toks[i] & toks[j]) / (len(toks[i] | toks[j]) or 1)) for j in sel)
                val = mmr_lambda * scores[i] - (1 - mmr_lambda) * sim
            if val > best_val:
                best, best_val = i, val
        sel.append(best)
        cand.remove(best)
    return sel


def learned_rank(scores, method_counts, k: int, base_weight: float = 1.0, method_bonus: float = 0.1):
    """The deterministic rerank branch of the reference (retrieval.py:544-563 with ranker.py:109-125, default
    LearnedRankerConfig, recency_weight = 0): new = base_weight*score + method_bonus*len(methods), float64, evaluated left
    to right as Python does; stable descending sort; cut to k.  -> (positions, new scores)."""
    new = [float(base_weight * float(s) + method_bonus * float(m) + 0.0 * 0.0) for s, m in zip(scores, method_counts)]
    order = sorted(range(len(new)), key=lambda i: new[i], reverse=True)[:k]
    return order, [new[i] for i in order]


def float_to_half_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> fp16 bit patterns via the C routine (cross-check of numpy's astype)."""
    flat = np.ascontiguousarray(x, dtype=np.float32).ravel()
    return np.array([lib().oracle_float_to_half(float(v)) for v in flat], dtype=np.uint16).reshape(np.shape(x))


# --------------------------------------------------------------------------- CPU baseline ("port")
def cpu_dense_topk(Xn32: np.ndarray, Q: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """The reference path with Milvus replaced by exact numpy:
    fp32 BLAS matmul on L2-normalised rows + argpartition + sort.  Xn32 must be
    pre-normalised float32 (done once at "ingest", outside the timed region)."""
    Qn = Q / np.maximum(np.linalg.norm(Q, axis=1, keepdims=True), 1e-30)
    S = Qn.astype(np.float32) @ Xn32.T  # [B, N]
    kk = min(k, S.shape[1])
    part = np.argpartition(-S, kk - 1, axis=1)[:, :kk]
    ps = np.take_along_axis(S, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    return np.take_along_axis(part, order, axis=1).astype(np.int64), np.take_along_axis(ps, order, axis=1)


def cpu_sparse_topk(csr, q_csr, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """scipy CSR·q^T for a batch of sparse queries, then top-k of the positive scores."""
    S = (csr @ q_csr.T).toarray().T  # [B, N] float32
    kk = min(k, S.shape[1])
    part = np.argpartition(-S, kk - 1, axis=1)[:, :kk]
    ps = np.take_along_axis(S, part, axis=1)
    order = np.lexsort((part, -ps), axis=1)
    ids = np.take_along_axis(part, order, axis=1).astype(np.int64)
    sc = np.take_along_axis(ps, order, axis=1)
    ids[sc <= 0] = -1
    return ids, sc


def cpu_hybrid(Xn32, csr, Q, q_csr, top_k: int = 20, wa: float = 0.7, wb: float = 0.3):
    """dense top-2k + sparse top-2k + RRF -> fused top_k, as HybridRetriever.retrieve does."""
    di, _ = cpu_dense_topk(Xn32, Q, 2 * top_k)
    si, _ = cpu_sparse_topk(csr, q_csr, 2 * top_k)
    out = []
    for b in range(Q.shape[0]):
        ids, sc, _ = rrf_py(di[b], [i for i in si[b] if i >= 0], (), wa, wb)
        out.append((ids[:top_k], sc[:top_k]))
    return out


# ---- filter expressions (test infrastructure, like everything in this package) ----------------------------------------
# What HybridRetriever._build_filter_expression emits (reference retrieval.py:565-632): `field OP literal` terms joined by
# " and ", OP in {>=, <=, >, <, ==, !=}, literals = integers, floats, True / False, or "double-quoted strings" with \\ and
# \" escapes.  Milvus' evaluation of them is third-party and absent (unpinned); the semantics below are
# the build's, written out explicitly so that the device kernel and the product's host evaluator are checked against a
# statement that does not lean on numpy's type promotion:
#   * FLOAT fields (float32 columns) compare against the literal ROUNDED TO float32;
#   * INT64 fields compare against an integer literal as integers (a bool counts as 0 / 1), against a float literal as float64;
#   * VARCHAR fields compare by the UTF-8 bytes of the strings (= by code point); a string literal on a numeric field or a
#     number on a string field is an error.
def _filter_terms(expr: str):
    """-> [(field, op, literal)] by a character scanner (no regular expressions): literal is str | bool | int | float."""
    i, n, out = 0, len(expr), []
    ops = (">=", "<=", "==", "!=", ">", "<")
    while True:
        while i < n and expr[i].isspace():
            i += 1
        j = i
        while j < n and (expr[j].isalnum() or expr[j] == "_"):
            j += 1
        if j == i:
            raise ValueError(f"field name expected at {i} in {expr!r}")
        field = expr[i:j]
        i = j
        while i < n and expr[i].isspace():
            i += 1
        op = next((o for o in ops if expr.startswith(o, i)), None)
        if op is None:
            raise ValueError(f"operator expected at {i} in {expr!r}")
        i += len(op)
        while i < n and expr[i].isspace():
            i += 1
        if i < n and expr[i] == '"':
            i += 1
            buf = []
            while True:
                if i >= n:
                    raise ValueError(f"unterminated string in {expr!r}")
                ch = expr[i]
                if ch == "\\" and i + 1 < n:
                    buf.append(expr[i + 1])
                    i += 2
                elif ch == '"':
                    i += 1
                    break
                else:
                    buf.append(ch)
                    i += 1
            lit = "".join(buf)
        else:
            j = expr.find(" and ", i)
            raw = (expr[i:] if j < 0 else expr[i:j]).strip()
            i = n if j < 0 else j
            if raw in ("True", "true"):
                lit = True
            elif raw in ("False", "false"):
                lit = False
            else:
                try:
                    lit = int(raw)
                except ValueError:
                    lit = float(raw)       # ValueError for anything else
        out.append((field, op, lit))
        while i < n and expr[i].isspace():
            i += 1
        if i >= n:
            return out
        if not expr.startswith("and", i):
            raise ValueError(f"'and' expected at {i} in {expr!r}")
        i += 3


def filter_mask(expr: str, columns, n_rows: int) -> np.ndarray:
    """Boolean row predicate of a conjunctive filter expression over `columns` (name -> numpy array of n_rows)."""
    import operator
    cmp = {">=": operator.ge, "<=": operator.le, ">": operator.gt, "<": operator.lt, "==": operator.eq, "!=": operator.ne}
    keep = np.ones(n_rows, dtype=bool)
    for field, op, lit in _filter_terms(expr):
        if field not in columns:
            raise ValueError(f"unknown filter field: {field}")
        col = np.asarray(columns[field])
        if col.dtype.kind in "US":
            if not isinstance(lit, str):
                raise ValueError(f"field {field} holds strings; got {lit!r}")
            a = np.char.encode(col, "utf-8") if col.dtype.kind == "U" else col
            keep &= cmp[op](a, np.bytes_(lit.encode("utf-8")))
        else:
            if isinstance(lit, str):
                raise ValueError(f"field {field} is numeric; got string {lit!r}")
            if col.dtype.kind == "f":
                keep &= cmp[op](col.astype(np.float32), np.float32(lit))
            elif isinstance(lit, float):
                keep &= cmp[op](col.astype(np.float64), np.float64(lit))
            else:
                keep &= cmp[op](col.astype(np.int64), np.int64(int(lit)))
    return keep
