"""ctypes binding of libhbmrag.so (C ABI: include/hbmrag.h).

The library is the product's compute path; there is no Python or CPU fallback.
If the shared object is missing or no HIP device is present, loading or
`ShardHandle(...)` raises — callers fail loudly instead of degrading.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence, Tuple

import numpy as np

HR_F32, HR_F16 = 0, 1
HR_METRIC_IP, HR_METRIC_COSINE = 0, 1
HR_METHOD_SEMANTIC, HR_METHOD_SPARSE, HR_METHOD_DOMAIN = 1, 2, 4
HR_MAX_TOPK = 256
HR_N_PHASES = 10
PHASE_NAMES = ("prep", "dense_scan", "group_select", "refine", "topk",
               "sparse_scan", "sparse_select", "sparse_refine", "sparse_topk", "finish_fused")
HR_DEBUG_FINISH_MODE, HR_DEBUG_FAIL_NEXT_BUILD, HR_DEBUG_DENSE_KERNELS, HR_DEBUG_SPARSE_RPB, HR_DEBUG_GROUP_ROWS = 1, 2, 3, 4, 5
HR_DEBUG_NO_TRIM = 6

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libhbmrag.so")

_lib = None
_lib_lock = threading.Lock()


class HbmRagError(RuntimeError):
    """HIP/runtime failure reported by libhbmrag (status 2..5)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libhbmrag status {status}: {message}")
        self.status = status


_c = ctypes
_SIGNATURES = {
    "hr_version": (_c.c_int, []),
    "hr_create": (_c.c_int, [_c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int64, _c.POINTER(_c.c_void_p)]),
    "hr_destroy": (None, [_c.c_void_p]),
    "hr_last_error": (_c.c_char_p, [_c.c_void_p]),
    "hr_set_row_offset": (_c.c_int, [_c.c_void_p, _c.c_int64]),
    "hr_reserve": (_c.c_int, [_c.c_void_p, _c.c_int64]),
    "hr_add_dense": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64]),
    "hr_add_dense_raw": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64]),
    "hr_add_dense_raw_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p]),
    "hr_add_sparse": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64]),
    "hr_finalize": (_c.c_int, [_c.c_void_p]),
    "hr_save": (_c.c_int, [_c.c_void_p, _c.c_char_p]),
    "hr_load": (_c.c_int, [_c.c_char_p, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "hr_get_info": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32),
                               _c.POINTER(_c.c_int64)]),
    "hr_num_rows": (_c.c_int64, [_c.c_void_p]),
    "hr_num_sparse_rows": (_c.c_int64, [_c.c_void_p]),
    "hr_device_bytes": (_c.c_int64, [_c.c_void_p]),
    "hr_dense_scan_bytes": (_c.c_int64, [_c.c_void_p]),
    "hr_search_dense": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                   _c.c_void_p]),
    "hr_search_sparse": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                    _c.c_float, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_search_dense_dmask": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                         _c.c_void_p]),
    "hr_search_sparse_dmask": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                          _c.c_float, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_filter_eval_dev": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                      _c.c_void_p, _c.c_void_p]),
    "hr_bm25_encode_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                      _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_hash_tokenize_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p]),
    "hr_fuse_rrf": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int,
                               _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_void_p, _c.c_void_p,
                               _c.c_void_p, _c.c_void_p]),
    "hr_search_dense_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                       _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_search_sparse_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p]),
    "hr_search_hybrid_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                        _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_void_p]),
    "hr_hybrid_prep_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                      _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "hr_hybrid_scan_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                      _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "hr_hybrid_finish_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_void_p]),
    "hr_fuse_rrf_dev": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int,
                                   _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_int, _c.c_void_p,
                                   _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_merge_topk_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                     _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_post_lists_dev": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p]),
    "hr_stream_create": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p, _c.c_int, _c.POINTER(_c.c_void_p)]),
    "hr_stream_destroy": (_c.c_int, [_c.c_int, _c.c_void_p]),
    "hr_set_scan_cus": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "hr_debug_option": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int]),
    "hr_rerank_linear_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                        _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_void_p,
                                        _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "hr_add_layernorm_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                            _c.c_int, _c.c_float, _c.c_void_p]),
    "hr_embed_layernorm_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_float, _c.c_int64,
                                              _c.c_int64, _c.c_void_p]),
    "hr_attention_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_float, _c.c_void_p]),
    "hr_attention_rows_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                             _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float,
                                             _c.c_void_p]),
    "hr_linear_rows_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                          _c.c_int64, _c.c_void_p]),
    "hr_attention_fr_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_float, _c.c_void_p]),
    "hr_encoder_tail_f16_dev": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                           _c.c_int64, _c.c_int, _c.c_int, _c.c_float, _c.c_int, _c.c_void_p]),
    "hr_set_profiling": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "hr_last_kernel_ms": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libhbmrag.so (building nothing: run __graft_entry__.build() or `make` first)."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("HBMRAG_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise FileNotFoundError(
                f"{p} not found: build it with `make -C advanced-rag-milvus_amd` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            # torch ships its own libamdhip64 with the same soname; import it first
            # so both sides share ONE HIP runtime (device pointers interoperate).
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the library itself
            pass
        L = ctypes.CDLL(p)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if path is None:
            _lib = L
        return L


def _vp(a) -> Optional[ctypes.c_void_p]:
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(int(a))  # raw device pointer (e.g. tensor.data_ptr())


class ShardHandle:
    """One GPU's shard: dense tiles + sparse postings, owned by libhbmrag."""

    def __init__(self, dim: int, dtype: int = HR_F16, metric: int = HR_METRIC_COSINE, sparse_dim: int = 0,
                 device: int = 0, _adopt: Optional[int] = None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if _adopt is not None:  # handle created by hr_load
            self._h = ctypes.c_void_p(_adopt)
            self.dim, self.dtype, self.metric, self.sparse_dim, self.device = dim, dtype, metric, sparse_dim, device
            return
        self.dim, self.dtype, self.metric, self.sparse_dim, self.device = dim, dtype, metric, sparse_dim, device
        rc = self._lib.hr_create(device, dim, dtype, metric, sparse_dim, ctypes.byref(self._h))
        if rc != 0:
            msg = (self._lib.hr_last_error(None) or b"").decode()
            self._h = ctypes.c_void_p()
            self._raise(rc, msg)

    # -- error mapping: HR_EINVAL -> ValueError (as the reference raises for bad
    #    collections/shapes, indexing.py:466-467); everything else -> HbmRagError
    def _raise(self, rc: int, msg: Optional[str] = None):
        if msg is None:
            msg = (self._lib.hr_last_error(self._h) or b"").decode()
        if rc == 1:
            raise ValueError(msg)
        raise HbmRagError(rc, msg)

    def _check(self, rc: int):
        if rc != 0:
            self._raise(rc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.hr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- snapshot
    def save(self, path: str):
        self._check(self._lib.hr_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path: str, dim: int, dtype: int = HR_F16, metric: int = HR_METRIC_COSINE, sparse_dim: int = 0,
             device: int = 0) -> "ShardHandle":
        """Load a snapshot; dim/dtype/metric/sparse_dim describe what the caller expects and are checked against what
        the file holds (a mismatch would make later calls read query or row buffers of the wrong size)."""
        lib = load_library()
        h = ctypes.c_void_p()
        rc = lib.hr_load(os.fsencode(path), device, ctypes.byref(h))
        if rc != 0:
            msg = (lib.hr_last_error(None) or b"").decode()
            if rc == 1:
                raise ValueError(msg)
            raise HbmRagError(rc, msg)
        f_dim, f_sdim = ctypes.c_int64(), ctypes.c_int64()
        f_dtype, f_metric = ctypes.c_int32(), ctypes.c_int32()
        lib.hr_get_info(h, ctypes.byref(f_dim), ctypes.byref(f_dtype), ctypes.byref(f_metric), ctypes.byref(f_sdim))
        got = (f_dim.value, f_dtype.value, f_metric.value, f_sdim.value)
        if got != (dim, dtype, metric, sparse_dim):
            lib.hr_destroy(h)
            raise ValueError(f"snapshot {path} holds (dim, dtype, metric, sparse_dim) = {got}, expected "
                             f"{(dim, dtype, metric, sparse_dim)}")
        return cls(dim, dtype, metric, sparse_dim, device, _adopt=h.value)

    # -- ingest
    def set_row_offset(self, first_row: int):
        self._check(self._lib.hr_set_row_offset(self._h, first_row))

    def reserve(self, n_rows: int):
        self._check(self._lib.hr_reserve(self._h, n_rows))

    def add_dense(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n,{self.dim}], got {rows.shape}")
        if rows.dtype == np.float32 and self.dtype == HR_F16:
            self._check(self._lib.hr_add_dense(self._h, _vp(rows), rows.shape[0]))
        elif (rows.dtype == np.float16 and self.dtype == HR_F16) or (rows.dtype == np.float32 and self.dtype == HR_F32):
            self._check(self._lib.hr_add_dense_raw(self._h, _vp(rows), rows.shape[0]))
        else:
            raise ValueError(f"cannot ingest {rows.dtype} rows into this shard")

    def add_dense_dev(self, d_ptr: int, n: int, stream: int = 0):
        self._check(self._lib.hr_add_dense_raw_dev(self._h, _vp(d_ptr), n, _vp(stream) if stream else None))

    def add_sparse(self, indptr: np.ndarray, indices: np.ndarray, values: np.ndarray):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        self._check(self._lib.hr_add_sparse(self._h, _vp(indptr), _vp(indices), _vp(values), indptr.shape[0] - 1))

    def finalize(self):
        self._check(self._lib.hr_finalize(self._h))

    @property
    def num_rows(self) -> int:
        return int(self._lib.hr_num_rows(self._h))

    @property
    def num_sparse_rows(self) -> int:
        return int(self._lib.hr_num_sparse_rows(self._h))

    @property
    def device_bytes(self) -> int:
        return int(self._lib.hr_device_bytes(self._h))

    @property
    def dense_scan_bytes(self) -> int:
        return int(self._lib.hr_dense_scan_bytes(self._h))

    # -- search, host buffers
    @staticmethod
    def _mask(rowmask, n_rows: int):
        """The library copies (n_rows + 7) // 8 bytes of a row mask: a shorter buffer would be over-read."""
        if rowmask is None:
            return None
        m = np.ascontiguousarray(rowmask, dtype=np.uint8)
        if m.size < (n_rows + 7) // 8:
            raise ValueError(f"row mask has {m.size} bytes, the collection's {n_rows} rows need {(n_rows + 7) // 8}")
        return m

    def search_dense(self, q: np.ndarray, k: int, rowmask: Optional[np.ndarray] = None,
                     d_rowmask: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """rowmask = packed host mask (uploaded), or d_rowmask = device pointer of a mask already in HBM."""
        q = np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)
        if q.shape[1] != self.dim:
            raise ValueError(f"query dim {q.shape[1]} != shard dim {self.dim}")
        B = q.shape[0]
        ids = np.empty((B, k), dtype=np.int64)
        sc = np.empty((B, k), dtype=np.float32)
        if d_rowmask:
            self._check(self._lib.hr_search_dense_dmask(self._h, _vp(q), B, k, _vp(d_rowmask), _vp(ids), _vp(sc)))
            return ids, sc
        m = self._mask(rowmask, self.num_rows)
        self._check(self._lib.hr_search_dense(self._h, _vp(q), B, k, _vp(m), _vp(ids), _vp(sc)))
        return ids, sc

    def search_sparse(self, queries: Sequence[Tuple[Sequence[int], Sequence[float]]], k: int,
                      drop_ratio: float = 0.0, rowmask: Optional[np.ndarray] = None,
                      d_rowmask: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        B = len(queries)
        indptr = np.zeros(B + 1, dtype=np.int64)
        for b, (qi, _) in enumerate(queries):
            indptr[b + 1] = indptr[b] + len(qi)
        idx = np.concatenate([np.asarray(qi, dtype=np.int32) for qi, _ in queries]) if indptr[-1] else np.zeros(0, np.int32)
        val = np.concatenate([np.asarray(qv, dtype=np.float32) for _, qv in queries]) if indptr[-1] else np.zeros(0, np.float32)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        ids = np.empty((B, k), dtype=np.int64)
        sc = np.empty((B, k), dtype=np.float32)
        if d_rowmask:
            self._check(self._lib.hr_search_sparse_dmask(self._h, _vp(indptr), _vp(idx), _vp(val), B, k, float(drop_ratio),
                                                         _vp(d_rowmask), _vp(ids), _vp(sc)))
            return ids, sc
        m = self._mask(rowmask, self.num_sparse_rows)
        self._check(self._lib.hr_search_sparse(self._h, _vp(indptr), _vp(idx), _vp(val), B, k, float(drop_ratio),
                                               _vp(m), _vp(ids), _vp(sc)))
        return ids, sc

    def fuse_rrf(self, ids_a, ids_b, ids_c=(), wa: float = 0.7, wb: float = 0.3, wc: float = 0.2, rrf_k: int = 60):
        a = np.ascontiguousarray(ids_a, dtype=np.int64)
        b = np.ascontiguousarray(ids_b, dtype=np.int64)
        c = np.ascontiguousarray(ids_c, dtype=np.int64)
        cap = max(len(a) + len(b) + len(c), 1)
        oi = np.empty(cap, dtype=np.int64)
        os_ = np.empty(cap, dtype=np.float64)
        om = np.empty(cap, dtype=np.int32)
        n = ctypes.c_int32(0)
        self._check(self._lib.hr_fuse_rrf(self._h, _vp(a) if len(a) else None, len(a), _vp(b) if len(b) else None,
                                          len(b), _vp(c) if len(c) else None, len(c), wa, wb, wc, rrf_k, _vp(oi),
                                          _vp(os_), _vp(om), ctypes.byref(n)))
        return oi[:n.value].copy(), os_[:n.value].copy(), om[:n.value].copy()

    # -- search, device buffers (raw pointers; asynchronous on `stream`)
    def search_dense_dev(self, d_q: int, B: int, k: int, d_ids: int, d_scores: int, d_flags: int = 0,
                         d_rowmask: int = 0, stream: int = 0):
        self._check(self._lib.hr_search_dense_dev(self._h, _vp(d_q), B, k, _vp(d_rowmask) if d_rowmask else None,
                                                  _vp(d_ids), _vp(d_scores), _vp(d_flags) if d_flags else None,
                                                  _vp(stream) if stream else None))

    def search_sparse_dev(self, d_indptr: int, d_idx: int, d_val: int, B: int, nnz_total: int, max_q_nnz: int, k: int,
                          d_ids: int, d_scores: int, d_flags: int = 0, d_rowmask: int = 0, stream: int = 0):
        self._check(self._lib.hr_search_sparse_dev(self._h, _vp(d_indptr), _vp(d_idx), _vp(d_val), B, nnz_total,
                                                   max_q_nnz, k, _vp(d_rowmask) if d_rowmask else None, _vp(d_ids),
                                                   _vp(d_scores), _vp(d_flags) if d_flags else None,
                                                   _vp(stream) if stream else None))

    def search_hybrid_dev(self, d_q: int, d_indptr: int, d_idx: int, d_val: int, B: int, nnz_total: int,
                          max_q_nnz: int, k: int, d_ids: int, d_scores: int, d_flags: int = 0, d_rowmask: int = 0,
                          stream: int = 0):
        """Dense + sparse lists of one batch in one call ([2][B][k] outputs; dense first)."""
        self._check(self._lib.hr_search_hybrid_dev(self._h, _vp(d_q), _vp(d_indptr), _vp(d_idx) if d_idx else None,
                                                   _vp(d_val) if d_val else None, B, nnz_total, max_q_nnz, k,
                                                   _vp(d_rowmask) if d_rowmask else None, _vp(d_ids), _vp(d_scores),
                                                   _vp(d_flags) if d_flags else None, _vp(stream) if stream else None))

    def hybrid_prep_dev(self, d_q: int, d_indptr: int, d_idx: int, d_val: int, B: int, nnz_total: int, max_q_nnz: int,
                        k: int, slot: int, stream: int = 0):
        """Query preparation of `slot` alone, on `stream`; the next hybrid_scan_dev on the slot enqueues the scans only."""
        self._check(self._lib.hr_hybrid_prep_dev(self._h, _vp(d_q), _vp(d_indptr), _vp(d_idx) if d_idx else None,
                                                 _vp(d_val) if d_val else None, B, nnz_total, max_q_nnz, k, slot,
                                                 _vp(stream) if stream else None))

    def hybrid_scan_dev(self, d_q: int, d_indptr: int, d_idx: int, d_val: int, B: int, nnz_total: int, max_q_nnz: int,
                        k: int, slot: int, stream: int = 0, d_rowmask: int = 0):
        self._check(self._lib.hr_hybrid_scan_dev(self._h, _vp(d_q), _vp(d_indptr), _vp(d_idx) if d_idx else None,
                                                 _vp(d_val) if d_val else None, B, nnz_total, max_q_nnz, k,
                                                 _vp(d_rowmask) if d_rowmask else None, slot,
                                                 _vp(stream) if stream else None))

    def hybrid_finish_dev(self, d_q: int, d_indptr: int, d_idx: int, d_val: int, B: int, max_q_nnz: int, k: int,
                          slot: int, d_ids: int, d_scores: int, d_flags: int = 0, stream: int = 0, d_rowmask: int = 0):
        self._check(self._lib.hr_hybrid_finish_dev(self._h, _vp(d_q), _vp(d_indptr), _vp(d_idx) if d_idx else None,
                                                   _vp(d_val) if d_val else None, B, max_q_nnz, k,
                                                   _vp(d_rowmask) if d_rowmask else None, slot, _vp(d_ids),
                                                   _vp(d_scores), _vp(d_flags) if d_flags else None,
                                                   _vp(stream) if stream else None))

    def set_scan_cus(self, n_cus: int):
        """Compute units the scans' stream may occupy (0 = all): sizes their persistent grids (hr_stream_create masks)."""
        self._check(self._lib.hr_set_scan_cus(self._h, int(n_cus)))

    def debug_option(self, key: int, value: int):
        self._check(self._lib.hr_debug_option(self._h, key, value))

    # -- measurement
    def set_profiling(self, level: int):
        self._check(self._lib.hr_set_profiling(self._h, level))

    def kernel_ms(self):
        """{phase: (mean ms per launch, launches)} since the previous call."""
        buf = np.zeros(2 * HR_N_PHASES, dtype=np.float32)
        self._check(self._lib.hr_last_kernel_ms(self._h, _vp(buf), buf.shape[0]))
        return {PHASE_NAMES[p]: (float(buf[p]), int(buf[HR_N_PHASES + p])) for p in range(HR_N_PHASES)}


def fuse_rrf_dev(d_a: int, ka: int, d_b: int, kb: int, d_c: int, kc: int, B: int, wa: float, wb: float, wc: float,
                 rrf_k: int, top_k: int, d_out_ids: int, d_out_scores: int, d_out_methods: int, d_n_out: int,
                 stream: int = 0):
    L = load_library()
    rc = L.hr_fuse_rrf_dev(_vp(d_a) if ka else None, ka, _vp(d_b) if kb else None, kb, _vp(d_c) if kc else None, kc,
                           B, wa, wb, wc, rrf_k, top_k, _vp(d_out_ids), _vp(d_out_scores), _vp(d_out_methods),
                           _vp(d_n_out), _vp(stream) if stream else None)
    if rc != 0:
        msg = (L.hr_last_error(None) or b"").decode()
        if rc == 1:
            raise ValueError(msg)
        raise HbmRagError(rc, msg)


def _raise_global(L, rc: int):
    msg = (L.hr_last_error(None) or b"").decode()
    if rc == 1:
        raise ValueError(msg)
    raise HbmRagError(rc, msg)


def merge_topk_dev(d_scores: int, d_ids: int, n_lists: int, B: int, k_in: int, k_out: int, d_out_ids: int,
                   d_out_scores: int, stream: int = 0, score_stride: int = 0, id_stride: int = 0):
    L = load_library()
    rc = L.hr_merge_topk_dev(_vp(d_scores), _vp(d_ids), n_lists, score_stride or B * k_in, id_stride or B * k_in, B,
                             k_in, k_out, _vp(d_out_ids), _vp(d_out_scores), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


class FilterTerm(ctypes.Structure):
    """hr_filter_term of include/hbmrag.h."""
    _fields_ = [("kind", _c.c_int32), ("op", _c.c_int32), ("col", _c.c_void_p), ("ival", _c.c_int64), ("dval", _c.c_double),
                ("fval", _c.c_float), ("reserved", _c.c_uint32), ("key", _c.c_uint64 * 2)]


HR_COL_I64, HR_COL_I64_VS_F64, HR_COL_F32, HR_COL_STR16 = 0, 1, 2, 3
FILTER_OPS = {"==": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5}


def filter_eval_dev(terms: Sequence["FilterTerm"], n_rows: int, d_deleted: int, d_mask: int, d_undecided: int,
                    d_counts: int, stream: int = 0):
    L = load_library()
    arr = (FilterTerm * max(len(terms), 1))(*terms)
    rc = L.hr_filter_eval_dev(ctypes.byref(arr), len(terms), n_rows, _vp(d_deleted) if d_deleted else None, _vp(d_mask),
                              _vp(d_undecided), _vp(d_counts), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def bm25_encode_dev(d_text: int, d_off: int, n_docs: int, sparse_dim: int, k1: float, b: float, avgdl: float, cap: int,
                    d_idx: int, d_val: int, d_nnz: int, d_flags: int, stream: int = 0):
    """hr_bm25_encode_dev: BM25 document payloads of a batch on the device (include/hbmrag.h)."""
    L = load_library()
    rc = L.hr_bm25_encode_dev(_vp(d_text) if d_text else None, _vp(d_off), n_docs, sparse_dim, float(k1), float(b), float(avgdl), cap,
                              _vp(d_idx) if d_idx else None, _vp(d_val) if d_val else None, _vp(d_nnz), _vp(d_flags),
                              _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def hash_tokenize_dev(d_text: int, d_off: int, n: int, max_len: int, vocab: int, d_ids: int, d_lens: int, d_flags: int,
                      stream: int = 0):
    """hr_hash_tokenize_dev: the hash tokenizer of the encoder hooks for a batch of texts (include/hbmrag.h)."""
    L = load_library()
    rc = L.hr_hash_tokenize_dev(_vp(d_text) if d_text else None, _vp(d_off), n, max_len, vocab, _vp(d_ids) if d_ids else None,
                                _vp(d_lens), _vp(d_flags), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


class PostArgs(ctypes.Structure):
    """hr_post_args of include/hbmrag.h, field for field."""
    _fields_ = [
        ("ids", _c.c_void_p * 3), ("scores", _c.c_void_p * 3), ("k_in", _c.c_int32 * 3), ("k_fuse", _c.c_int32 * 3),
        ("n_lists", _c.c_int32), ("rrf_k", _c.c_int32), ("id_stride", _c.c_int64), ("score_stride", _c.c_int64),
        ("merged_ids", _c.c_void_p * 3), ("merged_scores", _c.c_void_p * 3), ("w", _c.c_double * 3),
        ("fused_ids", _c.c_void_p), ("fused_scores", _c.c_void_p), ("fused_methods", _c.c_void_p),
        ("fused_n", _c.c_void_p), ("top_k", _c.c_int32), ("rerank", _c.c_int32), ("base_w", _c.c_double),
        ("method_bonus", _c.c_double), ("recency_w", _c.c_double), ("recency", _c.c_void_p), ("k_out", _c.c_int32),
        ("reserved", _c.c_int32), ("rr_ids", _c.c_void_p), ("rr_scores", _c.c_void_p), ("rr_orig", _c.c_void_p),
        ("flags", _c.c_void_p), ("flag_stride", _c.c_int64), ("n_flag_rows", _c.c_int32), ("reserved2", _c.c_int32),
        ("agg_flags", _c.c_void_p), ("w_query", _c.c_void_p),
    ]


def post_lists_dev(args: PostArgs, B: int, stream: int = 0):
    """[merge of every modality's exchanged lists] -> RRF -> [learned-ranker rerank] in one launch."""
    L = load_library()
    rc = L.hr_post_lists_dev(ctypes.byref(args), B, _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def debug_option(key: int, value: int):
    L = load_library()
    rc = L.hr_debug_option(None, key, value)
    if rc != 0:
        _raise_global(L, rc)


def stream_create(device: int = 0, priority: int = 0, cu_mask: Optional[Sequence[int]] = None) -> int:
    """A HIP stream (raw handle) with a priority or, when `cu_mask` (32-bit words, bit i = CU i) is given, confined to
    those compute units.  Wrap it with torch.cuda.ExternalStream; release it with stream_destroy."""
    L = load_library()
    out = ctypes.c_void_p()
    words = np.ascontiguousarray(cu_mask, dtype=np.uint32) if cu_mask is not None else None
    rc = L.hr_stream_create(device, priority, _vp(words) if words is not None else None,
                            0 if words is None else int(words.shape[0]), ctypes.byref(out))
    if rc != 0:
        _raise_global(L, rc)
    return int(out.value)


def stream_destroy(device: int, stream: int):
    L = load_library()
    rc = L.hr_stream_destroy(device, _vp(stream))
    if rc != 0:
        _raise_global(L, rc)


def cu_mask_words(n_cus_total: int, lo: int, hi: int):
    """32-bit mask words with bits [lo, hi) set out of n_cus_total."""
    words = np.zeros((n_cus_total + 31) // 32, dtype=np.uint32)
    for i in range(lo, hi):
        words[i >> 5] |= np.uint32(1 << (i & 31))
    return words


def add_layernorm_f16_dev(d_x: int, d_residual: int, d_gamma: int, d_beta: int, d_out: int, rows: int, hidden: int,
                          eps: float, stream: int = 0):
    """out = LayerNorm(x (+ residual)) * gamma + beta over fp16 rows (device pointers; d_residual may be 0)."""
    L = load_library()
    rc = L.hr_add_layernorm_f16_dev(_vp(d_x), _vp(d_residual) if d_residual else None, _vp(d_gamma), _vp(d_beta),
                                    _vp(d_out), rows, hidden, eps, _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def embed_layernorm_f16_dev(d_ids: int, d_types: int, d_word: int, d_pos: int, d_seg: int, d_gamma: int, d_beta: int, d_out: int,
                            n_seq: int, T: int, hidden: int, eps: float, n_word: int, n_seg: int, stream: int = 0):
    """out[s, t] = LayerNorm(word[ids[s, t]] + pos[t] + seg[types[s, t]]): int64 ids / types (clamped into the tables),
    fp16 tables and output."""
    L = load_library()
    rc = L.hr_embed_layernorm_f16_dev(_vp(d_ids), _vp(d_types), _vp(d_word), _vp(d_pos), _vp(d_seg), _vp(d_gamma), _vp(d_beta),
                                      _vp(d_out), n_seq, T, hidden, float(eps), n_word, n_seg, _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def attention_f16_dev(d_qkv: int, d_lengths: int, d_out: int, n_seq: int, T: int, heads: int, head_dim: int, scale: float,
                      stream: int = 0):
    """softmax(scale Q K^T) V per head from a fused [n_seq, T, 3, heads, head_dim] fp16 QKV buffer into [n_seq, T, H]."""
    L = load_library()
    rc = L.hr_attention_f16_dev(_vp(d_qkv), _vp(d_lengths) if d_lengths else None, _vp(d_out), n_seq, T, heads, head_dim,
                                float(scale), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def linear_rows_f16_dev(d_x: int, x_fr: bool, d_w_packed: int, d_bias: int, d_out: int, rows: int, K: int, N: int, out_stride: int,
                        stream: int = 0):
    """out[r][:N] = x[r][:K] W^T + bias through the hand-written MFMA kernel (x row-major with naturally packed weights, or
    in fragment order with weights in accumulator k order: encoder_kernels.pack_natural / pack_accumulator_order)."""
    L = load_library()
    rc = L.hr_linear_rows_f16_dev(_vp(d_x), 1 if x_fr else 0, _vp(d_w_packed), _vp(d_bias), _vp(d_out), rows, K, N, out_stride,
                                  _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def attention_fr_f16_dev(d_qkv: int, d_lengths: int, d_out_fr: int, n_seq: int, T: int, heads: int, head_dim: int, scale: float,
                         stream: int = 0):
    """hr_attention_f16_dev with its output in fragment order (what hr_encoder_tail_f16_dev reads)."""
    L = load_library()
    rc = L.hr_attention_fr_f16_dev(_vp(d_qkv), _vp(d_lengths) if d_lengths else None, _vp(d_out_fr), n_seq, T, heads, head_dim,
                                   float(scale), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def encoder_tail_f16_dev(d_attn_fr: int, d_x: int, x_fr: bool, d_out: int, out_fr: bool, d_wstream: int, d_tables: int, rows: int,
                         hidden: int, intermediate: int, eps: float, gelu_erf: bool, stream: int = 0):
    """Everything of a post-LN layer after the attention in one launch (csrc/encoder_layer.h)."""
    L = load_library()
    rc = L.hr_encoder_tail_f16_dev(_vp(d_attn_fr), _vp(d_x), 1 if x_fr else 0, _vp(d_out), 1 if out_fr else 0, _vp(d_wstream),
                                   _vp(d_tables), rows, hidden, intermediate, float(eps), 1 if gelu_erf else 0,
                                   _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def attention_rows_f16_dev(d_q: int, q_seq_stride: int, q_token_stride: int, d_k: int, d_v: int, kv_seq_stride: int,
                           kv_token_stride: int, d_lengths: int, d_out: int, n_seq: int, T: int, n_queries: int, heads: int,
                           head_dim: int, scale: float, stream: int = 0):
    """The attention kernel with operands by pointer and stride (halves): the first n_queries tokens are the queries."""
    L = load_library()
    rc = L.hr_attention_rows_f16_dev(_vp(d_q), q_seq_stride, q_token_stride, _vp(d_k), _vp(d_v), kv_seq_stride, kv_token_stride,
                                     _vp(d_lengths) if d_lengths else None, _vp(d_out), n_seq, T, n_queries, heads, head_dim,
                                     float(scale), _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)


def rerank_linear_dev(d_ids: int, d_scores: int, d_methods: int, d_n: int, B: int, k_in: int, base_w: float,
                      method_bonus: float, recency_w: float, k_out: int, d_out_ids: int, d_out_scores: int,
                      d_out_orig: int, stream: int = 0, d_recency: int = 0):
    L = load_library()
    rc = L.hr_rerank_linear_dev(_vp(d_ids), _vp(d_scores), _vp(d_methods), _vp(d_n),
                                _vp(d_recency) if d_recency else None, B, k_in, base_w, method_bonus, recency_w,
                                k_out, _vp(d_out_ids), _vp(d_out_scores), _vp(d_out_orig),
                                _vp(stream) if stream else None)
    if rc != 0:
        _raise_global(L, rc)
