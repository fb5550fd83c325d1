"""Filter expressions evaluated on the GPU over HBM-resident scalar columns.

The reference hands `expr` to Milvus (indexing.py:503-525) which evaluates it server-side over the scalar fields of
the schema (indexing.py:191-225).  Here the payload columns live on the host (columns.py) and a COPY of the
filterable ones lives in HBM — uploaded the first time a field is filtered on, extended by what later appends
added — so that an expression becomes one `hr_filter_eval_dev` launch that writes the packed row mask the search
kernels take (no row-sized transfer in either direction; the first filtered request at 10M rows took 44 ms on the
host, the mask was re-uploaded with every search).

Strings (doc_id, chunk_id, timestamp) are compared through order-preserving 16-byte prefix keys; the rare rows whose
first 16 bytes equal the literal's come back in an "undecided" mask and are settled here on the full strings.
`filters.evaluate` (numpy, host) is the restatement the tests hold this against, bit for bit.
"""
from __future__ import annotations

import threading
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import _native as nat
from . import filters as _filters
from .columns import FLOAT_COLUMNS, INT_COLUMNS, StringColumn

_STRING_FIELDS = {"id": "id", "chunk_id": "id", "doc_id": "doc_id", "timestamp": "timestamp"}


def mask_bytes(n_rows: int) -> int:
    return 8 * ((n_rows + 63) // 64)


def unpack_mask(mask_u8, n_rows: int):
    """uint8 CUDA tensor of packed bits -> bool CUDA tensor [n_rows]."""
    import torch
    shifts = torch.arange(8, device=mask_u8.device, dtype=torch.uint8)
    return ((mask_u8[:, None] >> shifts[None, :]) & 1).reshape(-1)[:n_rows].bool()


def pack_mask(keep, n_rows: int):
    """bool CUDA tensor [n_rows] -> uint8 CUDA tensor of mask_bytes(n_rows) packed bits."""
    import torch
    padded = torch.zeros(mask_bytes(n_rows) * 8, dtype=torch.uint8, device=keep.device)
    padded[:n_rows] = keep.to(torch.uint8)
    weights = (1 << torch.arange(8, device=keep.device, dtype=torch.int32))
    return (padded.view(-1, 8).to(torch.int32) * weights[None, :]).sum(dim=1).to(torch.uint8)


class DeviceFilters:
    def __init__(self, columns, device: int):
        self.columns = columns                  # PayloadColumns, or None for a payload-free (synthetic) collection
        self.device = int(device)
        self._dev: Dict[str, Tuple[Any, int]] = {}   # column -> (device tensor with spare capacity, rows uploaded)
        self.stats = {"evaluations": 0, "undecided_rows": 0, "uploaded_bytes": 0}
        # one evaluation at a time: the column tensors are owned by `_dev` alone, so a second thread that re-uploads a
        # column (the dense and the sparse search of an uncoalesced retrieve() evaluate a new expression concurrently)
        # would free the tensor the first thread's kernel is about to read
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ columns in HBM
    def _tensor(self, name: str, host_rows, n: int, width: int = 1):
        """Device copy of a host column, valid for rows < n: uploads only what is not there yet."""
        import torch
        dev = torch.device("cuda", self.device)
        cur, have = self._dev.get(name, (None, 0))
        if cur is None or cur.shape[0] < n:
            cap = max(n, (cur.shape[0] * 3 // 2) if cur is not None else 0, 1024)
            shape = (cap, width) if width > 1 else (cap,)
            grown = torch.empty(shape, dtype=torch.int64 if host_rows(0, 0).dtype.kind in "iu" else torch.float32, device=dev)
            if cur is not None and have:
                grown[:have] = cur[:have]
            cur = grown
        if have < n:
            part = np.ascontiguousarray(host_rows(have, n))
            if part.dtype == np.uint64:
                part = part.view(np.int64)       # same bits; the kernel compares them as unsigned
            cur[have:n] = torch.from_numpy(part).to(dev)
            self.stats["uploaded_bytes"] += part.nbytes
        self._dev[name] = (cur, n)
        return cur

    def _column(self, field: str, n: int):
        if self.columns is None:   # payload-free rows: what the synthetic id encodes (chunk_index = row % 10)
            import torch
            cur, have = self._dev.get("chunk_index", (None, 0))
            if cur is None or have < n:
                cur = torch.arange(n, dtype=torch.int64, device=torch.device("cuda", self.device)) % 10
                self._dev["chunk_index"] = (cur, n)
            return cur
        if field in _STRING_FIELDS:
            col = self.columns[_STRING_FIELDS[field]]
            return self._tensor("key:" + _STRING_FIELDS[field], lambda a, b: col.keys()[a:b], n, width=2)
        col = self.columns[field]
        return self._tensor(field, lambda a, b: col.array()[a:b], n)

    # ------------------------------------------------------------------ expression -> terms
    def _terms(self, expr: str, n: int) -> Tuple[List["nat.FilterTerm"], List[Tuple[str, str, str]]]:
        terms, string_terms = [], []
        for field, op, value in _filters.parse(expr):
            if self.columns is None and field != "chunk_index":
                raise ValueError("this shard was bulk-ingested without payload columns: only chunk_index (= row % 10) "
                                 f"can be filtered on, not {[field]}")
            t = nat.FilterTerm()
            t.op = nat.FILTER_OPS[op]
            if field in INT_COLUMNS:
                if isinstance(value, str):
                    raise ValueError(f"field {field} is numeric; got string {value!r}")
                if isinstance(value, float):
                    t.kind, t.dval = nat.HR_COL_I64_VS_F64, float(value)
                else:
                    t.kind, t.ival = nat.HR_COL_I64, int(value)
            elif field in FLOAT_COLUMNS:
                if isinstance(value, str):
                    raise ValueError(f"field {field} is numeric; got string {value!r}")
                t.kind, t.fval = nat.HR_COL_F32, float(np.float32(value))
            elif field in _STRING_FIELDS:
                if not isinstance(value, str):
                    raise ValueError(f"field {field} is a string column; got {value!r}")
                t.kind = nat.HR_COL_STR16
                t.key[0], t.key[1] = StringColumn.key_of(value)
                string_terms.append((_STRING_FIELDS[field], op, value))
            else:
                raise ValueError(f"unknown filter field: {field}")
            t.col = self._column(field, n).data_ptr()
            terms.append(t)
        return terms, string_terms

    # ------------------------------------------------------------------ evaluation
    def evaluate(self, expr: Optional[str], n_rows: int, deleted: Optional[np.ndarray] = None, stream=None):
        """-> (uint8 CUDA tensor of mask_bytes(n_rows) packed bits, rows kept).  expr None = tombstones only."""
        with self._lock:
            return self._evaluate_locked(expr, n_rows, deleted, stream)

    def _evaluate_locked(self, expr, n_rows, deleted, stream):
        import torch
        dev = torch.device("cuda", self.device)
        terms, string_terms = self._terms(expr, n_rows) if expr else ([], [])
        if len(terms) > 16:
            raise ValueError("a filter expression may hold up to 16 terms")
        mask = torch.empty(mask_bytes(n_rows), dtype=torch.uint8, device=dev)
        und = torch.empty(mask_bytes(n_rows), dtype=torch.uint8, device=dev)
        counts = torch.zeros(2, dtype=torch.int32, device=dev)
        d_del = None
        if deleted is not None and deleted[:n_rows].any():
            bits = np.zeros(mask_bytes(n_rows), dtype=np.uint8)
            packed = np.packbits(deleted[:n_rows], bitorder="little")
            bits[: packed.size] = packed
            d_del = torch.from_numpy(bits).to(dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):   # the read-backs and the fix-ups below are ordered behind the kernel on ITS stream
            nat.filter_eval_dev(terms, n_rows, d_del.data_ptr() if d_del is not None else 0, mask.data_ptr(), und.data_ptr(),
                                counts.data_ptr(), st.cuda_stream)
            kept, undecided = (int(x) for x in counts.cpu().tolist())   # synchronises the stream
            self.stats["evaluations"] += 1
            if undecided:
                # rows that tie with a literal on the first 16 bytes: every other term has already passed them; settle the
                # string terms on the full strings (host) and switch the survivors on
                self.stats["undecided_rows"] += undecided
                rows = np.nonzero(np.unpackbits(und.cpu().numpy(), bitorder="little")[:n_rows])[0]
                ok = np.ones(rows.shape[0], dtype=bool)
                for col_name, op, value in string_terms:
                    ok &= self.columns[col_name].compare_rows(rows, op, value)
                rows = rows[ok]
                kept += int(rows.shape[0])
                for b in range(8):   # rows with the same bit position touch distinct bytes: plain indexed updates
                    sel = rows[(rows & 7) == b]
                    if sel.size:
                        idx = torch.from_numpy(sel >> 3).to(dev)
                        mask[idx] = mask[idx] | (1 << b)
                # the mask is cached and handed to searches on OTHER streams (the front's dense / sparse / hybrid streams,
                # the host view of _row_mask): only a finished mask may leave this function
                st.synchronize()
        return mask, kept
