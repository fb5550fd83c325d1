"""Append-only payload columns of a collection, keyed by global row.

The reference keeps the scalar fields of its schema (indexing.py:191-225: id / chunk_id, doc_id, content,
chunk_index, token_count, entropy, redundancy, domain_density, timestamp, metadata_json) inside Milvus; here they
stay on the host — the GPU only ever sees row numbers — in arrow-style columns:

  * numeric fields: one growable numpy array each (int64 / float32, amortised O(batch) appends);
  * string fields: one growable UTF-8 byte buffer + int64 offsets each — no Python object per row, so ten million
    rows of ids and doc ids cost a few hundred MB instead of GBs of str objects, and nothing is rebuilt after an append.

For filter expressions (filters.py / device_filters.py) a string column can hand out an ORDER-PRESERVING 16-byte
prefix key per row (two big-endian uint64 words of the zero-padded UTF-8 bytes; UTF-8 byte order is code-point
order): `key < key(v)` decides `s < v` for every row whose first 16 bytes differ from v's, and only the rows that tie
on the prefix need the full strings.  Keys are built lazily, per column, on first use, and extended by later appends.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

STRING_COLUMNS = ("id", "doc_id", "content", "timestamp", "metadata_json")
INT_COLUMNS = ("chunk_index", "token_count")
FLOAT_COLUMNS = ("entropy", "redundancy", "domain_density")
KEY_BYTES = 16


def _grown(arr: np.ndarray, need: int) -> np.ndarray:
    if need <= arr.shape[0]:
        return arr
    cap = max(need, arr.shape[0] + arr.shape[0] // 2, 1024)
    out = np.empty(cap, dtype=arr.dtype)
    out[: arr.shape[0]] = arr
    return out


class NumericColumn:
    def __init__(self, dtype):
        self._a = np.empty(0, dtype=dtype)
        self._n = 0

    def __len__(self) -> int:
        return self._n

    def append(self, v) -> None:
        self._a = _grown(self._a, self._n + 1)
        self._a[self._n] = v
        self._n += 1

    def extend(self, values) -> None:
        vals = np.asarray(values if isinstance(values, np.ndarray) else list(values), dtype=self._a.dtype)
        self._a = _grown(self._a, self._n + vals.shape[0])
        self._a[self._n: self._n + vals.shape[0]] = vals
        self._n += vals.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.array()[i].tolist()
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._a[i].item()      # Python int / float (a float32 value as the float that equals it)

    def __iter__(self) -> Iterator:
        return iter(self.array().tolist())

    def __eq__(self, other) -> bool:
        return list(self) == list(other)

    def array(self) -> np.ndarray:
        """The column as a numpy view (no copy): what the filters compare against."""
        return self._a[: self._n]

    def tolist(self) -> list:
        return self.array().tolist()

    @property
    def nbytes(self) -> int:
        return self._a.nbytes


class StringColumn:
    def __init__(self):
        self._buf = np.empty(0, dtype=np.uint8)
        self._used = 0
        self._off = np.zeros(1, dtype=np.int64)   # _off[r] .. _off[r + 1] = bytes of row r
        self._n = 0
        self._keys: Optional[np.ndarray] = None   # [cap, 2] uint64 prefix keys, valid for rows < _n_keyed
        self._n_keyed = 0

    def __len__(self) -> int:
        return self._n

    def append(self, v) -> None:
        self.extend((v,))

    def extend(self, values: Iterable) -> None:
        enc = [str(v).encode("utf-8") for v in values]
        if not enc:
            return
        lens = np.fromiter((len(b) for b in enc), dtype=np.int64, count=len(enc))
        total = int(lens.sum())
        self._buf = _grown(self._buf, self._used + total)
        self._buf[self._used: self._used + total] = np.frombuffer(b"".join(enc), dtype=np.uint8)
        self._off = _grown(self._off, self._n + len(enc) + 1)
        np.cumsum(lens, out=self._off[self._n + 1: self._n + len(enc) + 1])
        self._off[self._n + 1: self._n + len(enc) + 1] += self._used
        self._used += total
        self._n += len(enc)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._buf[self._off[i]: self._off[i + 1]].tobytes().decode("utf-8")

    def __iter__(self) -> Iterator[str]:
        whole = self._buf[: self._used].tobytes()
        off = self._off
        return (whole[off[i]: off[i + 1]].decode("utf-8") for i in range(self._n))

    def __eq__(self, other) -> bool:
        return list(self) == list(other)

    def tolist(self) -> List[str]:
        return list(self)

    def as_str_array(self) -> np.ndarray:
        """numpy unicode array (fixed width = the longest row): only for small collections and tests."""
        return np.asarray(self.tolist(), dtype=str) if self._n else np.zeros(0, dtype=str)

    @property
    def nbytes(self) -> int:
        return self._buf.nbytes + self._off.nbytes + (self._keys.nbytes if self._keys is not None else 0)

    # -- order-preserving prefix keys --------------------------------------------------------------------------
    @staticmethod
    def key_of(value: str) -> Tuple[int, int]:
        b = value.encode("utf-8")[:KEY_BYTES].ljust(KEY_BYTES, b"\0")
        return int.from_bytes(b[:8], "big"), int.from_bytes(b[8:], "big")

    def keys(self) -> np.ndarray:
        """[n, 2] uint64: big-endian words of the first 16 bytes of every row, zero padded."""
        n = self._n
        if self._keys is None:
            self._keys, self._n_keyed = np.empty((max(n, 1024), 2), dtype=np.uint64), 0
        if self._keys.shape[0] < n:
            grown = np.empty((max(n, self._keys.shape[0] * 3 // 2), 2), dtype=np.uint64)
            grown[: self._n_keyed] = self._keys[: self._n_keyed]
            self._keys = grown
        step = 1 << 20       # bounded temporaries: 16 MB of gathered bytes per piece
        for a in range(self._n_keyed, n, step):
            b = min(n, a + step)
            start, end = self._off[a:b], self._off[a + 1: b + 1]
            pos = start[:, None] + np.arange(KEY_BYTES, dtype=np.int64)[None, :]
            valid = pos < end[:, None]
            if self._used:
                raw = self._buf[np.minimum(pos, self._used - 1)]
                raw[~valid] = 0
            else:
                raw = np.zeros((b - a, KEY_BYTES), dtype=np.uint8)
            self._keys[a:b] = np.ascontiguousarray(raw).view(">u8").astype(np.uint64)
        self._n_keyed = n
        return self._keys[:n]

    def compare_rows(self, rows: np.ndarray, op: str, value: str) -> np.ndarray:
        """Exact `row OP value` for the given rows (the ones a prefix key could not decide)."""
        vb = value.encode("utf-8")
        out = np.empty(len(rows), dtype=bool)
        whole = self._buf
        for i, r in enumerate(np.asarray(rows, dtype=np.int64).tolist()):
            s = whole[self._off[r]: self._off[r + 1]].tobytes()
            out[i] = {"==": s == vb, "!=": s != vb, "<": s < vb, "<=": s <= vb, ">": s > vb, ">=": s >= vb}[op]
        return out


class PayloadColumns:
    """dict-like: columns["id"][row], columns["entropy"].array(), len(columns["id"])."""

    def __init__(self):
        self._c: Dict[str, Any] = {k: StringColumn() for k in STRING_COLUMNS}
        self._c.update({k: NumericColumn(np.int64) for k in INT_COLUMNS})
        self._c.update({k: NumericColumn(np.float32) for k in FLOAT_COLUMNS})

    def __getitem__(self, name: str):
        return self._c["id" if name == "chunk_id" else name]

    def __contains__(self, name: str) -> bool:
        return name == "chunk_id" or name in self._c

    def __iter__(self):
        return iter(self._c)

    def items(self):
        return self._c.items()

    def keys(self):
        return self._c.keys()

    def __len__(self) -> int:
        return len(self._c)

    @property
    def n_rows(self) -> int:
        return len(self._c["id"])

    @property
    def nbytes(self) -> int:
        return sum(c.nbytes for c in self._c.values())

    def filter_columns(self) -> Dict[str, np.ndarray]:
        """What filters.evaluate takes: numpy arrays per field (string fields as unicode arrays — fine for small
        collections; large ones go through device_filters, which never materialises them)."""
        out = {k: self._c[k].array() for k in INT_COLUMNS + FLOAT_COLUMNS}
        for k in ("id", "doc_id", "timestamp"):
            out[k] = self._c[k].as_str_array()
        out["chunk_id"] = out["id"]
        return out
