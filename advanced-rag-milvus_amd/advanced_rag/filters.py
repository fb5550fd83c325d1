"""Filter expressions -> row bitmask.

HybridRetriever._build_filter_expression (reference retrieval.py:573-632) emits
Milvus boolean expressions of the form
    field OP value and field OP value ...
with OP in {>=, <=, >, <, ==, !=}, values int/float/bool literals or
double-quoted strings with \\ and \" escapes, over the scalar fields of the
collection schema (indexing.py:191-225).  Milvus evaluates them server-side;
here they become a per-row predicate over host-side columns and are handed to
the kernels as a packed bitmask (bit r%8 of byte r/8).
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Tuple

import numpy as np

_OPS = (">=", "<=", "==", "!=", ">", "<")
NUMERIC_FIELDS = {"chunk_index": np.int64, "token_count": np.int64, "entropy": np.float32,
                  "redundancy": np.float32, "domain_density": np.float32}
STRING_FIELDS = ("id", "chunk_id", "doc_id", "timestamp")


def _split_terms(expr: str) -> List[str]:
    """Split on ' and ' that is not inside a double-quoted string."""
    terms, buf, in_str, i = [], [], False, 0
    while i < len(expr):
        ch = expr[i]
        if in_str:
            buf.append(ch)
            if ch == "\\" and i + 1 < len(expr):
                buf.append(expr[i + 1])
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            buf.append(ch)
        elif expr.startswith(" and ", i):
            terms.append("".join(buf))
            buf = []
            i += 4
        else:
            buf.append(ch)
        i += 1
    if in_str:
        raise ValueError(f"unterminated string in filter expression: {expr!r}")
    terms.append("".join(buf))
    return [t.strip() for t in terms if t.strip()]


def _unquote(tok: str) -> str:
    out, i = [], 1
    while i < len(tok) - 1:
        if tok[i] == "\\" and i + 1 < len(tok) - 1:
            out.append(tok[i + 1])
            i += 2
        else:
            out.append(tok[i])
            i += 1
    return "".join(out)


_TERM = re.compile(r"^\s*([A-Za-z_]\w*)\s*(>=|<=|==|!=|>|<)\s*(.+?)\s*$", re.S)


def parse(expr: str) -> List[Tuple[str, str, Any]]:
    """-> [(field, op, python value)].  The operator is the one right after the field name: a quoted value
    may itself contain ' >= ', ' and ' or escaped quotes (doc_id == "a >= b")."""
    parsed = []
    for term in _split_terms(expr):
        m = _TERM.match(term)
        if m is None:
            raise ValueError(f"cannot parse filter term: {term!r}")
        field, op, raw = m.group(1), m.group(2), m.group(3)
        if raw.startswith('"') and raw.endswith('"') and len(raw) >= 2:
            value: Any = _unquote(raw)
        elif raw in ("True", "true"):
            value = True
        elif raw in ("False", "false"):
            value = False
        else:
            try:
                value = int(raw)
            except ValueError:
                try:
                    value = float(raw)
                except ValueError:
                    raise ValueError(f"bad literal in filter term: {term!r}")
        parsed.append((field, op, value))
    return parsed


def _compare(col: np.ndarray, op: str, value: Any) -> np.ndarray:
    if op == "==":
        return col == value
    if op == "!=":
        return col != value
    if op == ">=":
        return col >= value
    if op == "<=":
        return col <= value
    if op == ">":
        return col > value
    return col < value


def evaluate(expr: str, columns: Dict[str, np.ndarray], n_rows: int) -> np.ndarray:
    """Boolean row predicate (length n_rows) of a conjunctive expression."""
    keep = np.ones(n_rows, dtype=bool)
    for field, op, value in parse(expr):
        if field not in columns:
            raise ValueError(f"unknown filter field: {field}")
        col = columns[field]
        if col.dtype.kind in "US":
            if not isinstance(value, str):
                raise ValueError(f"field {field} is a string column; got {value!r}")
        elif isinstance(value, str):
            raise ValueError(f"field {field} is numeric; got string {value!r}")
        keep &= _compare(col, op, value)
    return keep


def pack(keep: np.ndarray) -> np.ndarray:
    return np.packbits(keep.astype(np.uint8), bitorder="little")
