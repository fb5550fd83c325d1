"""Deterministic query expansion run BEFORE embedding (so parity harnesses must
apply it too): reference query_rewriting.py:41-60."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

_EXPANSIONS = (("rag", "retrieval augmented generation"), ("llm", "large language model"))


@dataclass
class QueryRewriterConfig:
    enable_expansion: bool = True


class QueryRewriter:
    def __init__(self, config: Optional[QueryRewriterConfig] = None) -> None:
        self.config = config or QueryRewriterConfig()

    def rewrite(self, query: str, context: Optional[Dict[str, Any]] = None) -> str:
        if not self.config.enable_expansion or not query:
            return query
        q = query.strip()
        low = q.lower()
        for abbrev, full in _EXPANSIONS:
            if abbrev in low and full not in low:
                return f"{q} ({full})"
        return q
