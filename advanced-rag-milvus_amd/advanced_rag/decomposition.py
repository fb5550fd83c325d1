"""Query decomposition for plan_and_execute (reference decomposition.py; split on ' and ')."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List


@dataclass
class DecompositionResult:
    sub_queries: List[str]
    strategy: str


class QueryDecomposer:
    def decompose(self, query: str) -> DecompositionResult:
        parts = [p.strip() for p in (query or "").split(" and ") if p.strip()]
        if len(parts) > 1:
            return DecompositionResult(sub_queries=parts, strategy="conjunction_split")
        return DecompositionResult(sub_queries=[query], strategy="single")
