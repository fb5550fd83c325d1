"""Entity/topic tags attached at ingest (reference semantic_enrichment.py; string heuristics, out of the hot path)."""
from __future__ import annotations

import re
from collections import Counter
from dataclasses import dataclass
from typing import List


@dataclass
class EnrichmentResult:
    entities: List[str]
    topics: List[str]


class SemanticEnricher:
    def enrich(self, text: str) -> EnrichmentResult:
        entities = sorted(set(re.findall(r"\b[A-Z][a-zA-Z0-9]+\b", text or "")))
        words = [w for w in re.findall(r"\b\w+\b", (text or "").lower()) if len(w) > 3]
        return EnrichmentResult(entities=entities, topics=[w for w, _ in Counter(words).most_common(5)])
