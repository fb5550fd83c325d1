"""Document diagnostics record + a cheap analyzer.  OUT OF THE HOT PATH: text
heuristics (reference diagnostics.py:17-45 for the record; its scoring is not
restated).  ingest_documents only needs the four numbers chunk rows carry."""
from __future__ import annotations

import math
from collections import Counter
from dataclasses import dataclass, field
from typing import Dict, Set

from .chunking import tokenize


@dataclass
class DiagnosticMetrics:
    information_entropy: float
    redundancy_score: float
    domain_density: float
    vocabulary_diversity: float
    semantic_coherence: float
    avg_sentence_complexity: float
    token_distribution: Dict[str, float] = field(default_factory=dict)
    n_gram_redundancy: Dict[int, float] = field(default_factory=dict)
    domain_terms: Set[str] = field(default_factory=set)

    def to_dict(self) -> Dict:
        return {k: getattr(self, k) for k in ("information_entropy", "redundancy_score", "domain_density",
                                              "vocabulary_diversity", "semantic_coherence",
                                              "avg_sentence_complexity", "n_gram_redundancy")}


class DocumentDiagnostics:
    def analyze_document(self, text: str) -> DiagnosticMetrics:
        toks = tokenize(text or "")
        if not toks:
            return DiagnosticMetrics(0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        counts = Counter(toks)
        n = len(toks)
        ent = -sum(c / n * math.log2(c / n) for c in counts.values())
        ent = ent / math.log2(len(counts)) if len(counts) > 1 else 0.0
        diversity = len(counts) / n
        long_terms = {t for t in counts if len(t) >= 9}
        sentences = max(1, sum(text.count(p) for p in ".!?"))
        return DiagnosticMetrics(information_entropy=float(ent), redundancy_score=float(1.0 - diversity),
                                 domain_density=float(sum(counts[t] for t in long_terms) / n),
                                 vocabulary_diversity=float(diversity), semantic_coherence=0.5,
                                 avg_sentence_complexity=float(n / sentences), domain_terms=long_terms)
