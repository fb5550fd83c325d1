"""Deterministic linear re-scorer used by HybridRetriever.rerank when
`enable_learned_ranker` is set.

Behaviour follows reference src/advanced_rag/ranker.py: features :57-78,
score rule :109-125 (base_weight*score + method_bonus*len(retrieval_methods)
+ recency_weight*metadata.recency), feedback collection :80-107.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional


@dataclass
class LearnedRankerConfig:
    base_weight: float = 1.0
    method_bonus: float = 0.1
    diversity_penalty: float = 0.0
    recency_weight: float = 0.0


@dataclass
class TrainingExample:
    query: str
    doc_id: str
    features: Dict[str, float]
    label: float


class LearnedRanker:
    def __init__(self, config: Optional[LearnedRankerConfig] = None) -> None:
        self.config = config or LearnedRankerConfig()
        self.training_examples: List[TrainingExample] = []

    @staticmethod
    def featurize(result: Dict[str, Any]) -> Dict[str, float]:
        meta = result.get("metadata") or {}
        recency = float(meta.get("recency", 0.0)) if isinstance(meta, dict) else 0.0
        return {
            "base_score": float(result.get("score", 0.0)),
            "method_count": float(len(result.get("retrieval_methods") or [])),
            "recency": recency,
        }

    def _linear(self, f: Dict[str, float]) -> float:
        c = self.config
        return float(c.base_weight * f["base_score"] + c.method_bonus * f["method_count"]
                     + c.recency_weight * f.get("recency", 0.0))

    async def score(self, query: str, results: List[Dict[str, Any]]) -> List[float]:
        return [self._linear(self.featurize(r)) for r in results]

    def update_from_feedback(self, query: str, results: List[Dict[str, Any]], feedback: List[Dict[str, Any]]) -> None:
        labels = {fb["id"]: float(fb.get("label", 0.0)) for fb in feedback}
        for r in results:
            rid = r.get("id")
            if rid in labels:
                self.training_examples.append(
                    TrainingExample(query=query, doc_id=rid, features=self.featurize(r), label=labels[rid]))
