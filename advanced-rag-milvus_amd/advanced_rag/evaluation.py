"""Per-request retrieval metrics.  OUT OF THE HOT PATH (≈0.5 ms of host float
math on <=5 results, reference evaluation.py:92-153); kept because
AdvancedRAGPipeline.retrieve returns (results, EvaluationMetrics)."""
from __future__ import annotations

from collections import deque
from dataclasses import asdict, dataclass
from typing import Any, Dict, List, Optional

import math

import numpy as np


@dataclass
class EvaluationMetrics:
    retrieval_precision: float
    retrieval_recall: float
    mean_reciprocal_rank: float
    ndcg_at_k: float
    hallucination_risk: float
    faithfulness_score: float
    coverage_score: float
    diversity_score: float
    confidence_score: float
    uncertainty_estimate: float

    def to_dict(self) -> Dict[str, float]:
        return asdict(self)


@dataclass
class DriftReport:
    drift_detected: bool
    drift_magnitude: float
    embedding_divergence: float
    distribution_shift: float
    temporal_decay: float
    affected_queries: List[str]
    recommendations: List[str]


class RAGEvaluator:
    def __init__(self, drift_threshold: float = 0.15, hallucination_threshold: float = 0.2):
        self.drift_threshold = drift_threshold
        self.hallucination_threshold = hallucination_threshold
        self.query_embeddings_history = deque(maxlen=1000)
        self.score_distributions_history = deque(maxlen=1000)

    async def evaluate_retrieval(self, query: str, results: List[Dict[str, Any]],
                                 context: Optional[Dict[str, Any]] = None) -> EvaluationMetrics:
        # plain Python arithmetic: the lists hold <= rerank_top_k entries, where a numpy call costs more than the sum it takes
        truth = set((context or {}).get("relevant_doc_ids", []) or [])
        ids = [r.get("id") for r in results]
        hits = [i in truth for i in ids] if truth else [False] * len(ids)
        precision = sum(hits) / len(ids) if ids and truth else 0.0
        recall = sum(hits) / len(truth) if truth else 0.0
        mrr = next((1.0 / (r + 1) for r, h in enumerate(hits) if h), 0.0)
        dcg = sum(1.0 / math.log2(r + 2) for r, h in enumerate(hits) if h)
        ideal = sum(1.0 / math.log2(r + 2) for r in range(min(len(truth), len(ids))))
        scores = [float(r.get("score", 0.0)) for r in results]
        q_tokens = set(query.lower().split())
        token_sets = [set((r.get("content") or "").lower().split()) for r in results]
        seen = set().union(*token_sets) if token_sets else set()
        coverage = len(q_tokens & seen) / len(q_tokens) if q_tokens else 0.0
        sims = [len(a & b) / (len(a | b) or 1) for i, a in enumerate(token_sets) for b in token_sets[i + 1:]]
        diversity = 1.0 - math.fsum(sims) / len(sims) if sims else 0.0
        n = len(scores)
        confidence = math.fsum(scores) / n if n else 0.0
        uncertainty = math.sqrt(math.fsum((x - confidence) ** 2 for x in scores) / n) if n else 1.0   # population std, as ndarray.std
        top = max(scores) if n else 0.0
        risk = min(1.0, max(0.0, 0.25 * (1 - min(1.0, confidence)) + 0.2 * (1 - diversity) + 0.3 * (1 - min(1.0, top))
                            + 0.25 * (1 - coverage))) if results else 1.0
        if n:
            e = [math.exp(x - top) for x in scores]
            z = math.fsum(e) + 1e-12
            self.score_distributions_history.append(np.array([x / z for x in e], dtype=np.float64))
        return EvaluationMetrics(precision, recall, mrr, float(dcg / ideal) if ideal else 0.0, risk,
                                 float(1.0 - risk), float(coverage), float(diversity), confidence, uncertainty)

    async def detect_drift(self, queries: List[str], index_manager) -> Dict[str, Any]:
        embs = [np.asarray(await index_manager._generate_semantic_embedding(q), dtype=np.float64) for q in queries]
        if not embs:
            return {"drift_detected": False, "drift_magnitude": 0.0}
        cur = np.mean(embs, axis=0)
        prev = np.mean(self.query_embeddings_history, axis=0) if self.query_embeddings_history else cur
        den = (np.linalg.norm(cur) * np.linalg.norm(prev)) or 1.0
        mag = float(1.0 - np.dot(cur, prev) / den)
        self.query_embeddings_history.extend(embs)
        return {"drift_detected": mag > self.drift_threshold, "drift_magnitude": mag}
