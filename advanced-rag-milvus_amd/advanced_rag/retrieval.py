"""Hybrid retrieval engine: dense + sparse (+ optional domain) search, RRF
fusion, MMR, rerank — the reference's HybridRetriever surface on top of the
HBM shard store.

Behavioural source: reference src/advanced_rag/retrieval.py
  QueryClassifier            :22-67      RetrievalConfig          :70-101
  profiles                   :142-213    retrieve (timeout)       :215-247
  _retrieve_inner            :249-339    _search_*                :341-419
  _fuse_results (RRF k=60)   :421-491    _mmr_diversify           :493-516
  rerank                     :518-563    _build_filter_expression :565-632
  CrossEncoderReranker       :651-681
Where the index manager is the HBM one, the rank fusion runs in the HIP kernel
(csrc/fuse.h) through `index_manager.fuse_rank_lists`; for any other duck-typed
manager (the reference's test fakes) the same arithmetic runs on the host.
"""
from __future__ import annotations

import asyncio
import logging
import re
from dataclasses import dataclass
from datetime import datetime
from typing import Any, Callable, Dict, List, Optional, Sequence, Set, Tuple

import numpy as np

from .constants import RetrievalConstants
from .ranker import LearnedRanker

logger = logging.getLogger(__name__)

_TROUBLE_WORDS = ("error", "exception", "stack trace", "failed", "failure", "bug")
_SUMMARY_WORDS = ("summarize", "summary", "tl;dr", "overview")
_METHOD_ORDER = ("semantic", "sparse", "domain")
_METHODS_OF_MASK = tuple(tuple(m for b, m in enumerate(_METHOD_ORDER) if (mask >> b) & 1) for mask in range(8))


class QueryClassifier:
    """Substring rules -> retrieval profile name (checked in this order:
    troubleshooting, summary, faq, analysis, default)."""

    def __init__(self, max_faq_len: int = 80, long_query_len: int = 200):
        self.max_faq_len = max_faq_len
        self.long_query_len = long_query_len

    def classify(self, query: str) -> str:
        text = (query or "").strip()
        if not text:
            return "default"
        low = text.lower()
        if any(w in low for w in _TROUBLE_WORDS):
            return "troubleshooting"
        if any(w in low for w in _SUMMARY_WORDS):
            return "summary"
        if text.endswith("?") and len(text) <= self.max_faq_len:
            return "faq"
        if len(text) >= self.long_query_len:
            return "analysis"
        return "default"


@dataclass
class RetrievalConfig:
    hybrid_alpha: float = 0.7      # carried around but unused by the fusion math (as in the reference)
    top_k: int = 20
    rerank_top_k: int = 5
    enable_reranking: bool = True
    dense_weight: float = 0.7
    sparse_weight: float = 0.3
    enable_mmr: bool = False
    mmr_lambda: float = 0.7
    enable_learned_ranker: bool = False
    semantic_search_params: Optional[Dict] = None
    sparse_search_params: Optional[Dict] = None

    def __post_init__(self):
        if self.semantic_search_params is None:
            self.semantic_search_params = {"metric_type": "COSINE", "params": {"ef": 64}}
        if self.sparse_search_params is None:
            self.sparse_search_params = {"metric_type": "IP", "params": {"drop_ratio_search": 0.2}}


def rrf_rank_lists(lists: Sequence[Sequence[Any]], weights: Sequence[float], rrf_k: int = 60):
    """Host RRF over ranked id lists: returns [(id, float64 score, [list indices])] in
    fused order.  Same arithmetic and ordering as the HIP kernel."""
    table: Dict[Any, List] = {}
    for li, (ids, w) in enumerate(zip(lists, weights)):
        for rank, doc_id in enumerate(ids, start=1):
            ent = table.get(doc_id)
            if ent is None:
                ent = table[doc_id] = [0.0, []]
            ent[0] += (1.0 / (rrf_k + rank)) * w
            ent[1].append(li)
    fused = [(doc_id, ent[0], ent[1]) for doc_id, ent in table.items()]
    fused.sort(key=lambda t: t[1], reverse=True)  # stable: ties keep first-seen order
    return fused


class HybridRetriever:
    ALLOWED_FILTER_FIELDS: Set[str] = {"doc_id", "chunk_id", "domain_density", "timestamp", "entropy", "redundancy",
                                       "chunk_index", "token_count"}
    ALLOWED_OPERATORS: Set[str] = {"$gte", "$lte", "$gt", "$lt", "$eq", "$ne"}
    _OP_TEXT = {"$gte": ">=", "$lte": "<=", "$gt": ">", "$lt": "<", "$eq": "==", "$ne": "!="}
    DOMAIN_WEIGHT = 0.2
    RRF_K = 60

    def __init__(self, index_manager, config: Optional[RetrievalConfig] = None,
                 weight_adapter: Optional[Callable[[str], Tuple[float, float]]] = None,
                 classifier: Optional[QueryClassifier] = None,
                 profiles: Optional[Dict[str, RetrievalConfig]] = None,
                 learned_ranker: Optional[LearnedRanker] = None):
        self.index_manager = index_manager
        self.config = config or RetrievalConfig()
        self.weight_adapter = weight_adapter
        self.classifier = classifier or QueryClassifier()
        self.profiles: Dict[str, RetrievalConfig] = profiles or self._build_default_profiles(self.config)
        self.reranker = None
        self.learned_ranker: Optional[LearnedRanker] = learned_ranker

    # ------------------------------------------------------------------ profiles
    @staticmethod
    def _build_default_profiles(base: RetrievalConfig) -> Dict[str, RetrievalConfig]:
        cap = getattr(RetrievalConstants, "MAX_TOP_K", None)

        def clamp_k(v: int) -> int:
            v = int(v)
            return max(1, min(v, cap if cap is not None else v))

        def clamp_rerank(v: int) -> int:
            return max(1, min(int(v), clamp_k(v)))

        def derive(top_k: int, rerank_k: int, rerank: bool, mmr: bool, lam: float) -> RetrievalConfig:
            return RetrievalConfig(hybrid_alpha=base.hybrid_alpha, top_k=clamp_k(top_k),
                                   rerank_top_k=clamp_rerank(rerank_k), enable_reranking=rerank,
                                   dense_weight=base.dense_weight, sparse_weight=base.sparse_weight,
                                   enable_mmr=mmr, mmr_lambda=lam)

        return {
            "default": base,
            "faq": derive(min(base.top_k, 10), base.rerank_top_k, True, False, 0.7),
            "troubleshooting": derive(max(base.top_k, 30), 10, True, True, 0.5),
            "summary": derive(max(base.top_k, 40), 10, False, False, 0.7),
            "analysis": derive(max(base.top_k, 30), 10, True, True, 0.8),
        }

    def _pick_profile(self, query: str, hint: Optional[str]) -> str:
        try:
            if hint and hint in self.profiles:
                return hint
            if self.classifier:
                return self.classifier.classify(query) or "default"
        except Exception:
            pass
        return "default"

    # ------------------------------------------------------------------ retrieve
    async def retrieve(self, query: str, filters: Optional[Dict[str, Any]] = None, use_domain_index: bool = False,
                       domain: Optional[str] = None, profile_hint: Optional[str] = None) -> List[Dict[str, Any]]:
        budget = float(getattr(RetrievalConstants, "TIMEOUT_SECONDS", 0.3))
        try:
            return await asyncio.wait_for(
                self._retrieve_inner(query=query, filters=filters, use_domain_index=use_domain_index, domain=domain,
                                     profile_hint=profile_hint), timeout=budget)
        except asyncio.TimeoutError:
            logger.warning("HybridRetriever.retrieve timed out after %.3f seconds", budget)
            return []

    async def _retrieve_inner(self, query: str, filters: Optional[Dict[str, Any]] = None,
                              use_domain_index: bool = False, domain: Optional[str] = None,
                              profile_hint: Optional[str] = None) -> List[Dict[str, Any]]:
        profile = self._pick_profile(query, profile_hint)
        # The active profile becomes self.config for this request (and stays so
        # afterwards) exactly as reference retrieval.py:281-284 does.
        self.config = self.profiles.get(profile, self.config)

        dense_q = await self._get_semantic_embedding(query)
        sparse_q = await self._get_sparse_embedding(query)
        expr = self._build_filter_expression(filters) if filters else None

        fused = await self._retrieve_one_round(query, dense_q, sparse_q, expr) if not (use_domain_index and domain) else None
        if fused is not None:
            return self._tag_profile(fused, profile)[:self.config.top_k]

        searches = [self._search_semantic(dense_q, expr), self._search_sparse(sparse_q, expr)]
        if use_domain_index and domain:
            domain_q = await self._get_domain_embedding(query, domain)
            searches.append(self._search_domain(domain_q, expr))
        hit_lists = await asyncio.gather(*searches)

        self._adapt_weights(query)
        fused = await self._fuse_results_async(hit_lists[0], hit_lists[1], hit_lists[2] if len(hit_lists) > 2 else [])
        return self._tag_profile(fused, profile)[:self.config.top_k]

    def _adapt_weights(self, query: str) -> None:
        if self.weight_adapter:
            try:
                dw, sw = self.weight_adapter(query)
                dw = min(1.0, max(0.0, float(dw)))
                sw = min(1.0, max(0.0, float(sw)))
                if dw + sw > 0:
                    self.config.dense_weight, self.config.sparse_weight = dw, sw
            except Exception:
                pass

    @staticmethod
    def _tag_profile(fused: List[Dict[str, Any]], profile: str) -> List[Dict[str, Any]]:
        for hit in fused:
            meta = hit.get("metadata")
            if isinstance(meta, dict):
                meta.setdefault("retrieval_profile", profile)
            else:
                hit["retrieval_profile"] = profile
        return fused

    async def _retrieve_one_round(self, query: str, dense_q, sparse_q, expr: Optional[str]) -> Optional[List[Dict[str, Any]]]:
        """Both searches and their rank fusion as ONE request to an index manager that offers `hybrid_search` (the HBM
        manager's batching front: one device round per retrieve() instead of two, and only the fused top_k hits are
        ever formatted).  Same lists, same arithmetic, same hit dicts as the general path below; None = take that path
        (MMR needs the whole fused list; a manager without the entry point; a request it declined)."""
        one_round = getattr(self.index_manager, "hybrid_search", None)
        if one_round is None or self.config.enable_mmr:
            return None
        known = getattr(self.index_manager, "collections", None)
        if known is None or "sparse_index" not in known or "semantic_index" not in known:
            return None
        cfg = self.config
        saved = (cfg.dense_weight, cfg.sparse_weight)
        self._adapt_weights(query)   # the adapter sees only the query: applying it before the searches changes nothing
        try:
            ranked = await one_round(dense_q, sparse_q, top_k=cfg.top_k, filters=expr,
                                     weights=(cfg.dense_weight, cfg.sparse_weight), rrf_k=self.RRF_K,
                                     semantic_params=cfg.semantic_search_params, sparse_params=cfg.sparse_search_params)
        except Exception:  # pragma: no cover - the general path owns the error behaviour
            logger.exception("one-round hybrid search failed; taking the general path")
            ranked = None
        if ranked is None:
            cfg.dense_weight, cfg.sparse_weight = saved   # the general path applies the adapter itself
            return None
        now = None
        fused = []
        for hit, score, mask in ranked:     # _finish_fused_hit, inlined for the 20 hits of every request
            hit["method"] = _METHOD_ORDER[0] if mask & 1 else _METHOD_ORDER[1]
            hit["original_score"] = hit["score"]
            hit.pop("_row", None)
            hit["score"] = score
            hit["retrieval_methods"] = list(_METHODS_OF_MASK[mask & 7])
            meta = hit.get("metadata")
            if isinstance(meta, dict) and meta.get("timestamp") and "recency" not in meta:
                now = now or datetime.utcnow()
                try:
                    age_days = max(0.0, (now - datetime.fromisoformat(str(meta["timestamp"]))).total_seconds() / 86400.0)
                    meta["recency"] = float(1.0 / (1.0 + age_days))
                except Exception:
                    pass
            fused.append(hit)
        return fused

    async def _tagged_search(self, method: str, embedding, collection: str, top_k: int, filters, params):
        try:
            hits = await self.index_manager.search(query_embedding=embedding, collection_name=collection,
                                                   top_k=top_k, filters=filters, search_params=params)
        except Exception:
            return []  # a failing modality degrades to "no hits" (reference :355-358, :387-389, :411-413)
        for h in hits:
            h["method"] = method
            h["original_score"] = h["score"]
        return hits

    async def _search_semantic(self, embedding, filters: Optional[str]) -> List[Dict[str, Any]]:
        return await self._tagged_search("semantic", embedding, "semantic_index", self.config.top_k * 2, filters,
                                         self.config.semantic_search_params)

    async def _search_sparse(self, embedding, filters: Optional[str]) -> List[Dict[str, Any]]:
        known = getattr(self.index_manager, "collections", None)
        if known is not None and "sparse_index" not in known:
            return []
        return await self._tagged_search("sparse", embedding, "sparse_index", self.config.top_k * 2, filters,
                                         self.config.sparse_search_params)

    async def _search_domain(self, embedding, filters: Optional[str]) -> List[Dict[str, Any]]:
        return await self._tagged_search("domain", embedding, "domain_index", self.config.top_k, filters,
                                         self.config.semantic_search_params)

    # ------------------------------------------------------------------ fusion
    def _rank_fusion(self, hit_lists: Sequence[List[Dict]], weights: Sequence[float]):
        """-> [(position_of_payload (list idx, rank idx), float64 score, [list indices])] in fused order."""
        id_lists = [[h["id"] for h in hits] for hits in hit_lists]
        device_fuse = getattr(self.index_manager, "fuse_rank_lists", None)
        if device_fuse is not None:
            rows = [[h.get("_row") for h in hits] for hits in hit_lists]
            if all(r is not None for lst in rows for r in lst) and any(rows):
                try:
                    return device_fuse(rows, id_lists, list(weights), self.RRF_K)
                except Exception:  # pragma: no cover - fall through to the host arithmetic
                    logger.exception("device rank fusion failed; using host arithmetic")
        return rrf_rank_lists(id_lists, weights, self.RRF_K)

    def _fuse_results(self, semantic_results: List[Dict], sparse_results: List[Dict],
                      domain_results: Optional[List[Dict]] = None) -> List[Dict[str, Any]]:
        hit_lists = [semantic_results or [], sparse_results or [], domain_results or []]
        weights = [self.config.dense_weight, self.config.sparse_weight, self.DOMAIN_WEIGHT]
        return self._assemble_fused(hit_lists, self._rank_fusion(hit_lists, weights))

    async def _fuse_results_async(self, semantic_results, sparse_results, domain_results=None) -> List[Dict[str, Any]]:
        """_fuse_results for a caller on the event loop: with the HBM index manager the rank fusion of concurrent
        retrieve() calls shares one device launch (index_manager.fuse_rank_lists_async); same arithmetic, same result."""
        hit_lists = [semantic_results or [], sparse_results or [], domain_results or []]
        weights = [self.config.dense_weight, self.config.sparse_weight, self.DOMAIN_WEIGHT]
        device_fuse = getattr(self.index_manager, "fuse_rank_lists_async", None)
        ranked = None
        if device_fuse is not None:
            rows = [[h.get("_row") for h in hits] for hits in hit_lists]
            if all(r is not None for lst in rows for r in lst) and any(rows):
                try:
                    ranked = await device_fuse(rows, [[h["id"] for h in hits] for hits in hit_lists], list(weights), self.RRF_K)
                except Exception:  # pragma: no cover - fall through to the host arithmetic
                    logger.exception("device rank fusion failed; using host arithmetic")
        if ranked is None:
            ranked = self._rank_fusion(hit_lists, weights)
        return self._assemble_fused(hit_lists, ranked)

    def _assemble_fused(self, hit_lists, ranked) -> List[Dict[str, Any]]:
        # payload of an id = the hit from the semantic list if present (overwritten
        # by a later semantic duplicate), else the first sparse/domain hit
        payload: Dict[Any, Dict] = {}
        for hit in hit_lists[0]:
            payload[hit["id"]] = hit
        for hits in hit_lists[1:]:
            for hit in hits:
                payload.setdefault(hit["id"], hit)

        now = datetime.utcnow()
        fused = [self._finish_fused_hit(payload[doc_id], score, seen_in, now) for doc_id, score, seen_in in ranked]
        if self.config.enable_mmr and fused:
            return self._mmr_diversify(fused, self.config.top_k, self.config.mmr_lambda)
        return fused

    @staticmethod
    def _finish_fused_hit(hit: Dict[str, Any], score: float, seen_in, now) -> Dict[str, Any]:
        hit.pop("_row", None)  # manager-internal row number (device rank fusion); not part of the reference's hit dict
        hit["score"] = score
        hit["retrieval_methods"] = [m for i, m in enumerate(_METHOD_ORDER) if i in seen_in]
        meta = hit.get("metadata")
        if isinstance(meta, dict) and meta.get("timestamp") and "recency" not in meta:   # "" never parses: no recency
            try:
                age_days = max(0.0, (now - datetime.fromisoformat(str(meta["timestamp"]))).total_seconds() / 86400.0)
                meta["recency"] = float(1.0 / (1.0 + age_days))
            except Exception:
                pass
        return hit

    @staticmethod
    def _mmr_diversify(ranked: List[Dict[str, Any]], k: int, mmr_lambda: float) -> List[Dict[str, Any]]:
        """Greedy MMR on token-Jaccard similarity; the first strictly better candidate wins."""
        pool = [(r, set((r.get("content") or "").lower().split())) for r in ranked]
        chosen: List[Tuple[Dict[str, Any], set]] = []
        while pool and len(chosen) < k:
            best_i, best_val = None, -1e9
            for i, (r, toks) in enumerate(pool):
                if not chosen:
                    val = r["score"]
                else:
                    sim = max((len(toks & t2) / (len(toks | t2) or 1)) for _, t2 in chosen)
                    val = mmr_lambda * r["score"] - (1 - mmr_lambda) * sim
                if val > best_val:
                    best_i, best_val = i, val
            if best_i is None:  # every candidate scored <= -1e9; the reference would append None here
                break
            chosen.append(pool.pop(best_i))
        return [r for r, _ in chosen]

    # ------------------------------------------------------------------ rerank
    async def rerank(self, query: str, results: List[Dict[str, Any]], top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        if not self.config.enable_reranking or not results:
            return results[:top_k] if top_k else results
        top_k = top_k or self.config.rerank_top_k
        if self.learned_ranker and self.config.enable_learned_ranker:
            new_scores = await self.learned_ranker.score(query, results)
        elif self.reranker:
            new_scores = await self.reranker.score([(query, r["content"]) for r in results])
        else:
            # the reference's placeholder: retrieval score + N(0, 0.01) (retrieval.py:549-553)
            # (one vector draw: the same values, in the same order, as a scalar draw per result from numpy's global stream)
            new_scores = [r["score"] + e for r, e in zip(results, np.random.normal(0, 0.01, len(results)).tolist())]
        for r, s in zip(results, new_scores):
            r["rerank_score"] = s
            r["original_retrieval_score"] = r["score"]
            r["score"] = s
        results.sort(key=lambda r: r["rerank_score"], reverse=True)
        return results[:top_k]

    # ------------------------------------------------------------------ filters
    @staticmethod
    def _quote(value: str) -> str:
        return '"' + value.replace("\\", "\\\\").replace('"', '\\"') + '"'

    def _build_filter_expression(self, filters: Dict[str, Any]) -> Optional[str]:
        """{"doc_id": "d1", "entropy": {"$gte": 0.2}} -> 'doc_id == "d1" and entropy >= 0.2'."""
        terms: List[str] = []
        for field, cond in filters.items():
            if field not in self.ALLOWED_FILTER_FIELDS:
                logger.warning("Invalid filter field attempted: %s", field)
                raise ValueError(f"Invalid filter field: {field}")
            if not re.match(r"^[a-zA-Z_][a-zA-Z0-9_]*$", field):
                raise ValueError(f"Invalid field name format: {field}")
            if isinstance(cond, dict):
                for op, operand in cond.items():
                    if op not in self.ALLOWED_OPERATORS:
                        logger.warning("Invalid operator attempted: %s", op)
                        raise ValueError(f"Invalid operator: {op}")
                    if not isinstance(operand, (int, float, str, bool)):
                        raise ValueError(f"Invalid value type for {field}: {type(operand)}")
                    rhs = self._quote(operand) if isinstance(operand, str) else f"{operand}"
                    terms.append(f"{field} {self._OP_TEXT[op]} {rhs}")
            elif isinstance(cond, str):
                terms.append(f"{field} == {self._quote(cond)}")
            elif isinstance(cond, (int, float, bool)):
                terms.append(f"{field} == {cond}")
            else:
                raise ValueError(f"Unsupported value type for {field}: {type(cond)}")
        return " and ".join(terms) if terms else None

    # ------------------------------------------------------------------ embeddings
    async def _get_semantic_embedding(self, text: str):
        return await self.index_manager._generate_semantic_embedding(text)

    async def _get_sparse_embedding(self, text: str):
        return await self.index_manager._generate_sparse_embedding(text)

    async def _get_domain_embedding(self, text: str, domain: str):
        return await self.index_manager._generate_domain_embedding(text, domain)


class CrossEncoderReranker:
    """Pair scorer plugged into HybridRetriever.reranker.  With `model` set
    (anything exposing predict(pairs) -> array, e.g. encoders.CrossEncoderModel)
    it scores with the model; without one it returns the reference's dummy
    0.5 + 0.1*N(0,1) scores (retrieval.py:680-681)."""

    def __init__(self, model_name: str = "cross-encoder/ms-marco-MiniLM-L-6-v2", model=None):
        self.model_name = model_name
        self.model = model

    async def score(self, pairs: List[tuple]) -> List[float]:
        if self.model:
            return np.asarray(self.model.predict(pairs)).tolist()
        return [0.5 + np.random.randn() * 0.1 for _ in pairs]
