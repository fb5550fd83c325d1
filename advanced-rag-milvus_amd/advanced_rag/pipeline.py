"""AdvancedRAGPipeline / PipelineConfig — the orchestrator surface of the
reference (src/advanced_rag/pipeline.py:26-118 types and wiring, :120-215
ingest_documents, :217-309 retrieve, :311-348 plan_and_execute, :365-412
telemetry) over the HBM shard store.

Kept quirks: `rerank_top_k` and `dense_weight` are
not forwarded to RetrievalConfig; `hybrid_alpha` is carried but unused by the
fusion; with reranking on and no reranker plugged in, rerank adds N(0,0.01)
noise exactly like the reference.  Build-only knobs are extra keyword
arguments (`dtype`, `device`, `semantic_dim`, ...), never new required fields.
"""
from __future__ import annotations

from dataclasses import dataclass
from datetime import datetime
from enum import Enum
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .chunking import AdaptiveChunker, ChunkMetadata
from .compliance import AuditLog, ComplianceManager
from .constants import APIConstants
from .decomposition import DecompositionResult, QueryDecomposer
from .diagnostics import DiagnosticMetrics, DocumentDiagnostics
from .evaluation import EvaluationMetrics, RAGEvaluator
from .indexing import IndexType, MilvusIndexManager
from .query_rewriting import QueryRewriter
from .ranker import LearnedRanker
from .retrieval import HybridRetriever, RetrievalConfig
from .semantic_enrichment import SemanticEnricher


class PipelineStage(Enum):
    DIAGNOSTICS = "diagnostics"
    CHUNKING = "chunking"
    INDEXING = "indexing"
    RETRIEVAL = "retrieval"
    RERANKING = "reranking"
    EVALUATION = "evaluation"


@dataclass
class PipelineConfig:
    target_latency_ms: float = 80.0
    enable_hierarchical_index: bool = True
    enable_sharding: bool = True
    hybrid_alpha: float = 0.7
    top_k: int = 20
    rerank_top_k: int = 5
    enable_reranking: bool = True
    min_relevance_score: float = 0.65
    max_hallucination_risk: float = 0.15
    enable_audit_logging: bool = True
    enable_versioning: bool = True
    retention_days: int = 90


@dataclass
class RetrievalResult:
    content: str
    chunk_id: str
    score: float
    metadata: ChunkMetadata  # a plain dict at run time, as in the reference
    retrieval_method: str
    latency_ms: float
    audit_trail: Optional[AuditLog] = None


def _ms_since(t0: datetime) -> float:
    return (datetime.now() - t0).total_seconds() * 1000.0


class AdvancedRAGPipeline:
    def __init__(self, milvus_host: str = "localhost", milvus_port: int = 19530,
                 config: Optional[PipelineConfig] = None, connect_to_milvus: bool = True,
                 query_rewriter: Optional[QueryRewriter] = None, **shard_options):
        self.config = config or PipelineConfig()
        self.diagnostics = DocumentDiagnostics()
        self.chunker = AdaptiveChunker()
        self.enricher = SemanticEnricher()
        self.decomposer = QueryDecomposer()
        self.query_rewriter = query_rewriter or QueryRewriter()
        self.index_manager = MilvusIndexManager(host=milvus_host, port=milvus_port,
                                                enable_sharding=self.config.enable_sharding,
                                                connect=connect_to_milvus, **shard_options)
        self.retriever = HybridRetriever(
            index_manager=self.index_manager,
            config=RetrievalConfig(hybrid_alpha=self.config.hybrid_alpha, top_k=self.config.top_k,
                                   enable_reranking=self.config.enable_reranking),
            learned_ranker=LearnedRanker())
        self.evaluator = RAGEvaluator()
        self.compliance = ComplianceManager(enable_audit=self.config.enable_audit_logging,
                                            enable_versioning=self.config.enable_versioning)
        self.stage_latencies: Dict[PipelineStage, List[float]] = {stage: [] for stage in PipelineStage}

    # ------------------------------------------------------------------ ingest
    async def ingest_documents(self, documents: List[Dict[str, Any]], domain: Optional[str] = None) -> Dict[str, Any]:
        t_all = datetime.now()
        report: Dict[str, Any] = {"total_documents": len(documents), "chunks_created": 0, "diagnostic_metrics": [],
                                  "indexing_summary": {}}
        chunks = []
        for n, doc in enumerate(documents):
            t0 = datetime.now()
            metrics = self.diagnostics.analyze_document(doc["text"])
            self._record_latency(PipelineStage.DIAGNOSTICS, _ms_since(t0))
            report["diagnostic_metrics"].append({"document_id": doc.get("id", n),
                                                 "entropy": metrics.information_entropy,
                                                 "redundancy": metrics.redundancy_score,
                                                 "domain_density": metrics.domain_density})
            report.setdefault("data_quality", []).append(self._assess_data_quality(doc, metrics))
            t0 = datetime.now()
            doc_meta = doc.get("metadata", {})
            doc_chunks = self.chunker.chunk_document(
                text=doc["text"], diagnostics=metrics,
                metadata={"doc_id": doc.get("id", n), "source": doc_meta.get("source", "unknown"),
                          "timestamp": datetime.now().isoformat(), **doc_meta})
            self._record_latency(PipelineStage.CHUNKING, _ms_since(t0))
            for ch in doc_chunks:
                tags = self.enricher.enrich(ch.text)
                ch.metadata.extra.setdefault("entities", tags.entities)
                ch.metadata.extra.setdefault("topics", tags.topics)
            chunks.extend(doc_chunks)
            report["chunks_created"] += len(doc_chunks)
        t0 = datetime.now()
        report["indexing_summary"] = await self.index_manager.index_chunks(chunks=chunks, domain=domain)
        self._record_latency(PipelineStage.INDEXING, _ms_since(t0))
        report["total_time_ms"] = _ms_since(t_all)
        if self.config.enable_audit_logging:
            await self.compliance.log_ingestion(document_count=len(documents), chunk_count=len(chunks), report=report)
        return report

    # ------------------------------------------------------------------ retrieve
    async def retrieve(self, query: str, filters: Optional[Dict[str, Any]] = None,
                       context: Optional[Dict[str, Any]] = None) -> Tuple[List[RetrievalResult], EvaluationMetrics]:
        t_all = datetime.now()
        rewritten = self.query_rewriter.rewrite(query, context or {})
        t0 = datetime.now()
        raw = await self.retriever.retrieve(query=rewritten, filters=filters,
                                            profile_hint=(context or {}).get("retrieval_profile") if context else None)
        self._record_latency(PipelineStage.RETRIEVAL, _ms_since(t0))
        if self.config.enable_reranking:
            t0 = datetime.now()
            ranked = await self.retriever.rerank(query=rewritten, results=raw, top_k=self.config.rerank_top_k)
            self._record_latency(PipelineStage.RERANKING, _ms_since(t0))
        else:
            ranked = raw[:self.config.rerank_top_k]
        t0 = datetime.now()
        metrics = await self.evaluator.evaluate_retrieval(query=query, results=ranked, context=context)
        self._record_latency(PipelineStage.EVALUATION, _ms_since(t0))
        if metrics.hallucination_risk > self.config.max_hallucination_risk:
            print(f"WARNING: High hallucination risk detected: {metrics.hallucination_risk:.3f}")
        total_ms = _ms_since(t_all)
        out: List[RetrievalResult] = []
        for hit in ranked:
            trail = None
            if self.config.enable_audit_logging:
                trail = await self.compliance.log_retrieval(query=query, chunk_id=hit["id"], score=hit["score"],
                                                            latency_ms=total_ms)
            out.append(RetrievalResult(content=hit["content"], chunk_id=hit["id"], score=hit["score"],
                                       metadata=hit["metadata"], retrieval_method=hit.get("method", "hybrid"),
                                       latency_ms=total_ms, audit_trail=trail))
        if total_ms > self.config.target_latency_ms:
            print(f"WARNING: SLA violation - latency {total_ms:.2f}ms exceeds target {self.config.target_latency_ms}ms")
        return out, metrics

    async def plan_and_execute(self, query: str, filters: Optional[Dict[str, Any]] = None,
                               context: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        plan: DecompositionResult = self.decomposer.decompose(query)
        answers = []
        for sub in plan.sub_queries:
            results, metrics = await self.retrieve(query=sub, filters=filters, context=context)
            answers.append({"query": sub, "results": results, "metrics": metrics})
        return {"decomposition": {"sub_queries": plan.sub_queries, "strategy": plan.strategy}, "subqueries": answers}

    async def detect_drift(self, sample_queries: List[str]) -> Dict[str, Any]:
        return await self.evaluator.detect_drift(queries=sample_queries, index_manager=self.index_manager)

    # ------------------------------------------------------------------ telemetry
    def get_performance_report(self) -> Dict[str, Any]:
        report: Dict[str, Any] = {"stage_latencies": {}, "sla_compliance": {}, "throughput_estimate": {}}
        for stage, xs in self.stage_latencies.items():
            if xs:
                report["stage_latencies"][stage.value] = {"p50": np.percentile(xs, 50), "p95": np.percentile(xs, 95),
                                                          "p99": np.percentile(xs, 99), "mean": np.mean(xs),
                                                          "std": np.std(xs)}
        n = len(self.stage_latencies[PipelineStage.RETRIEVAL])
        if n:
            chain = (PipelineStage.RETRIEVAL, PipelineStage.RERANKING, PipelineStage.EVALUATION)
            totals = [sum(self.stage_latencies[s][i] for s in chain if len(self.stage_latencies[s]) > i)
                      for i in range(n)]
            ok = sum(1 for t in totals if t <= self.config.target_latency_ms)
            report["sla_compliance"] = {"target_ms": self.config.target_latency_ms, "compliance_rate": ok / len(totals),
                                        "p95_latency": np.percentile(totals, 95)}
        return report

    def _record_latency(self, stage: PipelineStage, latency_ms: float):
        xs = self.stage_latencies[stage]
        xs.append(latency_ms)
        if len(xs) > 1000:
            self.stage_latencies[stage] = xs[-1000:]

    def _assess_data_quality(self, doc: Dict[str, Any], metrics: DiagnosticMetrics) -> Dict[str, Any]:
        text = (doc.get("text") or "").strip()
        flags = []
        if not text:
            flags.append("empty_text")
        if len(text) > APIConstants.MAX_DOCUMENT_TEXT_LENGTH:
            flags.append("text_too_long")
        if metrics.redundancy_score > 0.95:
            flags.append("high_redundancy")
        if metrics.information_entropy < 0.05:
            flags.append("very_low_entropy")
        return {"document_id": doc.get("id"), "flags": flags}

    async def close(self):
        await self.index_manager.close()
        if self.config.enable_audit_logging:
            await self.compliance.close()
