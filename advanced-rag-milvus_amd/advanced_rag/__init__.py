"""advanced_rag — MI355X-native drop-in for the search hot path of
rnaarla/advanced-rag-milvus.  Same public names as the reference package
(src/advanced_rag/__init__.py:6-39); the Milvus server is replaced by an in-HBM
shard store driven through libhbmrag (HIP kernels for gfx950)."""
from .pipeline import AdvancedRAGPipeline, PipelineConfig, PipelineStage, RetrievalResult
from .diagnostics import DocumentDiagnostics, DiagnosticMetrics
from .chunking import AdaptiveChunker, ChunkMetadata, Chunk
from .indexing import MilvusIndexManager, HbmIndexManager, IndexType, IndexConfig
from .retrieval import HybridRetriever, RetrievalConfig, CrossEncoderReranker, QueryClassifier
from .ranker import LearnedRanker, LearnedRankerConfig
from .semantic_enrichment import SemanticEnricher, EnrichmentResult
from .decomposition import QueryDecomposer, DecompositionResult
from .evaluation import RAGEvaluator, EvaluationMetrics, DriftReport
from .compliance import ComplianceManager, AuditLog, AuditEventType
from .bm25 import BM25SparseEncoder

__version__ = "1.0.0"

__all__ = [
    "AdvancedRAGPipeline", "PipelineConfig", "PipelineStage", "RetrievalResult",
    "DocumentDiagnostics", "DiagnosticMetrics", "AdaptiveChunker", "ChunkMetadata", "Chunk",
    "MilvusIndexManager", "HbmIndexManager", "IndexType", "IndexConfig",
    "HybridRetriever", "RetrievalConfig", "CrossEncoderReranker", "QueryClassifier",
    "LearnedRanker", "LearnedRankerConfig", "SemanticEnricher", "EnrichmentResult",
    "QueryDecomposer", "DecompositionResult", "RAGEvaluator", "EvaluationMetrics", "DriftReport",
    "ComplianceManager", "AuditLog", "AuditEventType", "BM25SparseEncoder",
]
