"""Numeric settings of the search hot path.

Values (not code) follow reference src/advanced_rag/constants.py: the
retrieval block :44-70, Milvus defaults :170-190, indexing :223-234,
performance/cache :94-111.  Only what the hot path reads is kept.
"""


class RetrievalConstants:
    DEFAULT_TOP_K = 20
    MAX_TOP_K = 100            # profiles clamp to this; searches over-retrieve 2x
    DEFAULT_RERANK_TOP_K = 5
    DEFAULT_HYBRID_ALPHA = 0.7
    TIMEOUT_SECONDS = 0.3      # end-to-end budget of HybridRetriever.retrieve
    RRF_K_PARAMETER = 60
    DEFAULT_DENSE_WEIGHT = 0.7
    DEFAULT_SPARSE_WEIGHT = 0.3
    DEFAULT_DOMAIN_WEIGHT = 0.2
    SEMANTIC_WEIGHT = DEFAULT_DENSE_WEIGHT
    SPARSE_WEIGHT = DEFAULT_SPARSE_WEIGHT
    DOMAIN_WEIGHT = DEFAULT_DOMAIN_WEIGHT
    DEFAULT_MMR_LAMBDA = 0.7


class PerformanceConstants:
    TARGET_LATENCY_MS = 80.0
    DEFAULT_MAX_CONCURRENCY = 64
    DEFAULT_RETRIEVE_TIMEOUT_MS = 300
    DEFAULT_CACHE_SIZE = 10000
    DEFAULT_CACHE_TTL_SECONDS = 3600


class MilvusConstants:
    """Kept under the reference's name: these describe the collections the
    HBM shard store stands in for."""
    DEFAULT_HNSW_M = 16
    DEFAULT_HNSW_EF_CONSTRUCTION = 200
    DEFAULT_HNSW_EF = 64
    DEFAULT_SEMANTIC_DIM = 1536
    DEFAULT_SPARSE_DIM = 10000
    DEFAULT_DOMAIN_DIM = 768
    DEFAULT_NUM_SHARDS = 4
    MAX_VARCHAR_LENGTH = 65535
    MAX_METADATA_JSON_LENGTH = 10000
    DEFAULT_SEARCH_TIMEOUT_SECONDS = 5.0
    DEFAULT_INSERT_TIMEOUT_SECONDS = 30.0
    SPARSE_DROP_RATIO_SEARCH = 0.2


class EmbeddingConstants:
    SEMANTIC_DIM = MilvusConstants.DEFAULT_SEMANTIC_DIM
    SPARSE_DIM = MilvusConstants.DEFAULT_SPARSE_DIM
    DOMAIN_DIM = MilvusConstants.DEFAULT_DOMAIN_DIM
    CACHE_MAX_SIZE = PerformanceConstants.DEFAULT_CACHE_SIZE
    CACHE_TTL_SECONDS = PerformanceConstants.DEFAULT_CACHE_TTL_SECONDS


class IndexingConstants:
    BATCH_SIZE = 64
    RETRY_ATTEMPTS = 3
    RETRY_WAIT_MIN = 0.5
    RETRY_WAIT_MAX = 5.0
    MILVUS_TIMEOUT_SECONDS = 5.0
    THREAD_POOL_WORKERS = 8


class APIConstants:
    MAX_DOCUMENT_TEXT_LENGTH = 1_000_000
    MAX_QUERY_LENGTH = 10_000
