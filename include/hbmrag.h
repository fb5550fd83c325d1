/*
 * hbmrag.h — C ABI of libhbmrag.so: the MI355X (gfx950) in-HBM shard store and
 * hybrid-search kernels that replace the Milvus server behind
 * advanced-rag-milvus's search hot path.
 *
 * The reference has no FFI of its own: its boundary is the duck-typed
 * index-manager protocol that HybridRetriever consumes
 * (reference src/advanced_rag/retrieval.py:341-419, :634-648) and that
 * MilvusIndexManager implements by RPC to a Milvus server
 * (reference src/advanced_rag/indexing.py:264-437 index_chunks,
 * :439-551 search).  Every entry point below names the reference call it
 * stands in for.  The Python host package (advanced-rag-milvus_amd/advanced_rag)
 * binds these with ctypes; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this line;
 *   - every function returns an hr_status (0 = ok) and never throws or aborts;
 *     hr_last_error() gives the message for the last failure on that handle
 *     (or, with NULL, the last failure of a call that had no handle);
 *   - the caller owns every host buffer it passes in; the library owns all
 *     device memory it allocates; `*_dev` entry points take DEVICE pointers
 *     owned by the caller (e.g. torch tensors) and are asynchronous on the
 *     given hipStream_t (passed as void*);
 *   - results are written into caller-allocated arrays of size B*k, padded
 *     with id -1 / score 0 when fewer than k rows qualify;
 *   - row ids are int64 "global row numbers": local row + the shard's row
 *     offset (hr_set_row_offset), so per-shard lists can be merged across GPUs;
 *   - search calls are thread-safe against each other: a host-form call takes a
 *     private workspace + stream from a pool; `*_dev` calls on the same stream
 *     share one workspace and are serialised while they enqueue (their kernels
 *     then run in stream order); add/finalize take an exclusive lock.
 *
 * Result semantics (what the oracle in oracle/ restates bit-for-bit)
 *   dense  : score32 = (float) S, S computed in fp64 by a k-ordered sequential
 *            sum of exact products x[k]*q[k];  COSINE divides by
 *            sqrt(sum x^2 * sum q^2) (0 when either norm is 0).  Rows are
 *            ranked by (score32 desc, row id asc).  Milvus' HNSW(ef=64) is
 *            approximate; this is the exact FLAT answer of the same metric.
 *   sparse : score32 = (float) sum over the row's stored entries, in stored
 *            (index) order, of value*query_value in fp64; only rows with
 *            score32 > 0 qualify; same ranking rule.
 *   fuse   : reciprocal-rank fusion exactly as reference
 *            retrieval.py:421-491 (float64, k=60 by default, stable order).
 */
#ifndef HBMRAG_H
#define HBMRAG_H

#include <stdint.h>
#include <stddef.h>

#if defined(HR_BUILD)
#define HR_API __attribute__((visibility("default")))
#else
#define HR_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hr_index hr_index; /* opaque shard handle (one per GPU / process) */

typedef enum hr_status {
    HR_OK = 0,
    HR_EINVAL = 1, /* bad argument / shape; Python maps to ValueError            */
    HR_ESTATE = 2, /* call not valid in this state (e.g. search before finalize) */
    HR_EHIP = 3,   /* HIP runtime failure (message holds hipGetErrorString)      */
    HR_ENOMEM = 4, /* host or device allocation failed                           */
    HR_ELIMIT = 5  /* argument exceeds a compiled-in limit (see HR_MAX_*)        */
} hr_status;

enum { HR_F32 = 0, HR_F16 = 1 };               /* storage dtype of the dense shard */
enum { HR_METRIC_IP = 0, HR_METRIC_COSINE = 1 }; /* dense metric_type                */
enum { HR_METHOD_SEMANTIC = 1, HR_METHOD_SPARSE = 2, HR_METHOD_DOMAIN = 4 };

#define HR_MAX_TOPK 256  /* reference clamps top_k to 100 and over-retrieves 2x (constants.py:47, retrieval.py:351) */
#define HR_MAX_DIM 4096  /* dense dimension limit of the LDS-resident query tile */
#define HR_MAX_QUERY_NNZ 4096

/* ---- lifecycle -------------------------------------------------------------
 * Replaces MilvusIndexManager._connect/_initialize_collections/_create_collection
 * (reference indexing.py:134-262): one handle holds the "semantic_index"
 * (dense) and "sparse_index" collections of one shard on one GPU.
 * dim = 0 or sparse_dim = 0 disables that collection. */
HR_API int hr_create(int device, int64_t dim, int dtype, int metric, int64_t sparse_dim, hr_index** out);
HR_API void hr_destroy(hr_index* h);
HR_API const char* hr_last_error(const hr_index* h);
/* Library/ABI version: major*10000 + minor*100 + patch. */
HR_API int hr_version(void);

/* Global id of local row 0 (multi-GPU row sharding; default 0). */
HR_API int hr_set_row_offset(hr_index* h, int64_t first_row);
/* Optional: pre-size the dense store so appends never reallocate. */
HR_API int hr_reserve(hr_index* h, int64_t n_rows);

/* ---- ingest ----------------------------------------------------------------
 * Replaces Collection.insert on semantic_index / sparse_index
 * (reference indexing.py:370-427).  Rows are appended; row numbers are
 * assigned in call order.  hr_add_dense takes row-major fp32 rows and stores
 * them in the shard's dtype (fp16: round-to-nearest-even, as numpy astype).
 * hr_add_dense_raw takes rows already in the shard's dtype (fp16 bits as
 * uint16).  The `_dev` forms take device pointers. */
HR_API int hr_add_dense(hr_index* h, const float* rows, int64_t n);
HR_API int hr_add_dense_raw(hr_index* h, const void* rows, int64_t n);
HR_API int hr_add_dense_raw_dev(hr_index* h, const void* d_rows, int64_t n, void* stream);
/* CSR batch: indptr[n+1], indices sorted ascending within a row, < sparse_dim
 * (the {"indices","values"} payload of reference indexing.py:647-654, batched
 * the way indexing.py:379-404 builds its scipy CSR). */
HR_API int hr_add_sparse(hr_index* h, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n);
/* Replaces Collection.flush/load (reference indexing.py:430-431, :259):
 * computes row norms, builds the doc-range-partitioned postings, sizes the
 * search workspaces.  May be called again after further adds. */
HR_API int hr_finalize(hr_index* h);

/* Shard snapshot (checkpoint / resume; the reference relies on Milvus'
 * persistence, indexing.py:185-188, :430-431).  One file: header (with the file's
 * total size), dense tiles + norms exactly as they sit in HBM, and the sparse CSR
 * (read back from the device); the range-major postings are rebuilt on load.
 * hr_save writes `path`.tmp and renames it over `path` once complete.  hr_load
 * does not trust the file: sizes are checked against the header and the CSR goes
 * through hr_add_sparse's checks; it creates a finalized handle on `device`. */
HR_API int hr_save(hr_index* h, const char* path);
HR_API int hr_load(const char* path, int device, hr_index** out);
/* What a handle (e.g. one returned by hr_load) holds: dimension, storage dtype, metric, sparse dimension.
 * Any out pointer may be NULL. */
HR_API int hr_get_info(const hr_index* h, int64_t* dim, int32_t* dtype, int32_t* metric, int64_t* sparse_dim);

HR_API int64_t hr_num_rows(const hr_index* h);        /* dense rows (Collection.num_entities, indexing.py:687) */
HR_API int64_t hr_num_sparse_rows(const hr_index* h);
HR_API int64_t hr_device_bytes(const hr_index* h);    /* HBM held by the shard */

/* ---- search (host buffers, synchronous) --------------------------------------
 * Replace Collection.search on "semantic_index" / "sparse_index"
 * (reference indexing.py:503-525) for a batch of B queries.
 *   q        [B*dim] fp32 query vectors
 *   rowmask  optional bitmask over local rows, 1 bit per row (bit r%8 of byte
 *            r/8), 1 = row passes the filter expression; NULL = all rows
 *   out_ids  [B*k] int64, out_scores [B*k] fp32, best first. */
HR_API int hr_search_dense(hr_index* h, const float* q, int B, int k, const uint8_t* rowmask,
                    int64_t* out_ids, float* out_scores);
/* Sparse queries as CSR (q_indptr[B+1]); drop_ratio = Milvus
 * drop_ratio_search (reference retrieval.py:97-101): the
 * floor(drop_ratio*nnz) smallest-|value| entries of each query are ignored. */
HR_API int hr_search_sparse(hr_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                     int B, int k, float drop_ratio, const uint8_t* rowmask,
                     int64_t* out_ids, float* out_scores);

/* The same two searches with the row mask already in HBM (hr_filter_eval_dev's output, or a caller's cached mask):
 * host buffers in and out, escalation included, no mask upload. */
HR_API int hr_search_dense_dmask(hr_index* h, const float* q, int B, int k, const uint8_t* d_rowmask,
                          int64_t* out_ids, float* out_scores);
HR_API int hr_search_sparse_dmask(hr_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val,
                           int B, int k, float drop_ratio, const uint8_t* d_rowmask,
                           int64_t* out_ids, float* out_scores);

/* Reciprocal-rank fusion of up to three ranked id lists of ONE query
 * (reference HybridRetriever._fuse_results, retrieval.py:421-491), executed
 * by the device kernel.  ids < 0 end a list.  out arrays hold na+nb+nc
 * entries; *n_out receives the number of fused ids. */
HR_API int hr_fuse_rrf(hr_index* h, const int64_t* ids_a, int na, const int64_t* ids_b, int nb,
                const int64_t* ids_c, int nc, double wa, double wb, double wc, int rrf_k,
                int64_t* out_ids, double* out_scores, int32_t* out_methods, int32_t* n_out);

/* ---- search (device buffers, asynchronous on `stream`) -------------------------
 * Same results as the host forms.  d_flags[B] (optional) receives 1 when the
 * candidate-generation bound proves the list exact, 0 when the caller must
 * re-run that query through the host form (which escalates by itself). */
/* The sparse form takes queries already reduced (drop_ratio applied) and
 * sorted by index, as CSR; max_q_nnz = the longest query (sets the scan's
 * rounding bound). */
HR_API int hr_search_dense_dev(hr_index* h, const float* d_q, int B, int k, const uint8_t* d_rowmask,
                        int64_t* d_ids, float* d_scores, int32_t* d_flags, void* stream);
HR_API int hr_search_sparse_dev(hr_index* h, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                         const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                         const uint8_t* d_rowmask, int64_t* d_ids, float* d_scores,
                         int32_t* d_flags, void* stream);
/* Both modalities of one query batch in ONE call: the dense scan runs alone on
 * `stream`; the sparse chain then runs on a library-owned side stream so that
 * it overlaps the dense path's latency-bound tail (candidate select, refine,
 * top-k); `stream` waits for both before the call's work is considered done.
 * Results are identical to calling the two *_dev forms one after the other.
 * d_ids/d_scores/d_flags: dense lists first, then sparse ([2][B][k], [2][B]). */
HR_API int hr_search_hybrid_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                         const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                         const uint8_t* d_rowmask, int64_t* d_ids, float* d_scores, int32_t* d_flags,
                         void* stream);
/* Two-phase form of the same search, for keeping several query batches in
 * flight: phase 1 (`scan`) = query prep + the two bandwidth-bound shard scans,
 * leaving the per-group maxima in workspace `slot` (0 <= slot < HR_MAX_SLOTS);
 * phase 2 (`finish`) = candidate select + canonical refine + top-k for both
 * modalities from that slot.  A caller alternates slots and puts the scans of
 * batch i+1 on one stream while the finish of batch i (plus exchange, merge,
 * fuse, rerank) runs on another — scans never overlap each other, the
 * latency-bound tail hides behind them.  The caller orders the phases of one
 * slot with events; `finish` takes the same query buffers as `scan`. */
#define HR_MAX_SLOTS 4
/* Optional phase 0: the query preparation of `slot` alone (unit-normalised fragment-order queries, canonical |q|^2,
 * fixed-stride sparse queries and their fixed-point scale) on a stream of the caller's choice, so that the stream
 * that carries the scans carries nothing else.  The next hr_hybrid_scan_dev on the same slot then enqueues the two
 * scans only; the caller orders prep -> scan with an event.  Without this call `scan` prepares the queries itself. */
HR_API int hr_hybrid_prep_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                       const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k, int slot, void* stream);
HR_API int hr_hybrid_scan_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                       const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                       const uint8_t* d_rowmask, int slot, void* stream);
HR_API int hr_hybrid_finish_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                         const float* d_q_val, int B, int max_q_nnz, int k, const uint8_t* d_rowmask, int slot,
                         int64_t* d_ids, float* d_scores, int32_t* d_flags, void* stream);
/* Batched RRF: lists are [B][ka], [B][kb], [B][kc] (kc = 0 / NULL for none);
 * outputs [B][top_k] ids / fp64 scores / method bitmasks and d_n_out[B]. */
HR_API int hr_fuse_rrf_dev(const int64_t* d_ids_a, int ka, const int64_t* d_ids_b, int kb,
                    const int64_t* d_ids_c, int kc, int B, double wa, double wb, double wc,
                    int rrf_k, int top_k, int64_t* d_out_ids, double* d_out_scores,
                    int32_t* d_out_methods, int32_t* d_n_out, void* stream);
/* Cross-shard merge after the RCCL all-gather: n_lists per-shard lists of
 * k_in (score, id) pairs per query, each sorted by score descending as the
 * search entry points write them (ids < 0 pad the tail); list l starts
 * l*score_stride floats / l*id_stride int64s after the base pointers and is
 * laid out [B][k_in] (so the lists can sit inside the ranks' slots of one
 * all-gather buffer); output the best k_out by (score desc, id asc).  (No
 * reference analogue: Milvus merges its num_shards=4 segments server-side,
 * indexing.py:234-239.) */
HR_API int hr_merge_topk_dev(const float* d_scores, const int64_t* d_ids, int n_lists, int64_t score_stride,
                      int64_t id_stride, int B, int k_in, int k_out, int64_t* d_out_ids,
                      float* d_out_scores, void* stream);
/* LearnedRanker.score + HybridRetriever.rerank's stable sort and cut
 * (reference ranker.py:109-125, retrieval.py:544-563) for a batch:
 *   new = base_w*score + method_bonus*popcount(methods) + recency_w*recency
 * Inputs are the outputs of hr_fuse_rrf_dev ([B][k_in], d_n[B] valid entries);
 * d_recency may be NULL (= 0).  Outputs [B][k_out]: ids, new scores, and the
 * fused score each kept entry had before (original_retrieval_score). */
HR_API int hr_rerank_linear_dev(const int64_t* d_ids, const double* d_scores, const int32_t* d_methods,
                         const int32_t* d_n, const double* d_recency, int B, int k_in, double base_w,
                         double method_bonus, double recency_w, int k_out, int64_t* d_out_ids,
                         double* d_out_scores, double* d_out_orig, void* stream);

/* Everything after the per-shard lists of a query batch in ONE launch (one block per query): [hr_merge_topk_dev of
 * every modality's exchanged lists] -> hr_fuse_rrf_dev -> [hr_rerank_linear_dev]; results are bit-identical to the
 * separate calls (reference retrieval.py:421-491 fusion, :518-563 rerank, after the server-side shard merge of
 * indexing.py:234-239).  Modality m = 0 semantic, 1 sparse, 2 domain; k_in[m] = 0 leaves a modality out.
 *   n_lists = 1: ids[m] are [B][k_in[m]] lists fused as they are (k_fuse[m] must equal k_in[m]; scores/merged_* unused);
 *   n_lists > 1: ids[m] / scores[m] point at list 0 of modality m inside the gathered buffer, list l lies l*id_stride
 *                int64s / l*score_stride floats further; the merged [B][k_fuse[m]] lists are written to merged_*[m]
 *                and fused.
 * fused_* are [B][top_k] (+ fused_n[B]); with rerank != 0 the learned-ranker outputs rr_* are [B][k_out].
 * B blocks are launched: agg_flags (if asked for) is filled by strided loops over all n_flag_rows. */
typedef struct hr_post_args {
    const int64_t* ids[3];
    const float* scores[3];
    int32_t k_in[3];
    int32_t k_fuse[3];
    int32_t n_lists;
    int32_t rrf_k;
    int64_t id_stride, score_stride;
    int64_t* merged_ids[3];
    float* merged_scores[3];
    double w[3];
    int64_t* fused_ids;
    double* fused_scores;
    int32_t* fused_methods;
    int32_t* fused_n;
    int32_t top_k;
    int32_t rerank;
    double base_w, method_bonus, recency_w;
    const double* recency;
    int32_t k_out;
    int32_t reserved;
    int64_t* rr_ids;
    double* rr_scores;
    double* rr_orig;
    /* optional (n_lists > 1): the per-list "proven exact" flags that travelled with the lists — list l's flags are
     * n_flag_rows int32 ([modality][B]) starting l*flag_stride int32s after `flags`; agg_flags[n_flag_rows] receives
     * their minimum over the lists (a list is proven iff every shard proved its part) */
    const int32_t* flags;
    int64_t flag_stride;
    int32_t n_flag_rows;
    int32_t reserved2;
    int32_t* agg_flags;
    /* optional: fusion weights per query, [B][3] doubles (semantic, sparse, domain) — the reference lets a
     * weight_adapter choose them per request (retrieval.py:251-262); NULL = `w` for every query of the batch */
    const double* w_query;
} hr_post_args;
HR_API int hr_post_lists_dev(const hr_post_args* args, int B, void* stream);

/* ---- filter expressions -> row mask, on the device ------------------------------------------------------------
 * Replaces the server-side evaluation of the `expr` argument of Collection.search (reference indexing.py:503-525; the
 * expressions come from HybridRetriever._build_filter_expression, retrieval.py:565-632: a conjunction of
 * `field OP literal` terms over the scalar fields of the schema, indexing.py:191-225).  Columns are device arrays of
 * n_rows entries: int64, float32, or — for string fields — two uint64 per row holding the big-endian words of the
 * first 16 UTF-8 bytes, zero padded (an order-preserving prefix key).  d_mask receives the packed predicate (bit r%8 of
 * byte r/8, what every search entry point takes as rowmask); rows a string term cannot decide from the key alone
 * (first 16 bytes equal to the literal's) are left 0 in d_mask and set in d_undecided for the caller to resolve on the
 * full strings.  Both buffers hold 8 * ceil(n_rows / 64) bytes.  d_counts[2] (zeroed by the call) receives the
 * number of rows kept / undecided.  d_deleted (optional) = tombstones, 1 bit per row, 1 = never passes. */
enum { HR_COL_I64 = 0, HR_COL_I64_VS_F64 = 1, HR_COL_F32 = 2, HR_COL_STR16 = 3 };
enum { HR_OP_EQ = 0, HR_OP_NE = 1, HR_OP_LT = 2, HR_OP_LE = 3, HR_OP_GT = 4, HR_OP_GE = 5 };
#define HR_MAX_FILTER_TERMS 16
typedef struct hr_filter_term {
    int32_t kind;        /* HR_COL_* */
    int32_t op;          /* HR_OP_*  */
    const void* col;     /* device column */
    int64_t ival;        /* literal for HR_COL_I64 */
    double dval;         /* literal for HR_COL_I64_VS_F64 */
    float fval;          /* literal for HR_COL_F32 (already rounded to float32) */
    uint32_t reserved;
    uint64_t key[2];     /* literal's prefix key for HR_COL_STR16 */
} hr_filter_term;
HR_API int hr_filter_eval_dev(const hr_filter_term* terms, int n_terms, int64_t n_rows, const uint8_t* d_deleted,
                       uint8_t* d_mask, uint8_t* d_undecided, int32_t* d_counts, void* stream);

/* ---- BM25 document payloads (ingest) ----------------------------------------------------------------------------
 * The `encode_sparse` hook of the ingest path (reference indexing.py:629-654, called per chunk from index_chunks
 * :379-404) for the BM25 encoder of this build (advanced_rag/bm25.py), a batch at a time on the device: d_text = the
 * documents' UTF-8 bytes back to back, d_off[n_docs + 1] their byte offsets.  Per document: lower-cased `\b\w+\b` tokens
 * (ASCII classes), slot = crc32(token) mod sparse_dim, weight = (float)(tf (k1 + 1) / (tf + k1 (1 - b + b dl / avgdl))) in
 * double — bit for bit what BM25SparseEncoder.encode_document returns.  Out: d_idx / d_val [n_docs][cap] (slots ascending),
 * d_nnz[n_docs], d_flags[n_docs]: 0 = encoded, 1 = the document holds a byte >= 0x80 (Unicode case mapping and
 * categories are the host's), 2 = longer than 65 535 bytes or more than `cap` distinct slots — flagged documents have
 * nnz = 0 and are encoded by the caller on the host.  sparse_dim <= 65 536.  Asynchronous on `stream`. */
HR_API int hr_bm25_encode_dev(const uint8_t* d_text, const int64_t* d_off, int n_docs, int sparse_dim, double k1, double b,
                       double avgdl, int cap, int32_t* d_idx, float* d_val, int32_t* d_nnz, int32_t* d_flags, void* stream);

/* The hash tokenizer in front of the sentence encoder (advanced_rag/encoders.py::HashTokenizer; reference hook
 * embedding_generator.encode_semantic, indexing.py:610-620 / :580-587) for a batch of single texts: tokens of the
 * lower-cased text = `\w+|[^\w\s]`, id = 1000 + crc32(token) mod (vocab - 1000), row = CLS(101) ids[: max_len - 2] SEP(102),
 * zero padded to max_len.  d_ids [n][max_len] int64, d_lens[n] (CLS and SEP included), d_flags[n]: 1 = the text holds a
 * byte >= 0x80 (tokenise on the host).  Asynchronous on `stream`. */
HR_API int hr_hash_tokenize_dev(const uint8_t* d_text, const int64_t* d_off, int n, int max_len, int vocab, int64_t* d_ids,
                         int32_t* d_lens, int32_t* d_flags, void* stream);

/* ---- encoder / cross-encoder forward: fused elementwise pieces -----------------
 * The GEMMs and the attention of the PyTorch-ROCm encoder forwards stay with
 * hipBLASLt / SDPA; this is the residual add + LayerNorm every post-LN BERT
 * layer runs twice (the model class the reference's CrossEncoderReranker names,
 * retrieval.py:651-662):  out = LayerNorm(x (+ residual)) * gamma + beta, fp16
 * rows of `hidden` values (multiple of 8, <= 1024), fp32 statistics, one pass.
 * All pointers are device pointers (16-byte aligned); residual may be NULL;
 * out may alias x or residual. */
HR_API int hr_add_layernorm_f16_dev(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta,
                             void* d_out, int64_t rows, int hidden, float eps, void* stream);

/* The embedding layer of the same models in one pass: d_out[s][t] = LayerNorm(word[ids[s][t]] + pos[t] + seg[types[s][t]])
 * (fp16 tables and output, hidden as above; ids / types int64, clamped into [0, n_word) / [0, n_seg); d_pos holds at least
 * T rows; the two adds are rounded to fp16 one after the other, as the unfused fp16 module rounds them). */
HR_API int hr_embed_layernorm_f16_dev(const int64_t* d_ids, const int64_t* d_types, const void* d_word, const void* d_pos,
                               const void* d_seg, const void* d_gamma, const void* d_beta, void* d_out, int64_t n_seq, int T,
                               int hidden, float eps, int64_t n_word, int64_t n_seg, void* stream);

/* Self-attention for head dimensions 32 and 64, from the fused QKV projection's output to the layout the output
 * projection reads (the PyTorch SDPA call plus the permute / transpose copies around it, in one kernel):
 *   d_qkv [n_seq][T][3][heads][head_dim] fp16, d_lengths[n_seq] valid tokens per sequence (padding at the tail; NULL = T),
 *   d_out [n_seq][T][heads * head_dim] fp16 = softmax(scale * Q K^T, keys < length) V per head, fp32 accumulation.
 * T <= 1024 at head_dim 32, <= 512 at 64 (K and V of a (sequence, head) are staged in LDS); HR_ELIMIT beyond. */
HR_API int hr_attention_f16_dev(const void* d_qkv, const int32_t* d_lengths, void* d_out, int64_t n_seq, int T, int heads,
                         int head_dim, float scale, void* stream);
/* The same kernel with its operands by pointer and stride (in halves, multiples of 8): query rows at
 * d_q + seq * q_seq_stride + token * q_token_stride + head * 32, key / value rows at d_k / d_v + seq * kv_seq_stride +
 * token * kv_token_stride + head * 32; the first n_queries (<= T) tokens of a sequence are its queries;
 * d_out [n_seq][n_queries][heads * 32].  The cross-encoder's last layer, whose head reads token 0 only, calls it with
 * n_queries = 1 over a [n_seq][T][2][heads][32] key / value buffer (advanced_rag/encoders.py). */
HR_API int hr_attention_rows_f16_dev(const void* d_q, int64_t q_seq_stride, int64_t q_token_stride, const void* d_k, const void* d_v,
                              int64_t kv_seq_stride, int64_t kv_token_stride, const int32_t* d_lengths, void* d_out,
                              int64_t n_seq, int T, int n_queries, int heads, int head_dim, float scale, void* stream);

/* ---- encoder / cross-encoder layer: the linear maps as hand-written MFMA kernels (round 4) ---------------------------
 * The forward passes behind the reference's plugin hooks (`CrossEncoderReranker.model.predict`, retrieval.py:651-685;
 * `embedding_generator.encode_semantic`, indexing.py:610-620) are post-LN BERT layers.  These two entry points are the
 * GEMM-shaped part of a layer (csrc/encoder_layer.h): weights travel as PACKED streams of 1 KiB pieces — one
 * v_mfma_f32_16x16x32_f16 A fragment (16 output features x 32 inputs) in register-image order: lane l = 16 g + r of the
 * piece holds 8 consecutive halves of row 16 t + r —
 *   natural k order:      inputs 32 s + 8 g + j,                                    j = 0..7
 *   accumulator k order:  inputs 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4)        (the operand comes out of an MFMA)
 * built once per model by advanced_rag/encoder_kernels.py.  fp16 activations, fp32 accumulation, fp32 biases / LayerNorm
 * parameters.  All pointers are device pointers, 16-byte aligned.
 *
 * FRAGMENT ORDER ("FR") is the activation layout between the launches of a layer: X_fr[tile = row / 16][s][lane = 16 g + c][j]
 * = X[16 tile + c][32 s + 16 (j >> 2) + 4 g + (j & 3)], 1 KiB per (16-row tile, k-step); buffers hold ceil(rows / 16) tiles.
 * It is the accumulator layout of two neighbouring output tiles AND the B operand of the next product in accumulator k
 * order, so rows move with 16-byte accesses contiguous over the wave.  Row-major stays available per operand.
 *
 * hr_linear_rows_f16_dev:  d_out[r][0..N) (row-major, row stride out_stride halves) = x[r][0..K) W^T + bias.  d_x: row-major
 *   [rows][K] (x_fr = 0: d_w_packed pieces [N/16][K/32] in natural k order) or FR (x_fr = 1: accumulator k order).
 *   The rows of W (and d_bias) are packed in STORE ORDER: packed row 32 so + 16 u + r = output feature
 *   32 so + 8 (r >> 2) + 4 u + (r & 3), which makes a lane's eight results of a stage eight consecutive features of its row.
 *   N a multiple of 32 (<= 4096); K = 384 (HR_ELIMIT otherwise: callers keep their GEMM).
 * hr_attention_fr_f16_dev: hr_attention_f16_dev with its output in FR ([rows = n_seq * T][heads * head_dim]).
 * hr_encoder_tail_f16_dev: out = LN2(x1 + W_down gelu(W_up x1 + b_up) + b_down), x1 = LN1(x + W_out attn + b_out)
 *   for rows of `hidden` halves: everything of a layer after the attention in ONE launch; the [rows][intermediate] FFN
 *   intermediate never leaves the compute unit.  d_attn_fr: FR.  d_x / d_out: FR or row-major per x_fr / out_fr.
 *   d_wstream: stages of hidden/16 pieces, ALL in accumulator k order —
 *     W_out, two output tiles per stage: hidden/32 stages;  then, with up(c) = output tiles 2c, 2c+1 of W_up [all k-steps]
 *     and down(c) = k-step c of every output tile of W_down:
 *     up(0) | up(1) down(0) | up(2) down(1) | ... | up(I/32-1) down(I/32-2) | down(I/32-1).
 *   d_tables (fp32): b_out[H] ln1_gamma[H] ln1_beta[H] b_down[H] ln2_gamma[H] ln2_beta[H] b_up[I].
 *   gelu_erf: 0 = tanh form, 1 = exact erf form.  (hidden, intermediate) = (384, 1536) (HR_ELIMIT otherwise). */
HR_API int hr_linear_rows_f16_dev(const void* d_x, int x_fr, const void* d_w_packed, const float* d_bias, void* d_out, int64_t rows,
                           int K, int N, int64_t out_stride, void* stream);
HR_API int hr_attention_fr_f16_dev(const void* d_qkv, const int32_t* d_lengths, void* d_out_fr, int64_t n_seq, int T, int heads,
                            int head_dim, float scale, void* stream);
HR_API int hr_encoder_tail_f16_dev(const void* d_attn_fr, const void* d_x, int x_fr, void* d_out, int out_fr, const void* d_wstream,
                            const float* d_tables, int64_t rows, int hidden, int intermediate, float eps, int gelu_erf,
                            void* stream);

/* ---- streams with a compute-unit mask -------------------------------------------
 * A HIP stream whose kernels may only occupy the compute units named by `cu_mask` (bit i of word i/32 = CU i;
 * n_words 32-bit words; on gfx950 consecutive bits alternate over the 8 XCDs) — the way to keep a latency-bound
 * finishing chain and the bandwidth-bound scans off each other's compute units.  priority: 0 = default, negative =
 * higher.  The stream is an ordinary hipStream_t for every other purpose (torch.cuda.ExternalStream wraps it).
 * hr_set_scan_cus tells the scans how many compute units their stream may use (their persistent grids are sized to
 * it; 0 = all of the device's). */
HR_API int hr_stream_create(int device, int priority, const uint32_t* cu_mask, int n_words, void** out_stream);
HR_API int hr_stream_destroy(int device, void* stream);
HR_API int hr_set_scan_cus(hr_index* h, int n_cus);

/* ---- test / diagnosis hooks ---------------------------------------------------------
 * HR_DEBUG_FINISH_MODE (process-wide, handle may be NULL): 0 = choose the finishing path by batch size (default),
 *   1 = always the multi-launch chain, 2 = the fused finishing kernel whenever the candidate set fits LDS — lets the
 *   parity tests drive both paths at any batch size.
 * HR_DEBUG_FAIL_NEXT_BUILD: value != 0 makes the next hr_finalize fail after the sparse rows have reached the device
 *   CSR and before the postings are rebuilt (what an allocation failure at that point leaves behind); the
 *   finalize after that must complete the build.
 * HR_DEBUG_DENSE_KERNELS (process-wide): bit mask that takes dense scan kernels out of the selection so that the
 *   others serve the shapes they would have served — 1 = no register-resident 256-query pass, 2 = no tiled-contraction
 *   pass, 4 = no k-chunked large-batch pass, 8 = prefer the tiled contraction where both 256-query passes apply, 16 = the 4 x 64-query register form
 *   (dense_scan_q64_kernel) instead of the 8 x 32-query one for the 256-query pass at D = 768.
 * HR_DEBUG_SPARSE_RPB (process-wide): doc ranges one sparse-scan block walks (0 = by shard size).
 * HR_DEBUG_GROUP_ROWS (process-wide): rows per candidate group (16 or 64; 0 = by shard size) of handles created
 *   afterwards.
 * HR_DEBUG_NO_TRIM (process-wide): 1 = refine every one of the C candidate groups of a query (round 3's behaviour) instead
 *   of only those whose maximum is within twice the scan's error bound of the k-th largest group maximum (A/B, tests).
 * The library reads no environment variables. */
enum { HR_DEBUG_FINISH_MODE = 1, HR_DEBUG_FAIL_NEXT_BUILD = 2, HR_DEBUG_DENSE_KERNELS = 3, HR_DEBUG_SPARSE_RPB = 4,
       HR_DEBUG_GROUP_ROWS = 5, HR_DEBUG_NO_TRIM = 6 };
HR_API int hr_debug_option(hr_index* h, int key, int value);

/* ---- measurement hooks -------------------------------------------------------
 * hr_set_profiling(1) brackets every dense-scan and sparse-scan launch with
 * HIP events on the stream it is launched on; (2) brackets all ten phases:
 * [0] query prep, [1] dense scan, [2] group select, [3] refine, [4] top-k,
 * [5] sparse scan, [6] sparse select, [7] sparse refine, [8] sparse top-k,
 * [9] the fused finishing kernel (select + refine + top-k of a batch in one launch).
 * hr_last_kernel_ms drains the recorded spans: out_ms[p] = mean ms per launch
 * of phase p since the previous call, out_ms[HR_N_PHASES + p] = launches
 * averaged.  n must be >= 2*HR_N_PHASES. */
#define HR_N_PHASES 10
HR_API int hr_set_profiling(hr_index* h, int enabled);
HR_API int hr_last_kernel_ms(hr_index* h, float* out_ms, int n);
/* Algorithmic bytes one dense scan launch reads (rows*dim*sizeof(elem) + 4*rows). */
HR_API int64_t hr_dense_scan_bytes(const hr_index* h);

#ifdef __cplusplus
}
#endif
#endif /* HBMRAG_H */
