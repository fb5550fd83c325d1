/*
 * oracle.c — CPU restatement of the hybrid-search hot path.  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product (advanced-rag-milvus_amd/) never does.
 *
 * What it restates
 *   dense  : the metric semantics of Collection.search on "semantic_index"
 *            (reference src/advanced_rag/indexing.py:503-525 with
 *            metric_type COSINE / IP, retrieval.py:93-96) as an exact FLAT scan.
 *            The arithmetic itself lives in the Milvus server (pymilvus>=2.3.0,
 *            milvusdb/milvus:v2.3.3 — not in the reference tree, no reference
 *            test pins it): PARITY UNPINNED at that boundary; the score
 *            definition below is this build's, shared bit-for-bit with the HIP
 *            refine kernel (csrc/dense.h refine_dense_kernel).
 *   sparse : SPARSE_INVERTED_INDEX / IP on "sparse_index"
 *            (indexing.py:156-167, :487-498) — same remark.
 *   drop   : drop_ratio_search (retrieval.py:97-101).
 *   rrf    : HybridRetriever._fuse_results (retrieval.py:421-491) — pinned by
 *            golden vectors generated from the imported reference
 *            (tests/golden/gen_golden.py).
 *
 * Build: oracle/build.py (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_F32 0
#define ORACLE_F16 1
#define ORACLE_IP 0
#define ORACLE_COSINE 1

static double half_to_double(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) & 1u;
    const uint32_t exp = (uint32_t)(h >> 10) & 0x1Fu;
    const uint32_t man = (uint32_t)h & 0x3FFu;
    double v;
    if (exp == 0) v = ldexp((double)man, -24);            /* subnormal / zero */
    else if (exp == 31) v = man ? NAN : INFINITY;
    else v = ldexp((double)(man | 0x400u), (int)exp - 25);
    return sign ? -v : v;
}

static inline double elem(const void* X, int dtype, int64_t i) {
    return dtype == ORACLE_F16 ? half_to_double(((const uint16_t*)X)[i]) : (double)((const float*)X)[i];
}

/* Canonical dense score of every row against one query.
 * S = sum_k x[k]*q[k] in fp64, k ascending, one add per product (the products
 * of an fp16/fp32 by an fp32 are exact in fp64, so fused or not is the same).
 * COSINE: S / sqrt(sum x^2 * sum q^2), 0 if either norm is 0.  Returned as fp32. */
void oracle_dense_scores(const void* X, int dtype, int64_t n, int dim, const float* q, int metric, float* out) {
    double qn2 = 0.0;
    for (int k = 0; k < dim; ++k) qn2 += (double)q[k] * (double)q[k];
    for (int64_t r = 0; r < n; ++r) {
        double s = 0.0, xn2 = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double x = elem(X, dtype, r * dim + k);
            s += x * (double)q[k];
            xn2 += x * x;
        }
        if (metric == ORACLE_COSINE) {
            const double d = xn2 * qn2;
            s = d > 0.0 ? s / sqrt(d) : 0.0;
        }
        out[r] = (float)s;
    }
}

/* Canonical sparse score: walk the row's stored entries in order, look the
 * index up in the query (sorted ascending), add exact products in fp64. */
void oracle_sparse_scores(const int64_t* indptr, const int32_t* idx, const float* val, int64_t n,
                          const int32_t* q_idx, const float* q_val, int q_nnz, float* out) {
    for (int64_t r = 0; r < n; ++r) {
        double s = 0.0;
        for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            const int32_t t = idx[e];
            int lo = 0, hi = q_nnz;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (q_idx[mid] < t) lo = mid + 1; else hi = mid;
            }
            if (lo < q_nnz && q_idx[lo] == t) s += (double)val[e] * (double)q_val[lo];
        }
        out[r] = (float)s;
    }
}

/* Ranking rule shared by every list the engine returns: score descending,
 * then row id ascending.  mask (optional): bit r%8 of byte r/8 set = row
 * allowed.  only_positive: rows need score > 0 (sparse).  Pads with -1 / 0. */
typedef struct { float s; int64_t r; } oracle_pair;
static int pair_cmp(const void* a, const void* b) {
    const oracle_pair* x = (const oracle_pair*)a;
    const oracle_pair* y = (const oracle_pair*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->r > y->r) - (x->r < y->r);
}
int oracle_topk(const float* scores, int64_t n, const uint8_t* mask, int only_positive, int k, int64_t row_offset,
                int64_t* out_ids, float* out_scores) {
    oracle_pair* p = (oracle_pair*)malloc(sizeof(oracle_pair) * (size_t)(n > 0 ? n : 1));
    if (!p) return -1;
    int64_t m = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (mask && !((mask[r >> 3] >> (r & 7)) & 1)) continue;
        if (only_positive && !(scores[r] > 0.0f)) continue;
        p[m].s = scores[r];
        p[m].r = r;
        ++m;
    }
    qsort(p, (size_t)m, sizeof(oracle_pair), pair_cmp);
    for (int i = 0; i < k; ++i) {
        if (i < m) {
            out_ids[i] = p[i].r + row_offset;
            out_scores[i] = p[i].s;
        } else {
            out_ids[i] = -1;
            out_scores[i] = 0.0f;
        }
    }
    free(p);
    return (int)(m < k ? m : k);
}

/* drop_ratio_search: ignore the floor(ratio*nnz) entries of smallest |value|
 * (among equal magnitudes the later entry goes first), keep the rest ordered by
 * index.  Returns the kept count; out arrays need room for nnz. */
typedef struct { float a; int i; } oracle_mag;
static int mag_cmp(const void* a, const void* b) {
    const oracle_mag* x = (const oracle_mag*)a;
    const oracle_mag* y = (const oracle_mag*)b;
    if (x->a < y->a) return -1;
    if (x->a > y->a) return 1;
    return (y->i > x->i) - (y->i < x->i);
}
typedef struct { int32_t t; float v; } oracle_term;
static int term_cmp(const void* a, const void* b) {
    const int32_t x = ((const oracle_term*)a)->t, y = ((const oracle_term*)b)->t;
    return (x > y) - (x < y);
}
int oracle_drop_query(const int32_t* idx, const float* val, int nnz, double ratio, int32_t* out_idx, float* out_val) {
    if (nnz <= 0) return 0;
    oracle_mag* m = (oracle_mag*)malloc(sizeof(oracle_mag) * (size_t)nnz);
    oracle_term* t = (oracle_term*)malloc(sizeof(oracle_term) * (size_t)nnz);
    if (!m || !t) { free(m); free(t); return -1; }
    for (int i = 0; i < nnz; ++i) { m[i].a = fabsf(val[i]); m[i].i = i; }
    qsort(m, (size_t)nnz, sizeof(oracle_mag), mag_cmp);
    const int n_drop = (int)floor(ratio * (double)nnz);
    int kept = 0;
    for (int j = n_drop; j < nnz; ++j) { t[kept].t = idx[m[j].i]; t[kept].v = val[m[j].i]; ++kept; }
    qsort(t, (size_t)kept, sizeof(oracle_term), term_cmp);
    for (int j = 0; j < kept; ++j) { out_idx[j] = t[j].t; out_val[j] = t[j].v; }
    free(m);
    free(t);
    return kept;
}

/* Reciprocal-rank fusion, following reference retrieval.py:432-487 line by line:
 *   k = 60; for rank, result in enumerate(list, start=1):
 *       fused[id].score += (1.0 / (k + rank)) * weight
 * lists visited semantic, sparse, domain; dict insertion order kept; then a
 * stable sort by score descending.  ids < 0 terminate a list.
 * Outputs up to na+nb+nc entries; returns the number of distinct ids. */
int oracle_rrf(const int64_t* a, int na, const int64_t* b, int nb, const int64_t* c, int nc, double wa, double wb,
               double wc, int rrf_k, int64_t* out_ids, double* out_scores, int32_t* out_methods) {
    const int cap = na + nb + nc;
    if (cap <= 0) return 0;
    int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
    double* sc = (double*)malloc(sizeof(double) * (size_t)cap);
    int32_t* me = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int* order = (int*)malloc(sizeof(int) * (size_t)cap);
    if (!ids || !sc || !me || !order) { free(ids); free(sc); free(me); free(order); return -1; }
    int n = 0;
    const int64_t* lists[3] = {a, b, c};
    const int lens[3] = {na, nb, nc};
    const double w[3] = {wa, wb, wc};
    for (int l = 0; l < 3; ++l) {
        for (int i = 0; i < lens[l]; ++i) {
            const int64_t id = lists[l][i];
            if (id < 0) break;
            const double rrf = 1.0 / (double)(rrf_k + i + 1);
            const double add = rrf * w[l];
            int slot = -1;
            for (int j = 0; j < n; ++j) if (ids[j] == id) { slot = j; break; }
            if (slot < 0) { slot = n++; ids[slot] = id; sc[slot] = 0.0; me[slot] = 0; }
            sc[slot] = sc[slot] + add;
            me[slot] |= (1 << l);
        }
    }
    /* stable insertion sort, descending */
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i) {
        const int cur = order[i];
        int j = i - 1;
        while (j >= 0 && sc[order[j]] < sc[cur]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = cur;
    }
    for (int i = 0; i < n; ++i) {
        out_ids[i] = ids[order[i]];
        out_scores[i] = sc[order[i]];
        out_methods[i] = me[order[i]];
    }
    free(ids); free(sc); free(me); free(order);
    return n;
}

/* fp32 -> fp16 bits, round to nearest even (numpy astype(float16) semantics);
 * used by tests to build fp16 shards on the host. */
uint16_t oracle_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t m = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t half = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
    return (uint16_t)(sign | half);
}
