// Reciprocal-rank fusion and the cross-shard top-k merge.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kFuseMax = 3 * HR_MAX_TOPK;  // entries per query across the three lists
constexpr size_t kMergeLdsMax = 60 * 1024;  // LDS staging of the cross-shard merge (~5000 entries); larger merges take the global form

// One block (256 threads) per query.  Restates HybridRetriever._fuse_results
// (reference src/advanced_rag/retrieval.py:421-491) operation for operation:
//   rrf = 1.0 / (k + rank)            (float64 division)
//   score[id] += rrf * weight          (float64 multiply, then add; no FMA)
// lists are visited semantic -> sparse -> domain, an id keeps the slot of the
// list that saw it first (dict insertion order), and the final order is the
// stable descending sort of the scores (list.sort(reverse=True)): ties keep
// insertion order.  Outputs the first top_k fused entries.
// (The body is a device function so that the stand-alone kernel and the fused post-exchange kernel below run the
// very same instructions; q = the query this 256-thread block fuses.)
__device__ inline void rrf_fuse_block(
    int q, const int64_t* __restrict__ ids_a, int ka, const int64_t* __restrict__ ids_b, int kb,
    const int64_t* __restrict__ ids_c, int kc, double wa, double wb, double wc, int rrf_k,
    int top_k, int64_t* __restrict__ out_ids, double* __restrict__ out_scores,
    int32_t* __restrict__ out_methods, int32_t* __restrict__ n_out) {
    __shared__ int64_t id[kFuseMax];
    __shared__ double sc[kFuseMax];
    __shared__ int owner[kFuseMax];   // for b/c entries: slot of the first list entry with the same id, or -1
    __shared__ int slot_of[kFuseMax]; // insertion slot of entries that open a new id
    __shared__ int meth[kFuseMax];
    __shared__ int s_na, s_nb, s_nc, s_new_c;
    const int tid = threadIdx.x;
    const int64_t* la = ids_a + (int64_t)q * ka;
    const int64_t* lb = kb ? ids_b + (int64_t)q * kb : nullptr;
    const int64_t* lc = kc ? ids_c + (int64_t)q * kc : nullptr;
    // list lengths = position of the first negative id (lists are -1 padded at the tail)
    if (tid == 0) { s_na = ka; s_nb = kb; s_nc = kc; }
    __syncthreads();
    for (int i = tid; i < ka; i += 256) if (la[i] < 0) atomicMin(&s_na, i);
    for (int i = tid; i < kb; i += 256) if (lb[i] < 0) atomicMin(&s_nb, i);
    for (int i = tid; i < kc; i += 256) if (lc[i] < 0) atomicMin(&s_nc, i);
    __syncthreads();
    const int na = s_na, nb = s_nb, nc = s_nc;
    // stage ids: [0,na) = a, [na, na+nb) = b, then c
    for (int i = tid; i < na; i += 256) id[i] = la[i];
    for (int i = tid; i < nb; i += 256) id[na + i] = lb[i];
    for (int i = tid; i < nc; i += 256) id[na + nb + i] = lc[i];
    __syncthreads();
    // a entries own themselves (ids inside one list are unique: one hit per row)
    for (int i = tid; i < na; i += 256) owner[i] = -1;
    for (int i = tid; i < nb; i += 256) {
        int o = na;   // first a entry with this id (ids inside a list are unique); no early exit: the reads are independent
        const int64_t me = id[na + i];
        for (int j = 0; j < na; ++j) o = (id[j] == me && j < o) ? j : o;
        owner[na + i] = o < na ? o : -1;
    }
    __syncthreads();
    for (int i = tid; i < nc; i += 256) {
        int first = na + nb;   // first a/b entry with this id
        const int64_t me = id[na + nb + i];
        for (int j = 0; j < na + nb; ++j) first = (id[j] == me && j < first) ? j : first;
        owner[na + nb + i] = first < na + nb ? ((owner[first] >= 0) ? owner[first] : first) : -1;
    }
    __syncthreads();
    // insertion slots: a -> 0..na-1; new b ids follow in b order; new c ids after them.
    // slot = na + #{earlier new entries}: counted per entry (lists are <= 256 long).
    if (tid == 0) s_new_c = 0;
    __syncthreads();
    for (int i = tid; i < na; i += 256) slot_of[i] = i;
    for (int e = na + tid; e < na + nb + nc; e += 256) {
        int sl = -1;
        if (owner[e] < 0) {
            sl = na;
            for (int j = na; j < e; ++j) sl += owner[j] < 0;
            atomicAdd(&s_new_c, 1);
        }
        slot_of[e] = sl;
    }
    __syncthreads();
    const int n_ids = na + s_new_c;
    // scores, accumulated per owning entry in list order a, b, c
    for (int e = tid; e < na + nb + nc; e += 256) {
        if (owner[e] >= 0) continue;  // this entry's id was opened by an earlier list
        double s = 0.0;
        int m = 0;
        const int64_t me = id[e];
        if (e < na) {
            s = __dadd_rn(s, __dmul_rn(1.0 / (double)(rrf_k + e + 1), wa));
            m |= HR_METHOD_SEMANTIC;
        }
        if (e < na + nb) {
            // contribution from list b: own entry, or the b entry that points here
            int r = -1;
            if (e >= na) r = e - na;
            else for (int j = nb - 1; j >= 0; --j) r = (owner[na + j] == e) ? j : r;   // the b entry that points here (at most one)
            if (r >= 0) {
                s = __dadd_rn(s, __dmul_rn(1.0 / (double)(rrf_k + r + 1), wb));
                m |= HR_METHOD_SPARSE;
            }
        }
        {
            int r = -1;
            if (e >= na + nb) r = e - na - nb;
            else for (int j = nc - 1; j >= 0; --j) r = (owner[na + nb + j] == e) ? j : r;
            if (r >= 0) {
                s = __dadd_rn(s, __dmul_rn(1.0 / (double)(rrf_k + r + 1), wc));
                m |= HR_METHOD_DOMAIN;
            }
        }
        (void)me;
        sc[e] = s;
        meth[e] = m;
    }
    __syncthreads();
    // stable descending order by counting: rank = #{better score} + #{equal score, earlier slot}
    for (int e = tid; e < na + nb + nc; e += 256) {
        if (owner[e] >= 0) continue;
        const double s = sc[e];
        const int my_slot = slot_of[e];
        int rank = 0;
        for (int j = 0; j < na + nb + nc; ++j) {
            if (owner[j] >= 0) continue;
            const double t = sc[j];
            rank += (t > s) || (t == s && slot_of[j] < my_slot);
        }
        if (rank < top_k) {
            out_ids[(int64_t)q * top_k + rank] = id[e];
            out_scores[(int64_t)q * top_k + rank] = s;
            out_methods[(int64_t)q * top_k + rank] = meth[e];
        }
    }
    const int n_fused = n_ids < top_k ? n_ids : top_k;
    for (int i = n_fused + tid; i < top_k; i += 256) {
        out_ids[(int64_t)q * top_k + i] = -1;
        out_scores[(int64_t)q * top_k + i] = 0.0;
        out_methods[(int64_t)q * top_k + i] = 0;
    }
    if (tid == 0) n_out[q] = n_fused;
}

__global__ __launch_bounds__(256) void rrf_fuse_kernel(
    const int64_t* __restrict__ ids_a, int ka, const int64_t* __restrict__ ids_b, int kb,
    const int64_t* __restrict__ ids_c, int kc, double wa, double wb, double wc, int rrf_k,
    int top_k, int64_t* __restrict__ out_ids, double* __restrict__ out_scores,
    int32_t* __restrict__ out_methods, int32_t* __restrict__ n_out) {
    rrf_fuse_block(blockIdx.x, ids_a, ka, ids_b, kb, ids_c, kc, wa, wb, wc, rrf_k, top_k, out_ids, out_scores,
                   out_methods, n_out);
}

// Cross-shard merge of one query: n_lists per-shard lists of k_in (score, id) pairs, each SORTED by score
// descending (what the top-k kernels write; ids < 0 pad the tail) -> the best k_out by (score desc, id asc).
// The lists are staged in LDS (s_id / s_sc: n_lists * k_in entries, s_len: n_lists words) and every entry computes its
// rank in the merged order from two counts per list (entries with a larger score; entries with the same score, which
// are then compared one by one: id ascending, and — for exact duplicates, which only a caller that feeds the same list
// twice produces — list number, then position, so that ranks are always a permutation).  The counts come from
// ~2 sqrt(k_in) independent LDS reads per list (splitters, then one stride) instead of the (n_lists * k_in)^2 global
// reads of the first form of this kernel: 8 lists of k' = 40 cost ~30k LDS reads per query against 100k global ones,
// 8 x 200 ~ 370k against 2.5M, and none of them waits for another.
__device__ inline void merge_runs_block(int q, const float* __restrict__ scores, const int64_t* __restrict__ ids,
                                        int n_lists, int64_t score_stride, int64_t id_stride, int k_in, int k_out,
                                        int64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                        int64_t* s_id, float* s_sc, int* s_len) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = n_lists * k_in;
    for (int r = tid; r < n_lists; r += nt) s_len[r] = k_in;
    __syncthreads();
    for (int e = tid; e < n; e += nt) {
        const int r = e / k_in, p = e - r * k_in;
        const int64_t o = (int64_t)q * k_in + p;
        const int64_t i = ids[(int64_t)r * id_stride + o];
        s_id[e] = i;
        s_sc[e] = scores[(int64_t)r * score_stride + o];
        if (i < 0) atomicMin(&s_len[r], p);
    }
    __syncthreads();
    int total = 0;
    for (int r = 0; r < n_lists; ++r) total += s_len[r];
    int stride = 1;
    while (stride * stride < k_in) ++stride;   // ~sqrt(k_in): splitters + one stride of entries per list
    for (int e = tid; e < n; e += nt) {
        const int r = e / k_in, p = e - r * k_in;
        if (p >= s_len[r]) continue;
        const float s = s_sc[e];
        const int64_t i = s_id[e];
        int rank = 0;
        for (int r2 = 0; r2 < n_lists; ++r2) {
            const float* sc = s_sc + r2 * k_in;
            const int64_t* id = s_id + r2 * k_in;
            const int len = s_len[r2];
            // entries of list r2 with a larger score (gt) and with a score at least as large (ge): two levels of
            // INDEPENDENT LDS reads — every stride-th entry, then the entries of one stride — instead of the dependent
            // steps of a binary search: the lists are sorted by score, so the splitters bracket both boundaries
            int b_gt = 0, b_ge = 0;   // splitters (entries stride-1, 2*stride-1, ...) that are > s / >= s
            for (int t = stride - 1; t < len; t += stride) {
                const float v = sc[t];
                b_gt += v > s;
                b_ge += v >= s;
            }
            int gt = b_gt * stride, ge = b_ge * stride;
            for (int t = b_gt * stride, e2 = min(len, (b_gt + 1) * stride); t < e2; ++t) gt += sc[t] > s;
            for (int t = b_ge * stride, e2 = min(len, (b_ge + 1) * stride); t < e2; ++t) ge += sc[t] >= s;
            rank += gt;
            for (int t = gt; t < ge; ++t) {   // equal scores: id ascending, then list, then position
                const int64_t u = id[t];
                rank += (u < i) || (u == i && (r2 < r || (r2 == r && t < p)));
            }
        }
        if (rank < k_out) {
            out_ids[(int64_t)q * k_out + rank] = i;
            out_scores[(int64_t)q * k_out + rank] = s;
        }
    }
    for (int j = (total < k_out ? total : k_out) + tid; j < k_out; j += nt) {
        out_ids[(int64_t)q * k_out + j] = -1;
        out_scores[(int64_t)q * k_out + j] = 0.f;
    }
}

// LDS bytes merge_runs_block needs for n entries in n_lists lists (ids 8-byte aligned first).
__host__ __device__ inline size_t merge_lds_bytes(int n_lists, int k_in) {
    return (size_t)n_lists * k_in * 12 + (size_t)n_lists * 4;
}

__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ scores,
                                                         const int64_t* __restrict__ ids, int n_lists,
                                                         int64_t score_stride, int64_t id_stride,
                                                         int k_in, int k_out,
                                                         int64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores) {
    extern __shared__ int64_t merge_lds[];
    const int n = n_lists * k_in;
    int64_t* s_id = merge_lds;
    float* s_sc = reinterpret_cast<float*>(s_id + n);
    int* s_len = reinterpret_cast<int*>(s_sc + n);
    merge_runs_block(blockIdx.x, scores, ids, n_lists, score_stride, id_stride, k_in, k_out, out_ids, out_scores, s_id,
                     s_sc, s_len);
}

// Fallback for merges too large for LDS (n_lists * k_in beyond ~5000 entries): rank by counting, straight from
// global memory; accepts unsorted lists.  One block per query.
__global__ __launch_bounds__(256) void merge_topk_big_kernel(const float* __restrict__ scores,
                                                             const int64_t* __restrict__ ids, int n_lists,
                                                             int64_t score_stride, int64_t id_stride,
                                                             int k_in, int k_out,
                                                             int64_t* __restrict__ out_ids,
                                                             float* __restrict__ out_scores) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_lists * k_in;
    auto at = [&](int e, float& s, int64_t& i) {
        const int64_t o = (int64_t)q * k_in + (e % k_in);
        s = scores[(int64_t)(e / k_in) * score_stride + o];
        i = ids[(int64_t)(e / k_in) * id_stride + o];
    };
    int n_valid = 0;
    for (int e = tid; e < n; e += 256) {
        float s; int64_t i;
        at(e, s, i);
        if (i < 0) continue;
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            float t; int64_t u;
            at(j, t, u);
            if (u < 0) continue;
            rank += (t > s) || (t == s && (u < i || (u == i && j < e)));
        }
        if (rank < k_out) {
            out_ids[(int64_t)q * k_out + rank] = i;
            out_scores[(int64_t)q * k_out + rank] = s;
        }
    }
    for (int j = 0; j < n; ++j) {
        float t; int64_t u;
        at(j, t, u);
        n_valid += u >= 0;
    }
    for (int i = (n_valid < k_out ? n_valid : k_out) + tid; i < k_out; i += 256) {
        out_ids[(int64_t)q * k_out + i] = -1;
        out_scores[(int64_t)q * k_out + i] = 0.f;
    }
}

// Linear re-scorer of LearnedRanker.score (reference ranker.py:109-125):
//   new = base_w*score + method_bonus*len(retrieval_methods) + recency_w*recency
// in float64, then HybridRetriever.rerank's stable descending sort and cut
// (retrieval.py:556-563).  One block per query; entries [B][k_in], n[B] valid.
__device__ inline void rerank_linear_block(
    int q, const int64_t* __restrict__ ids, const double* __restrict__ scores, const int32_t* __restrict__ methods,
    const int32_t* __restrict__ n_valid, const double* __restrict__ recency, int k_in, double base_w,
    double method_bonus, double recency_w, int k_out, int64_t* __restrict__ out_ids,
    double* __restrict__ out_scores, double* __restrict__ out_orig) {
    __shared__ double ns[HR_MAX_TOPK];
    const int tid = threadIdx.x;
    const int n = min(n_valid[q], k_in);
    for (int e = tid; e < n; e += 256) {
        const int64_t o = (int64_t)q * k_in + e;
        const double mc = (double)__popc((unsigned)methods[o]);
        const double rec = recency ? recency[o] : 0.0;
        double v = __dmul_rn(base_w, scores[o]);
        v = __dadd_rn(v, __dmul_rn(method_bonus, mc));
        v = __dadd_rn(v, __dmul_rn(recency_w, rec));
        ns[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < n; e += 256) {
        const double v = ns[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (ns[j] > v) || (ns[j] == v && j < e);
        if (rank < k_out) {
            const int64_t o = (int64_t)q * k_in + e;
            out_ids[(int64_t)q * k_out + rank] = ids[o];
            out_scores[(int64_t)q * k_out + rank] = v;
            out_orig[(int64_t)q * k_out + rank] = scores[o];
        }
    }
    for (int i = (n < k_out ? n : k_out) + tid; i < k_out; i += 256) {
        out_ids[(int64_t)q * k_out + i] = -1;
        out_scores[(int64_t)q * k_out + i] = 0.0;
        out_orig[(int64_t)q * k_out + i] = 0.0;
    }
}

__global__ __launch_bounds__(256) void rerank_linear_kernel(
    const int64_t* __restrict__ ids, const double* __restrict__ scores, const int32_t* __restrict__ methods,
    const int32_t* __restrict__ n_valid, const double* __restrict__ recency, int k_in, double base_w,
    double method_bonus, double recency_w, int k_out, int64_t* __restrict__ out_ids,
    double* __restrict__ out_scores, double* __restrict__ out_orig) {
    rerank_linear_block(blockIdx.x, ids, scores, methods, n_valid, recency, k_in, base_w, method_bonus, recency_w, k_out,
                        out_ids, out_scores, out_orig);
}

// Everything that follows the per-shard lists of a query batch, in ONE launch (one 256-thread block per query):
// [merge of the n_lists exchanged lists of every modality] -> reciprocal-rank fusion -> [learned-ranker rerank].
// The steps are the device functions above, run back to back on data the block itself has just written (global
// memory, visible to the whole workgroup after the barrier), so the results are bit-identical to the three separate
// launches — which cost the finishing stream of a sharded search two kernel boundaries per modality more.
struct PostArgs {
    hr_post_args a;
};

__global__ __launch_bounds__(256) void post_lists_kernel(PostArgs pa) {
    extern __shared__ int64_t merge_lds[];
    const hr_post_args& a = pa.a;
    const int q = blockIdx.x;
    if (a.agg_flags && a.n_lists > 1)
        for (int i = q * 256 + threadIdx.x; i < a.n_flag_rows; i += gridDim.x * 256) {
            int32_t f = a.flags[i];
            for (int l = 1; l < a.n_lists; ++l) f = min(f, a.flags[(int64_t)l * a.flag_stride + i]);
            a.agg_flags[i] = f;
        }
    const int64_t* fuse_ids[3];
    for (int m = 0; m < 3; ++m) {
        fuse_ids[m] = a.k_in[m] ? a.ids[m] : nullptr;
        if (a.n_lists > 1 && a.k_in[m]) {
            const int n = a.n_lists * a.k_in[m];
            int64_t* s_id = merge_lds;
            float* s_sc = reinterpret_cast<float*>(s_id + n);
            int* s_len = reinterpret_cast<int*>(s_sc + n);
            merge_runs_block(q, a.scores[m], a.ids[m], a.n_lists, a.score_stride, a.id_stride, a.k_in[m], a.k_fuse[m],
                             a.merged_ids[m], a.merged_scores[m], s_id, s_sc, s_len);
            fuse_ids[m] = a.merged_ids[m];
            __syncthreads();  // the merged list is read back below; the LDS staging is reused by the next modality
        }
    }
    double w0 = a.w[0], w1 = a.w[1], w2 = a.w[2];
    if (a.w_query) {  // per-request weights (a weight_adapter's): the same arithmetic, other operands
        w0 = a.w_query[3 * (int64_t)q];
        w1 = a.w_query[3 * (int64_t)q + 1];
        w2 = a.w_query[3 * (int64_t)q + 2];
    }
    rrf_fuse_block(q, fuse_ids[0], a.k_fuse[0], fuse_ids[1], a.k_in[1] ? a.k_fuse[1] : 0, fuse_ids[2],
                   a.k_in[2] ? a.k_fuse[2] : 0, w0, w1, w2, a.rrf_k, a.top_k, a.fused_ids, a.fused_scores,
                   a.fused_methods, a.fused_n);
    if (!a.rerank) return;
    __syncthreads();
    rerank_linear_block(q, a.fused_ids, a.fused_scores, a.fused_methods, a.fused_n, a.recency, a.top_k, a.base_w,
                        a.method_bonus, a.recency_w, a.k_out, a.rr_ids, a.rr_scores, a.rr_orig);
}

}  // namespace hbmrag
