// Reciprocal-rank fusion and the cross-shard top-k merge.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kFuseMax = 3 * HR_MAX_TOPK;  // entries per query across the three lists

// One block (256 threads) per query.  Restates HybridRetriever._fuse_results
// (reference src/advanced_rag/retrieval.py:421-491) operation for operation:
//   rrf = 1.0 / (k + rank)            (float64 division)
//   score[id] += rrf * weight          (float64 multiply, then add; no FMA)
// lists are visited semantic -> sparse -> domain, an id keeps the slot of the
// list that saw it first (dict insertion order), and the final order is the
// stable descending sort of the scores (list.sort(reverse=True)): ties keep
// insertion order.  Outputs the first top_k fused entries.
__global__ __launch_bounds__(256) void rrf_fuse_kernel(
    const int64_t* __restrict__ ids_a, int ka, const int64_t* __restrict__ ids_b, int kb,
    const int64_t* __restrict__ ids_c, int kc, double wa, double wb, double wc, int rrf_k,
    int top_k, int64_t* __restrict__ out_ids, double* __restrict__ out_scores,
    int32_t* __restrict__ out_methods, int32_t* __restrict__ n_out) {
    __shared__ int64_t id[kFuseMax];
    __shared__ double sc[kFuseMax];
    __shared__ int owner[kFuseMax];   // for b/c entries: slot of the first list entry with the same id, or -1
    __shared__ int slot_of[kFuseMax]; // insertion slot of entries that open a new id
    __shared__ int meth[kFuseMax];
    __shared__ int s_na, s_nb, s_nc, s_new_c;
    const int q = blockIdx.x, tid = threadIdx.x;
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
        int o = -1;
        const int64_t me = id[na + i];
        for (int j = 0; j < na; ++j) if (id[j] == me) { o = j; break; }
        owner[na + i] = o;
    }
    __syncthreads();
    for (int i = tid; i < nc; i += 256) {
        int o = -1;
        const int64_t me = id[na + nb + i];
        for (int j = 0; j < na + nb; ++j) if (id[j] == me) { o = (owner[j] >= 0) ? owner[j] : j; break; }
        owner[na + nb + i] = o;
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
            else for (int j = 0; j < nb; ++j) if (owner[na + j] == e) { r = j; break; }
            if (r >= 0) {
                s = __dadd_rn(s, __dmul_rn(1.0 / (double)(rrf_k + r + 1), wb));
                m |= HR_METHOD_SPARSE;
            }
        }
        {
            int r = -1;
            if (e >= na + nb) r = e - na - nb;
            else for (int j = 0; j < nc; ++j) if (owner[na + nb + j] == e) { r = j; break; }
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

// Cross-shard merge: [n_lists][B][k_in] (score, id) -> best k_out per query by
// (score desc, id asc); ids < 0 are padding.  One block per query.
__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ scores,
                                                         const int64_t* __restrict__ ids, int n_lists,
                                                         int64_t score_stride, int64_t id_stride,
                                                         int B, int k_in, int k_out,
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
            rank += (t > s) || (t == s && u < i);
        }
        if (rank < k_out) {
            out_ids[(int64_t)q * k_out + rank] = i;
            out_scores[(int64_t)q * k_out + rank] = s;
        }
    }
    // padding: count valid entries (every thread, cheap) and clear the tail
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
__global__ __launch_bounds__(256) void rerank_linear_kernel(
    const int64_t* __restrict__ ids, const double* __restrict__ scores, const int32_t* __restrict__ methods,
    const int32_t* __restrict__ n_valid, const double* __restrict__ recency, int k_in, double base_w,
    double method_bonus, double recency_w, int k_out, int64_t* __restrict__ out_ids,
    double* __restrict__ out_scores, double* __restrict__ out_orig) {
    __shared__ double ns[HR_MAX_TOPK];
    const int q = blockIdx.x, tid = threadIdx.x;
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

}  // namespace hbmrag
