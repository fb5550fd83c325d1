// Exact selection kernels: block-wide radix select over unique 64-bit ranking
// keys (score desc, index asc).  Used twice per modality:
//   select_groups : top-C candidate groups from the scan's per-group maxima,
//                   two-level (buckets of 64 groups first) so a query touches
//                   ~n_groups/64 + 64*C values instead of n_groups
//   select_topk   : final top-k rows from the refined (canonical) candidates
// plus the proof-of-exactness flag that ties the two together.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kBucketGroups = 64;  // candidate groups per level-2 bucket (4096 rows)

struct SelectScratch {
    unsigned int hist[256];
    int digit;
    int remaining;
    int count;
    int n_live;      // select_groups_block: candidate slots that hold a group after the trim (the rest are -1)
    unsigned gk_bits; // ... and the K-th largest candidate maximum it trimmed against
};

// One 8-bit radix pass: histogram of digit `shift` over the keys that match
// (prefix, mask), then wave 0 finds the digit holding the `remaining`-th largest
// (4 bins per lane + a wave scan instead of a serial walk over 256 bins).
template <typename KeyFn>
__device__ inline void radix_pass(KeyFn key, int64_t n, uint64_t prefix, uint64_t mask, int shift, int remaining,
                                  SelectScratch& sh) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) sh.hist[i] = 0;
    __syncthreads();
    // (per-lane atomics: folding the lanes of a wave that share a digit into one add — the upper digits of a candidate set
    // are nearly all equal — measured 10 % SLOWER, profiles/r4_experiments/finish_hashfilter_aggregated_*.txt)
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t kk = key(i);
        if ((kk & mask) == prefix) atomicAdd(&sh.hist[(kk >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // lane l owns bins 255-4l .. 252-4l (descending), so an inclusive scan over
        // lanes is the count of keys in all higher bins.
        const int lane = threadIdx.x;
        const int top = 255 - 4 * lane;
        const int c0 = (int)sh.hist[top], c1 = (int)sh.hist[top - 1], c2 = (int)sh.hist[top - 2],
                  c3 = (int)sh.hist[top - 3];
        const int mine = c0 + c1 + c2 + c3;
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        const int before = incl - mine;  // keys in bins above this lane's
        if (before < remaining && incl >= remaining) {  // exactly one lane
            int cum = before, d = top, cd = c0;
            if (cum + c0 >= remaining) { d = top; cd = c0; }
            else if ((cum += c0) + c1 >= remaining) { d = top - 1; cd = c1; }
            else if ((cum += c1) + c2 >= remaining) { d = top - 2; cd = c2; }
            else { cum += c2; d = top - 3; cd = c3; }
            sh.digit = d;
            sh.remaining = remaining - cum;
            sh.count = cd;  // keys sharing the chosen digit (1 => the prefix already pins the key)
        }
    }
    __syncthreads();
}

// K-th largest of n keys given by key(i), all threads of the block take part.
// Keys are (ordered score << 32) | (~index): unique except for the invalid key 0.
// Requires 1 <= K <= n.  Four passes resolve the score; when exactly one key
// carries that score (the usual case) it is fetched directly, otherwise four
// more passes resolve the index bits among the ties.
// score_only: stop after the four passes that resolve the upper 32 bits (the caller wants the K-th largest SCORE, its keys
// need not be unique); the lower 32 bits of the result are zero.
template <typename KeyFn>
__device__ inline uint64_t block_kth_largest(KeyFn key, int64_t n, int K, SelectScratch& sh, bool score_only = false) {
    uint64_t prefix = 0, mask = 0;
    int remaining = K;
    int same = 0;
    for (int shift = 56; shift >= 32; shift -= 8) {
        radix_pass(key, n, prefix, mask, shift, remaining, sh);
        prefix |= (uint64_t)sh.digit << shift;
        mask |= 0xFFull << shift;
        remaining = sh.remaining;
        same = sh.count;
        __syncthreads();
    }
    if (score_only) return prefix;
    if (same == 1) {  // a single key has this score: find it
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t kk = key(i);
            if ((kk & mask) == prefix) sh.hist[0] = (unsigned int)(kk & 0xFFFFFFFFull);
        }
        __syncthreads();
        prefix |= (uint64_t)sh.hist[0];
        __syncthreads();
        return prefix;
    }
    for (int shift = 24; shift >= 0; shift -= 8) {
        radix_pass(key, n, prefix, mask, shift, remaining, sh);
        prefix |= (uint64_t)sh.digit << shift;
        mask |= 0xFFull << shift;
        remaining = sh.remaining;
        __syncthreads();
    }
    return prefix;
}

// Level-2 maxima: a wave takes kBucketsPerWave consecutive buckets of 64 groups of one query and
// issues all of their (coalesced, 256-byte) loads before the first reduction, so the pass streams the
// table of group maxima instead of paying one round trip per bucket.
constexpr int kBucketsPerWave = 8;

// The selection kernels take their arguments per MODALITY (dense list, sparse list): the two finishing chains of a hybrid
// search are independent and each of these launches covers only B blocks, so the hybrid path issues ONE launch for both
// (blockIdx.y / .z = modality) instead of two in a row — three dependent launches less on the stream that finishes a
// batch.  Single-modality searches pass n = 1.
struct GroupSelArgs {
    const float* gmax;    // [B][n_groups] group maxima of the scan
    float* bmax;          // [B][n_buckets] level-2 maxima (written by bucket_max_kernel when two_level)
    int64_t n_groups, n_buckets;
    int C;                // candidate groups per query
    int two_level;        // n_groups > C && n_buckets > C
    int32_t* cand;        // [B][C] out
    float* a_cut;         // [B] out
    // Data-dependent candidate set (round 4): with K_trim = the length of the list being built, a group whose maximum lies
    // more than twice the scan's error bound below the K_trim-th largest group maximum cannot hold a row of the top K_trim
    // (K_trim groups each hold a row at or above that maximum, up to the error; every row of the other group lies below it,
    // up to the error) — such candidates are dropped (slot = -1), the kept ones moved to the front in descending order of
    // their maxima, and a_cut becomes the largest dropped maximum.  eps_* as in TopkArgs.  0 = keep all C.
    int K_trim;
    float eps_abs, eps_rel;
    const float* eps_abs_q;
};
struct GroupSelPair {
    GroupSelArgs m[2];
    int n;
};

__global__ __launch_bounds__(256) void bucket_max_kernel(GroupSelPair p) {
    const GroupSelArgs& a = p.m[blockIdx.z];
    if (!a.two_level) return;
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kBucketsPerWave;
    if (b0 >= a.n_buckets) return;
    const float* gm = a.gmax + (int64_t)q * a.n_groups;
    float v[kBucketsPerWave];
#pragma unroll
    for (int i = 0; i < kBucketsPerWave; ++i) {
        const int64_t g = (b0 + i) * kBucketGroups + lane;
        v[i] = g < a.n_groups ? gm[g] : -__builtin_inff();
    }
    float mine = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < kBucketsPerWave; ++i) {
        const float m = wave_max(v[i]);
        if (lane == i) mine = m;
    }
    if (lane < kBucketsPerWave && b0 + lane < a.n_buckets) a.bmax[(int64_t)q * a.n_buckets + b0 + lane] = mine;
}

// One block per query.  gmax[q][n_groups] (+ bmax[q][n_buckets]) -> cand[q][C]
// (group ids, -1 padded) and a_cut[q]: an upper bound on the approximate score
// of every row OUTSIDE the candidate groups (-inf when every group is a
// candidate).  Groups outside the selected buckets are bounded by the best
// unselected bucket maximum, groups inside them by the (C+1)-th candidate key.
// Block function (any block size that is a multiple of 64): `bm` = the bucket maxima of THIS query (global table row or
// an LDS copy), `out` = the C candidate slots of this query (global or LDS).
// (Staging the maxima of the selected buckets' groups in LDS once, instead of fetching them in each level-1 pass, did not
// pay: the passes are bound by their LDS work and barriers, not by those L2 hits — profiles/r4_experiments/.)
__device__ inline void select_groups_block(const GroupSelArgs& a, int q, const float* bm, int32_t* out,
                                           SelectScratch& sh, int32_t* sel_bucket) {
    const int64_t n_groups = a.n_groups, n_buckets = a.n_buckets;
    const int C = a.C;
    float* __restrict__ a_cut = a.a_cut;
    const float* gm = a.gmax + (int64_t)q * n_groups;
    const float NEG_INF = -__builtin_inff();
    if (n_groups <= C) {
        for (int i = threadIdx.x; i < C; i += blockDim.x) out[i] = (i < n_groups) ? i : -1;
        if (threadIdx.x == 0) {
            a_cut[q] = NEG_INF;
            sh.n_live = C;
        }
        return;
    }
    // ---- level 2: the C buckets with the largest maxima
    float bucket_cut = NEG_INF;
    int nb;
    if (n_buckets <= C) {
        nb = (int)n_buckets;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) sel_bucket[i] = i;
    } else {
        auto bkey = [&](int64_t i) { return rank_key(bm[i], (uint32_t)i); };
        const uint64_t t_b = block_kth_largest(bkey, n_buckets, C + 1, sh);
        bucket_cut = key_score(t_b);
        if (threadIdx.x == 0) sh.count = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n_buckets; i += blockDim.x)
            if (bkey(i) > t_b) sel_bucket[atomicAdd(&sh.count, 1)] = (int32_t)i;
        nb = C;
    }
    __syncthreads();
    // ---- level 1: the C best groups inside those buckets
    const int64_t n_cand = (int64_t)nb * kBucketGroups;
    auto gkey = [&](int64_t i) -> uint64_t {
        const int64_t g = (int64_t)sel_bucket[i >> 6] * kBucketGroups + (i & 63);
        return g < n_groups ? rank_key(gm[g], (uint32_t)g) : 0ull;  // key 0 = padding, never selected
    };
    const uint64_t t_g = block_kth_largest(gkey, n_cand, C + 1, sh);
    if (threadIdx.x == 0) {
        sh.count = 0;
        const float inner = t_g ? key_score(t_g) : NEG_INF;
        a_cut[q] = fmaxf(inner, bucket_cut);
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n_cand; i += blockDim.x) {
        const uint64_t kk = gkey(i);
        if (kk > t_g) {  // exactly C keys beat the (C+1)-th largest (n_groups > C guarantees C+1 valid keys)
            const int slot = atomicAdd(&sh.count, 1);
            if (slot < C) out[slot] = (int32_t)key_row(kk);
        }
    }
    if (threadIdx.x == 0) sh.n_live = C;
    // ---- trim: keep the candidates that can still hold a row of the top K_trim (see GroupSelArgs)
    const int K = a.K_trim;
    if (K <= 0 || K >= C || C > (int)blockDim.x) return;      // one candidate per thread below
    __syncthreads();
    float* vals = reinterpret_cast<float*>(sel_bucket);         // the selected buckets are no longer needed (C <= 512 entries)
    const int i = threadIdx.x;
    int32_t grp = -1;
    float v = NEG_INF;
    if (i < C) {
        grp = out[i];
        v = grp >= 0 ? gm[grp] : NEG_INF;
        vals[i] = v;
    }
    __syncthreads();
    int rank = 0;                                               // position by (maximum desc, slot asc): a permutation of 0 .. C-1
    if (i < C) {
        for (int j = 0; j < C; ++j) {
            const float w = vals[j];
            rank += (w > v) || (w == v && j < i);
        }
        if (rank == K - 1) sh.gk_bits = __builtin_bit_cast(unsigned, v);
    }
    __syncthreads();
    const float gk = __builtin_bit_cast(float, sh.gk_bits);
    const float eps = a.eps_abs + (a.eps_abs_q ? a.eps_abs_q[q] : 0.f) + a.eps_rel * fabsf(gk);
    const float thr = gk - 2.0f * eps - 1e-6f * fabsf(gk);      // (gk = -inf: fewer than K live groups — nothing is dropped)
    const bool keep = i < C && grp >= 0 && (v >= thr || !(gk > NEG_INF));
    if (threadIdx.x == 0) {
        sh.n_live = 0;
        sh.hist[1] = 0;   // ord_f32 of the largest dropped maximum (0 = nothing dropped)
    }
    __syncthreads();
    if (i < C) {
        if (keep) atomicMax(&sh.n_live, rank + 1);              // the kept ones are exactly the best n_live by rank
        else if (grp >= 0) atomicMax(&sh.hist[1], ord_f32(v));
        out[rank] = keep ? grp : -1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && sh.hist[1] != 0) a_cut[q] = fmaxf(a_cut[q], unord_f32(sh.hist[1]));
}

__global__ __launch_bounds__(1024) void select_groups_kernel(GroupSelPair p) {
    __shared__ SelectScratch sh;
    __shared__ int32_t sel_bucket[HR_MAX_TOPK * 2];  // C <= k + k/2 rounded to 16 (<= 400)
    const GroupSelArgs& a = p.m[blockIdx.y];
    const int q = blockIdx.x;
    select_groups_block(a, q, a.bmax + (int64_t)q * a.n_buckets, a.cand + (int64_t)q * a.C, sh, sel_bucket);
}

// One block per query.  n candidates (score, row; row < 0 = invalid) -> the
// best K by (score desc, row asc), sorted, as global ids.  flags[q] = 1 when
// the result is provably the exact top-K of the whole shard:
//   every group was a candidate, or nothing outside can qualify
//   (a_cut <= floor), or K rows were found and the K-th canonical score beats
//   a_cut by more than the scan's error bound.
// norm_mode: 0 = scores compare to a_cut as they are; 1 = divide by |q| first
// (inner-product metric: the scan works on the unit-normalised query).
struct TopkArgs {
    const float* cscore;   // [B][n] canonical candidate scores
    const int32_t* crow;   // [B][n] candidate rows (< 0 = invalid)
    int n, K;
    int64_t row_offset;
    const float* a_cut;
    float cut_floor, eps_abs;
    const float* eps_abs_q;
    float eps_rel;
    int norm_mode;
    const double* qn2;
    int64_t* out_ids;
    float* out_scores;
    int32_t* flags;
};
struct TopkPair {
    TopkArgs m[2];
    int n;
};

// Block function: key(i), i < n = the ranking key of candidate i (0 = invalid) from wherever the refine left it.
template <typename KeyFn>
__device__ inline void select_topk_block(const TopkArgs& a, int q, KeyFn key, SelectScratch& sh, uint64_t* sel) {
    const int n = a.n, K = a.K;
    const int64_t row_offset = a.row_offset;
    int64_t* __restrict__ out_ids = a.out_ids;
    float* __restrict__ out_scores = a.out_scores;
    const int Ke = K < n ? K : n;
    uint64_t thr = 0;
    if (Ke > 0) thr = block_kth_largest(key, n, Ke, sh);
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint64_t kk = key(i);
        if (kk != 0 && kk >= thr) {
            int slot = atomicAdd(&sh.count, 1);
            if (slot < HR_MAX_TOPK) sel[slot] = kk;
        }
    }
    __syncthreads();
    const int found = sh.count < Ke ? sh.count : Ke;
    // rank by counting (keys are unique)
    for (int i = threadIdx.x; i < found; i += blockDim.x) {
        const uint64_t me = sel[i];
        int rank = 0;
        for (int j = 0; j < found; ++j) rank += sel[j] > me;
        out_ids[(int64_t)q * K + rank] = (int64_t)key_row(me) + row_offset;
        out_scores[(int64_t)q * K + rank] = key_score(me);
    }
    for (int i = found + threadIdx.x; i < K; i += blockDim.x) {
        out_ids[(int64_t)q * K + i] = -1;
        out_scores[(int64_t)q * K + i] = 0.f;
    }
    if (a.flags && threadIdx.x == 0) {
        const float cut = a.a_cut[q];
        int exact = 0;
        if (cut == -__builtin_inff() || cut <= a.cut_floor) {
            exact = 1;
        } else if (found == K) {
            uint64_t kth = sel[0];
            for (int j = 1; j < found; ++j) kth = sel[j] < kth ? sel[j] : kth;
            double sk = (double)key_score(kth);
            if (a.norm_mode == 1) {
                double nq = a.qn2[q];
                sk = nq > 0.0 ? sk / sqrt(nq) : 0.0;
            }
            double bound = (double)cut + (double)a.eps_abs + (a.eps_abs_q ? (double)a.eps_abs_q[q] : 0.0) +
                           (double)a.eps_rel * fabs((double)cut);
            exact = sk > bound;
        }
        a.flags[q] = exact;
    }
}

__global__ __launch_bounds__(1024) void select_topk_kernel(TopkPair p) {
    __shared__ SelectScratch sh;
    __shared__ uint64_t sel[HR_MAX_TOPK];
    const TopkArgs& a = p.m[blockIdx.y];
    const int q = blockIdx.x;
    const float* cs = a.cscore + (int64_t)q * a.n;
    const int32_t* cr = a.crow + (int64_t)q * a.n;
    auto key = [&](int64_t i) -> uint64_t {
        int32_t r = cr[i];
        return r < 0 ? 0ull : rank_key(cs[i], (uint32_t)r);
    };
    select_topk_block(a, q, key, sh, sel);
}

}  // namespace hbmrag
