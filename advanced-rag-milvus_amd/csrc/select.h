// Exact selection kernels: block-wide radix select over unique 64-bit ranking
// keys (score desc, index asc).  Used twice per modality:
//   select_groups : top-C candidate groups from the scan's per-group maxima
//   select_topk   : final top-k rows from the refined (canonical) candidates
// plus the proof-of-exactness flag that ties the two together.
#pragma once
#include "common.h"

namespace hbmrag {

struct SelectScratch {
    unsigned int hist[256];
    int digit;
    int remaining;
    int count;
};

// K-th largest of n keys given by key(i), all threads of the block take part.
// Keys must be unique except for the invalid key 0.  Requires 1 <= K <= n.
template <typename KeyFn>
__device__ inline uint64_t block_kth_largest(KeyFn key, int64_t n, int K, SelectScratch& sh) {
    uint64_t prefix = 0, mask = 0;
    int remaining = K;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) sh.hist[i] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            uint64_t kk = key(i);
            if ((kk & mask) == prefix) atomicAdd(&sh.hist[(kk >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int cum = 0, d = 255;
            for (; d > 0; --d) {
                int c = (int)sh.hist[d];
                if (cum + c >= remaining) break;
                cum += c;
            }
            sh.digit = d;
            sh.remaining = remaining - cum;
        }
        __syncthreads();
        prefix |= (uint64_t)sh.digit << shift;
        mask |= 0xFFull << shift;
        remaining = sh.remaining;
        __syncthreads();
    }
    return prefix;
}

// One block per query.  gmax[q][n_groups] -> cand[q][C] (group ids, -1 padded)
// and a_cut[q]: the largest approximate score any row OUTSIDE the candidate
// groups can have (-inf when every group is a candidate).
__global__ __launch_bounds__(1024) void select_groups_kernel(const float* __restrict__ gmax,
                                                             int64_t n_groups, int C,
                                                             int32_t* __restrict__ cand,
                                                             float* __restrict__ a_cut) {
    __shared__ SelectScratch sh;
    const int q = blockIdx.x;
    const float* gm = gmax + (int64_t)q * n_groups;
    int32_t* out = cand + (int64_t)q * C;
    if (n_groups <= C) {
        for (int i = threadIdx.x; i < C; i += blockDim.x) out[i] = (i < n_groups) ? i : -1;
        if (threadIdx.x == 0) a_cut[q] = -__builtin_inff();
        return;
    }
    auto key = [&](int64_t i) { return rank_key(gm[i], (uint32_t)i); };
    // (C+1)-th largest group max bounds everything that is left out.
    const uint64_t t_next = block_kth_largest(key, n_groups, C + 1, sh);
    if (threadIdx.x == 0) {
        sh.count = 0;
        a_cut[q] = key_score(t_next);
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n_groups; i += blockDim.x) {
        if (key(i) > t_next) {
            int slot = atomicAdd(&sh.count, 1);
            out[slot] = (int32_t)i;  // exactly C keys are > the (C+1)-th largest
        }
    }
}

// One block per query.  n candidates (score, row; row < 0 = invalid) -> the
// best K by (score desc, row asc), sorted, as global ids.  flags[q] = 1 when
// the result is provably the exact top-K of the whole shard:
//   every group was a candidate, or nothing outside can qualify
//   (a_cut <= floor), or K rows were found and the K-th canonical score beats
//   a_cut by more than the scan's error bound.
// norm_mode: 0 = scores compare to a_cut as they are; 1 = divide by |q| first
// (inner-product metric: the scan works on the unit-normalised query).
__global__ __launch_bounds__(256) void select_topk_kernel(
    const float* __restrict__ cscore, const int32_t* __restrict__ crow, int n, int K,
    int64_t row_offset, const float* __restrict__ a_cut, float cut_floor, float eps_abs,
    float eps_rel, int norm_mode, const double* __restrict__ qn2, int64_t* __restrict__ out_ids,
    float* __restrict__ out_scores, int32_t* __restrict__ flags) {
    __shared__ SelectScratch sh;
    __shared__ uint64_t sel[HR_MAX_TOPK];
    const int q = blockIdx.x;
    const float* cs = cscore + (int64_t)q * n;
    const int32_t* cr = crow + (int64_t)q * n;
    auto key = [&](int64_t i) -> uint64_t {
        int32_t r = cr[i];
        return r < 0 ? 0ull : rank_key(cs[i], (uint32_t)r);
    };
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
    if (flags && threadIdx.x == 0) {
        const float cut = a_cut[q];
        int exact = 0;
        if (cut == -__builtin_inff() || cut <= cut_floor) {
            exact = 1;
        } else if (found == K) {
            uint64_t kth = sel[0];
            for (int j = 1; j < found; ++j) kth = sel[j] < kth ? sel[j] : kth;
            double sk = (double)key_score(kth);
            if (norm_mode == 1) {
                double nq = qn2[q];
                sk = nq > 0.0 ? sk / sqrt(nq) : 0.0;
            }
            double bound = (double)cut + (double)eps_abs + (double)eps_rel * fabs((double)cut);
            exact = sk > bound;
        }
        flags[q] = exact;
    }
}

}  // namespace hbmrag
