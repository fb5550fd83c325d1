// Sparse shard kernels: doc-range-partitioned postings build, the
// term-at-a-time scan with LDS accumulators, and the canonical refine over CSR.
//
// Replaces Milvus' SPARSE_INVERTED_INDEX / IP search on "sparse_index"
// (reference src/advanced_rag/indexing.py:156-167, :487-498, :503-525).  The
// weighting (BM25, SPLADE, the reference's |N(0,1)| placeholder) lives in the
// vectors; the device computes the sparse inner product.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kRangeDocs = 4096;                      // docs per range = LDS accumulator length
constexpr int kRangeGroups = kRangeDocs / kGroupRows;  // 64 candidate groups per range
constexpr int kScanTermChunk = 256;                   // query terms staged per pass

// ---- build: CSR (doc-major) -> range-major postings ---------------------------
// rt_off[range][t] counts, then (after the per-range exclusive scan) offsets of
// term t's run inside the range's posting block.
__global__ void sparse_count_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                    int64_t n_docs, int64_t V1, unsigned int* __restrict__ rt_off) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    unsigned int* row = rt_off + (d / kRangeDocs) * V1;
    for (int64_t e = indptr[d]; e < indptr[d + 1]; ++e) atomicAdd(&row[idx[e]], 1u);
}

// One block per range: in-place exclusive scan of V counts; slot V gets the total.
__global__ __launch_bounds__(1024) void sparse_scan_offsets_kernel(unsigned int* __restrict__ rt_off, int64_t V1,
                                                                   unsigned long long* __restrict__ range_total) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry;
    unsigned int* row = rt_off + (int64_t)blockIdx.x * V1;
    const int64_t V = V1 - 1;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = 0; base < V; base += blockDim.x) {
        int64_t i = base + threadIdx.x;
        unsigned int v = (i < V) ? row[i] : 0u;
        unsigned int x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        unsigned int wbase = 0;
        for (int j = 0; j < w; ++j) wbase += wsum[j];
        unsigned int excl = carry + wbase + x - v;
        if (i < V) row[i] = excl;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        row[V] = carry;
        range_total[blockIdx.x] = carry;
    }
}

// Scatter postings.  cursor = copy of rt_off; order inside a run follows the
// atomics (the scan's sums are order-independent up to fp32 rounding, which
// the refine step makes irrelevant).
__global__ void sparse_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                   const float* __restrict__ val, int64_t n_docs, int64_t V1,
                                   unsigned int* __restrict__ cursor, const int64_t* __restrict__ range_base,
                                   uint16_t* __restrict__ post_doc, float* __restrict__ post_val) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    const int64_t range = d / kRangeDocs;
    unsigned int* cur = cursor + range * V1;
    const int64_t base = range_base[range];
    const uint16_t local = (uint16_t)(d - range * kRangeDocs);
    for (int64_t e = indptr[d]; e < indptr[d + 1]; ++e) {
        unsigned int slot = atomicAdd(&cur[idx[e]], 1u);
        post_doc[base + slot] = local;
        post_val[base + slot] = val[e];
    }
}

// ---- scan: grid (n_ranges, B), 256 threads --------------------------------------
// The block owns docs [range*4096, +4096) of one query: accumulators live in
// LDS (16 KiB), every posting of the query's terms inside the range is applied
// with an LDS float atomic, and only the per-64-doc maxima leave the CU.
// Algorithmic HBM bytes per (query, range): sum over query terms of run_len * 6
// (uint16 doc + fp32 weight) + 2*4 per term for the run bounds + 64*4 out.
__global__ __launch_bounds__(256) void sparse_scan_kernel(
    const unsigned int* __restrict__ rt_off, int64_t V1, const int64_t* __restrict__ range_base,
    const uint16_t* __restrict__ post_doc, const float* __restrict__ post_val,
    const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
    const float* __restrict__ q_val, const uint8_t* __restrict__ rowmask, int64_t n_docs,
    int64_t n_groups, float* __restrict__ gmax) {
    __shared__ float acc[kRangeDocs];
    __shared__ unsigned int run_lo[kScanTermChunk];
    __shared__ unsigned int run_pre[kScanTermChunk + 1];  // exclusive prefix of run lengths
    __shared__ float run_w[kScanTermChunk];
    const int64_t range = blockIdx.x;
    const int qi = blockIdx.y;
    const int tid = threadIdx.x;
    for (int i = tid; i < kRangeDocs; i += 256) acc[i] = 0.f;
    const unsigned int* offs = rt_off + range * V1;
    const int64_t base = range_base[range];
    const int64_t t0 = q_indptr[qi], t1 = q_indptr[qi + 1];

    for (int64_t tc = t0; tc < t1; tc += kScanTermChunk) {
        const int nt = (int)((t1 - tc) < kScanTermChunk ? (t1 - tc) : kScanTermChunk);
        __syncthreads();
        unsigned int len = 0;
        if (tid < nt) {
            int32_t t = q_idx[tc + tid];
            unsigned int lo = offs[t], hi = offs[t + 1];
            run_lo[tid] = lo;
            run_w[tid] = q_val[tc + tid];
            len = hi - lo;
        }
        // block exclusive scan of len over 256 threads
        {
            __shared__ unsigned int wsum[4];
            const int lane = tid & 63, w = tid >> 6;
            unsigned int x = len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                unsigned int y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            if (lane == 63) wsum[w] = x;
            __syncthreads();
            unsigned int wbase = 0;
            for (int j = 0; j < w; ++j) wbase += wsum[j];
            run_pre[tid] = wbase + x - len;
            if (tid == 255) run_pre[256] = wbase + x;
        }
        __syncthreads();
        const unsigned int total = run_pre[256];
        for (unsigned int p = tid; p < total; p += 256) {
            // largest i with run_pre[i] <= p
            int lo = 0, hi = nt - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (run_pre[mid] <= p) lo = mid; else hi = mid - 1;
            }
            const int64_t e = base + run_lo[lo] + (p - run_pre[lo]);
            atomicAdd(&acc[post_doc[e]], run_w[lo] * post_val[e]);
        }
    }
    __syncthreads();
    // per-group maxima: thread = (group, sub) with 4 threads per group
    {
        const int grp = tid >> 2, sub = tid & 3;
        float m = 0.f;
        const int64_t doc0 = range * kRangeDocs + grp * kGroupRows;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            int local = j * 4 + sub;
            float v = acc[grp * kGroupRows + local];
            if (rowmask) {
                int64_t d = doc0 + local;
                if (d < n_docs && !((rowmask[d >> 3] >> (d & 7)) & 1)) v = 0.f;
            }
            m = fmaxf(m, v);
        }
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        const int64_t group = range * kRangeGroups + grp;
        if (sub == 0 && group < n_groups) gmax[(int64_t)qi * n_groups + group] = m;
    }
}

// ---- refine: one wave per (query, candidate group); lane = doc -------------------
// Canonical score: walk the doc's CSR entries in stored order, look each index
// up in the query's sorted terms, accumulate exact products in fp64.
// Restated in oracle/oracle.c:sparse_score().
__global__ __launch_bounds__(64) void refine_sparse_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
    const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
    const float* __restrict__ q_val, const uint8_t* __restrict__ rowmask,
    const int32_t* __restrict__ cand, int C, int64_t n_docs, float* __restrict__ out_score,
    int32_t* __restrict__ out_row) {
    const int qi = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    const int32_t group = cand[(int64_t)qi * C + ci];
    const int64_t o = ((int64_t)qi * C + ci) * kGroupRows + lane;
    const int64_t doc = (int64_t)group * kGroupRows + lane;
    bool valid = group >= 0 && doc < n_docs;
    if (valid && rowmask) valid = (rowmask[doc >> 3] >> (doc & 7)) & 1;
    float score = 0.f;
    if (valid) {
        const int64_t t0 = q_indptr[qi];
        const int nt = (int)(q_indptr[qi + 1] - t0);
        const int32_t* qi_idx = q_idx + t0;
        const float* qi_val = q_val + t0;
        double s = 0.0;
        for (int64_t e = indptr[doc]; e < indptr[doc + 1]; ++e) {
            const int32_t t = idx[e];
            int lo = 0, hi = nt;  // first position with qi_idx[pos] >= t
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (qi_idx[mid] < t) lo = mid + 1; else hi = mid;
            }
            if (lo < nt && qi_idx[lo] == t) s = __dadd_rn(s, __dmul_rn((double)val[e], (double)qi_val[lo]));
        }
        score = (float)s;
    }
    const bool keep = valid && score > 0.f;
    out_score[o] = keep ? score : -__builtin_inff();
    out_row[o] = keep ? (int32_t)doc : -1;
}

}  // namespace hbmrag
