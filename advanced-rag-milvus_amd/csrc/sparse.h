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

constexpr int kRangeDocs = 16384;                     // docs per range = LDS accumulator length (u16 local ids)
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

// A posting is 4 bytes: fp16 weight (high half) | uint16 doc id local to the range.
// The scan is only the candidate generator (the refine recomputes from the fp32
// CSR), so the weight may be rounded; a positive weight never rounds to zero, so
// "accumulator > 0  <=>  some positive product" still holds.
__device__ inline uint32_t pack_posting(uint16_t local_doc, float w) {
    union { _Float16 h; unsigned short u; } cv;
    cv.h = (_Float16)w;  // round to nearest even
    unsigned short hb = cv.u;
    if ((hb & 0x7FFFu) == 0 && w != 0.f) hb = (unsigned short)((w < 0.f ? 0x8000u : 0u) | 1u);  // keep the sign, min subnormal
    return ((uint32_t)hb << 16) | local_doc;
}
__device__ inline float posting_weight(uint32_t p) {
    union { _Float16 h; unsigned short u; } cv;
    cv.u = (unsigned short)(p >> 16);
    return (float)cv.h;
}

// Scatter postings.  cursor = copy of rt_off; order inside a run follows the
// atomics (the scan's sums are order-independent up to fp32 rounding, which
// the refine step makes irrelevant).
__global__ void sparse_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                   const float* __restrict__ val, int64_t n_docs, int64_t V1,
                                   unsigned int* __restrict__ cursor, const int64_t* __restrict__ range_base,
                                   uint32_t* __restrict__ post) {
    int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    const int64_t range = d / kRangeDocs;
    unsigned int* cur = cursor + range * V1;
    const int64_t base = range_base[range];
    const uint16_t local = (uint16_t)(d - range * kRangeDocs);
    for (int64_t e = indptr[d]; e < indptr[d + 1]; ++e) {
        unsigned int slot = atomicAdd(&cur[idx[e]], 1u);
        post[base + slot] = pack_posting(local, val[e]);
    }
}

// ---- query prep: fixed-point scale per query -------------------------------------
// The scan accumulates in 32-bit FIXED POINT (LDS integer atomics run ~4x the
// rate of LDS float atomics on gfx950, and integer sums do not depend on the
// order the waves arrive in).  With S = sum |w_q| and M = max |doc weight| every
// partial sum is bounded by S*M, so scale = 2^30/(S*M) cannot overflow int32.
// Each contribution is rounded AWAY from zero (a positive product always counts
// at least 1), hence |fixed/scale - exact| <= (nnz+1)/scale: that bound is
// written to q_eps for the exactness check of select_topk.
// It also re-lays the CSR queries out at a fixed stride (pq_idx / pq_w = weight*scale, zero padded,
// pq_n = term count), so a scan block can fetch its query's terms without first waiting for q_indptr:
// one dependent round trip less per block.
__global__ __launch_bounds__(256) void sparse_query_prep_kernel(const int64_t* __restrict__ q_indptr,
                                                                const int32_t* __restrict__ q_idx,
                                                                const float* __restrict__ q_val, float max_doc_w,
                                                                int stride, float* __restrict__ q_scale,
                                                                float* __restrict__ q_eps, int32_t* __restrict__ pq_n,
                                                                int32_t* __restrict__ pq_idx,
                                                                float* __restrict__ pq_w) {
    // q_eps = absolute part of the scan's error bound: fixed-point rounding (nnz+1)/scale plus the
    // fp16 floor of tiny doc weights (6e-8 per unit of query weight); the relative part (fp16
    // rounding of normal weights, 2^-11) is passed to select_topk as eps_rel.
    __shared__ float part[256];
    const int qi = blockIdx.x, tid = threadIdx.x;
    const int64_t t0 = q_indptr[qi], t1 = q_indptr[qi + 1];
    float s = 0.f;
    for (int64_t i = t0 + tid; i < t1; i += 256) s += fabsf(q_val[i]);
    part[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {  // fixed tree: the same value in every launch
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    const float bound = part[0] * max_doc_w;
    const float scale = bound > 0.f ? 1073741824.0f / bound : 0.f;
    if (tid == 0) {
        q_scale[qi] = scale;
        // a query longer than the caller's max_q_nnz would be truncated by the fixed-stride layout: make its
        // list "never proven" so it is redone through the host form
        q_eps[qi] = (t1 - t0 > stride) ? __builtin_inff()
                                       : (scale > 0.f ? (float)(t1 - t0 + 1) / scale + part[0] * 6.0e-8f : 0.f);
        pq_n[qi] = (int32_t)min((int64_t)stride, t1 - t0);
    }
    for (int i = tid; i < stride; i += 256) {
        const bool in = t0 + i < t1;
        pq_idx[(int64_t)qi * stride + i] = in ? q_idx[t0 + i] : 0;
        pq_w[(int64_t)qi * stride + i] = in ? q_val[t0 + i] * scale : 0.f;
    }
}

// ---- scan: grid (n_ranges, B), 1024 threads -------------------------------------
// The block owns docs [range*16384, +16384) of one query: accumulators live in
// LDS (64 KiB; two blocks per CU), every posting of the query's terms inside
// the range is applied with an LDS integer atomic, and only the per-64-doc
// maxima leave the CU.  Work is cut into items of kItemPostings consecutive
// postings of one run; each wave takes items in a strided loop, kItemsInFlight
// at a time, so it keeps 16 independent coalesced loads in flight instead of one
// (a typical (query, range) — 80 runs of ~164 postings = ~240 items — is then ONE round trip).
// Algorithmic HBM bytes per (query, range): sum over query terms of run_len * 4
// (packed uint16 doc + fp16 weight) + 2*4 per term for the run bounds + group maxima out.
constexpr int kScanThreads = 1024;
constexpr int kDocsPerThread = kRangeDocs / kScanThreads;  // consecutive docs one thread reduces in the group-max pass (16 or 32)
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kItemPostings = 64;
constexpr int kItemsInFlight = 16;
constexpr int kItemTable = 4096;  // items whose run is looked up from a table instead of searched

__device__ inline int fixed_contrib(float p) {  // round away from zero, branch-free
    const int a = __float2int_ru(fabsf(p));
    return p < 0.f ? -a : a;
}
// Accumulator index with one pad word per 16 docs: the per-group maxima (one
// thread per 16 docs, stride 17 words) then read conflict-free.
__device__ inline int acc_index(int d) { return d + (d >> 4); }

#ifdef HR_TRACE
__device__ unsigned long long hr_trace[16];
#define TR(i) do { if (tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&hr_trace[i], t_ - tr_last); tr_last = t_; } } while (0)
#define TRWAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define TR(i)
#define TRWAIT()
#endif
__global__ __launch_bounds__(kScanThreads) void sparse_scan_kernel(
    const unsigned int* __restrict__ rt_off, int64_t V1, const int64_t* __restrict__ range_base,
    const uint32_t* __restrict__ post,
    const int32_t* __restrict__ pq_n, const int32_t* __restrict__ pq_idx, const float* __restrict__ pq_w,
    int stride, const float* __restrict__ q_scale, const uint8_t* __restrict__ rowmask,
    int64_t n_docs, int64_t n_groups, int group_docs, float* __restrict__ gmax) {
    __shared__ int acc[kRangeDocs + kRangeDocs / 16];
    __shared__ uint8_t item_run[kItemTable];
    __shared__ unsigned int run_lo[kScanTermChunk], run_hi[kScanTermChunk];
    __shared__ unsigned int item_pre[kScanTermChunk + 1];  // exclusive prefix of items per run
    __shared__ float run_w[kScanTermChunk];
    __shared__ unsigned int wsum[16];
    const int64_t range = blockIdx.x;
    const int qi = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef HR_TRACE
    unsigned long long tr_last = __builtin_amdgcn_s_memtime();
    if (tid == 0) atomicAdd(&hr_trace[15], 1ull);
#endif
    {
        int4* a4 = reinterpret_cast<int4*>(acc);
        for (int i = tid; i < (kRangeDocs + kRangeDocs / 16) / 4; i += kScanThreads) a4[i] = make_int4(0, 0, 0, 0);
    }
    TRWAIT(); TR(0);
    const unsigned int* offs = rt_off + range * V1;
    const uint32_t* pp = post + range_base[range];
    // every load below is independent of the others except run bounds <- term: two round trips, not three
    const int n_terms = pq_n[qi];
    const float scale = q_scale[qi];
    const int32_t* my_idx = pq_idx + (int64_t)qi * stride;
    const float* my_w = pq_w + (int64_t)qi * stride;

    for (int tc = 0; tc == 0 || tc < n_terms; tc += kScanTermChunk) {
        // speculative fetch: slots past the query's length hold term 0 / weight 0 (padding written by the prep)
        const bool slot = tid < kScanTermChunk && tc + tid < stride;
        const int32_t t = slot ? my_idx[tc + tid] : 0;
        const float wq = slot ? my_w[tc + tid] : 0.f;
        TRWAIT(); TR(1);
        const unsigned int lo = slot ? offs[t] : 0u, hi = slot ? offs[t + 1] : 0u;
        const int nt = min(kScanTermChunk, max(n_terms - tc, 0));
        TRWAIT(); TR(2);
        if (tc) __syncthreads();  // the run tables of the previous chunk are still being read
        const unsigned int items = tid < nt ? (hi - lo + kItemPostings - 1) / kItemPostings : 0u;
        // exclusive scan of `items` over the first 256 threads (4 waves)
        unsigned int x = items;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        if (tid < kScanTermChunk) {
            unsigned int wbase = 0;
            for (int j = 0; j < wave; ++j) wbase += wsum[j];
            const unsigned int first = wbase + x - items;
            item_pre[tid] = first;
            if (tid == kScanTermChunk - 1) item_pre[kScanTermChunk] = wbase + x;
            if (tid < nt) {
                run_lo[tid] = lo;
                run_hi[tid] = hi;
                run_w[tid] = wq;
                // run of each of the first kItemTable items
                for (unsigned int c = 0; c < items && first + c < kItemTable; ++c) item_run[first + c] = (uint8_t)tid;
            }
        }
        __syncthreads();
        const unsigned int total_items = item_pre[kScanTermChunk];
        TR(3);

        for (unsigned int base = wave; base < total_items; base += kScanWaves * kItemsInFlight) {
            uint32_t pk[kItemsInFlight];
            int rr[kItemsInFlight];
#pragma unroll
            for (int u = 0; u < kItemsInFlight; ++u) {  // every load is issued before the first use
                const unsigned int item = base + kScanWaves * u;
                rr[u] = 0;
                pk[u] = 0u;
                if (item < total_items) {
                    int r;
                    if (item < kItemTable) {
                        r = item_run[item];
                    } else {  // largest r with item_pre[r] <= item (wave-uniform)
                        r = 0;
                        int top = nt - 1;
                        while (r < top) {
                            const int mid = (r + top + 1) >> 1;
                            if (item_pre[mid] <= item) r = mid; else top = mid - 1;
                        }
                    }
                    const unsigned int e = run_lo[r] + (item - item_pre[r]) * kItemPostings + lane;
                    rr[u] = r;
                    if (e < run_hi[r]) pk[u] = pp[e];
                }
            }
            TRWAIT(); TR(4);
#pragma unroll
            for (int u = 0; u < kItemsInFlight; ++u)  // weight bits 0 = no posting (or a zero weight): nothing to add
                if (pk[u] >> 16) atomicAdd(&acc[acc_index((int)(pk[u] & 0xFFFFu))], fixed_contrib(run_w[rr[u]] * posting_weight(pk[u])));
        }
    }
    TRWAIT(); TR(5);
    __syncthreads();
    TR(6);
    // per-group maxima: every thread reduces 16 consecutive docs; 64-doc groups
    // finish with a 4-lane reduction
    {
        int m[kDocsPerThread / 16];
        const int64_t doc0 = range * kRangeDocs + (int64_t)tid * kDocsPerThread;
#pragma unroll
        for (int c = 0; c < kDocsPerThread / 16; ++c) {
            m[c] = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                int v = acc[acc_index(tid * kDocsPerThread + c * 16 + j)];
                if (rowmask) {
                    const int64_t dd = doc0 + c * 16 + j;
                    if (dd < n_docs && !((rowmask[dd >> 3] >> (dd & 7)) & 1)) v = 0;
                }
                m[c] = max(m[c], v);
            }
        }
        const float inv = scale > 0.f ? 1.0f / scale : 0.f;
        if (group_docs == 16) {
#pragma unroll
            for (int c = 0; c < kDocsPerThread / 16; ++c) {
                const int64_t group = range * (kRangeDocs / 16) + tid * (kDocsPerThread / 16) + c;
                if (group < n_groups) gmax[(int64_t)qi * n_groups + group] = (float)m[c] * inv;
            }
        } else {
            int mm = m[0];
#pragma unroll
            for (int c = 1; c < kDocsPerThread / 16; ++c) mm = max(mm, m[c]);
            constexpr int kLanesPerGroup = 64 / kDocsPerThread;  // 4 or 2
#pragma unroll
            for (int off = 1; off < kLanesPerGroup; off <<= 1) mm = max(mm, __shfl_xor(mm, off));
            const int64_t group = range * (kRangeDocs / 64) + tid / kLanesPerGroup;
            if (tid % kLanesPerGroup == 0 && group < n_groups) gmax[(int64_t)qi * n_groups + group] = (float)mm * inv;
        }
    }
    TRWAIT(); TR(7);
}

// ---- refine: one WAVE per candidate doc at a time (group_docs = 16 or 64 docs per group) ----
// Canonical score: walk the doc's CSR entries in stored order, look each index
// up in the query's sorted terms, accumulate exact products in fp64.
// Restated in oracle/oracle.c:oracle_sparse_scores().
// The block (4 waves x 64 candidate docs of one query) stages the query in LDS
// together with a 32768-bit hashed membership filter, so the ~99 % of entries
// that cannot match cost one LDS read instead of a binary search.
// Each wave walks its 64 docs one after the other with lane = entry: the doc's indices
// and values arrive as two coalesced loads per 64 entries (the entries of the next
// kRefinePrefetch docs are already in flight), every lane tests its own entry, and the
// few matching products are added in ENTRY ORDER (ballot, lowest lane first), which is
// the canonical order.  (One thread per doc, the first form of this kernel, read each
// row with a 400-byte stride between lanes and thrashed the L1: 0.97 ms per 128-query
// batch against the scans it shares the chip with.)
constexpr int kFilterBits = 32768;
constexpr int kRefinePrefetch = 2;

__global__ __launch_bounds__(256) void refine_sparse_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
    const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
    const float* __restrict__ q_val, const uint8_t* __restrict__ rowmask,
    const int32_t* __restrict__ cand, int C, int group_docs, int64_t n_docs, int q_cap,
    float* __restrict__ out_score, int32_t* __restrict__ out_row) {
    // dynamic LDS: filter words, then q_cap query indices and q_cap query values (q_cap = the batch's
    // longest query rounded up, so short queries leave the CU room for many blocks)
    extern __shared__ unsigned int refine_lds[];
    unsigned int* s_filter = refine_lds;
    int32_t* s_idx = reinterpret_cast<int32_t*>(refine_lds + kFilterBits / 32);
    float* s_val = reinterpret_cast<float*>(s_idx + q_cap);
    const int qi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int64_t t0 = q_indptr[qi];
    // a query longer than q_cap was already marked "never proven" by the prep kernel (q_eps = inf)
    const int nt = min((int)(q_indptr[qi + 1] - t0), q_cap);
    for (int i = tid; i < kFilterBits / 32; i += 256) s_filter[i] = 0u;
    __syncthreads();
    for (int i = tid; i < nt; i += 256) {
        const int32_t t = q_idx[t0 + i];
        s_idx[i] = t;
        s_val[i] = q_val[t0 + i];
        atomicOr(&s_filter[(t & (kFilterBits - 1)) >> 5], 1u << (t & 31));
    }
    __syncthreads();
    const int n_slots = C * group_docs;
    const int slot0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 256 + (tid & ~63));  // first doc slot of this wave
    if (slot0 >= n_slots) return;
    const int n_here = min(64, n_slots - slot0);
    // lane l describes doc slot0 + l
    const int slot = slot0 + lane;
    int64_t doc = -1, p0 = 0;
    int len = 0;
    bool valid = false;
    if (lane < n_here) {
        const int32_t group = cand[(int64_t)qi * C + slot / group_docs];
        doc = (int64_t)group * group_docs + slot % group_docs;
        valid = group >= 0 && doc < n_docs;
        if (valid && rowmask) valid = (rowmask[doc >> 3] >> (doc & 7)) & 1;
        if (valid) {
            p0 = indptr[doc];
            len = (int)(indptr[doc + 1] - p0);
        }
    }
    const unsigned int p0_lo = (unsigned int)p0, p0_hi = (unsigned int)((unsigned long long)p0 >> 32);
    auto doc_start = [&](int j) -> int64_t {
        return (int64_t)(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)p0_hi, j) << 32) |
                         (unsigned int)__builtin_amdgcn_readlane((int)p0_lo, j));
    };
    int32_t ri[kRefinePrefetch][2];
    float rv[kRefinePrefetch][2];
    auto fetch = [&](int j, int32_t (&ti)[2], float (&tv)[2]) {
        ti[0] = ti[1] = -1;
        tv[0] = tv[1] = 0.f;
        if (j < n_here) {
            const int n = __builtin_amdgcn_readlane(len, j);
            const int64_t st = doc_start(j);
            if (lane < n) {
                ti[0] = idx[st + lane];
                tv[0] = val[st + lane];
            }
            if (lane + 64 < n) {
                ti[1] = idx[st + 64 + lane];
                tv[1] = val[st + 64 + lane];
            }
        }
    };
    double s = 0.0;
    auto consume = [&](int32_t tt, float vv) {  // one entry per lane; adds the matches in lane (= entry) order
        bool hit = false;
        double prod = 0.0;
        if (tt >= 0 && ((s_filter[(tt & (kFilterBits - 1)) >> 5] >> (tt & 31)) & 1u)) {
            int lo = 0, hi = nt;  // first position with s_idx[pos] >= tt
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_idx[mid] < tt) lo = mid + 1; else hi = mid;
            }
            if (lo < nt && s_idx[lo] == tt) {
                hit = true;
                prod = __dmul_rn((double)vv, (double)s_val[lo]);
            }
        }
        unsigned long long m = __ballot(hit);
        const unsigned long long pb = (unsigned long long)__double_as_longlong(prod);
        const int pl = (int)(unsigned int)pb, ph = (int)(unsigned int)(pb >> 32);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const unsigned long long bits = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(ph, l) << 32) |
                                            (unsigned int)__builtin_amdgcn_readlane(pl, l);
            s = __dadd_rn(s, __longlong_as_double((long long)bits));
        }
    };
#pragma unroll
    for (int u = 0; u < kRefinePrefetch; ++u) fetch(u, ri[u], rv[u]);
    float my_score = 0.f;
    for (int jb = 0; jb < n_here; jb += kRefinePrefetch) {
#pragma unroll
        for (int u = 0; u < kRefinePrefetch; ++u) {
            const int j = jb + u;
            const int32_t i0 = ri[u][0], i1 = ri[u][1];
            const float v0 = rv[u][0], v1 = rv[u][1];
            fetch(j + kRefinePrefetch, ri[u], rv[u]);
            if (j < n_here) {
                const int n = __builtin_amdgcn_readlane(len, j);
                s = 0.0;
                if (n > 0) {
                    consume(i0, v0);
                    if (n > 64) consume(i1, v1);
                    if (n > 128) {  // long docs: the rest, not prefetched
                        const int64_t st = doc_start(j);
                        for (int off = 128; off < n; off += 64) {
                            const bool in = off + lane < n;
                            consume(in ? idx[st + off + lane] : -1, in ? val[st + off + lane] : 0.f);
                        }
                    }
                }
                if (lane == j) my_score = (float)s;
            }
        }
    }
    if (lane < n_here) {
        const int64_t o = (int64_t)qi * n_slots + slot;
        const bool keep = valid && my_score > 0.f;
        out_score[o] = keep ? my_score : -__builtin_inff();
        out_row[o] = keep ? (int32_t)doc : -1;
    }
}

}  // namespace hbmrag
