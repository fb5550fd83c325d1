// The finishing chain of a search as ONE kernel: per (query, modality) block
//   bucket maxima -> candidate groups (select.h) -> canonical refine (dense.h / sparse.h) -> top-k + exactness flag.
// The steps are the block functions the separate kernels are made of, so the lists are bit-identical to the
// five-launch chain (bucket_max, select_groups, refine_dense, refine_sparse, select_topk); what changes is where the
// intermediates live — bucket maxima, candidate groups and the refined keys stay in LDS — and that the stream which
// finishes a query batch carries one launch instead of five: on a rank-sized shard the chain's launches, each
// starved by the scans it runs beside, were as long as the scans themselves.
// Used when a batch has enough queries to fill the chip with one block per (query, modality) and the candidate set
// fits LDS; small batches (the single-query latency path) and escalated searches keep the multi-launch chain, whose
// refine kernels spread one query over many compute units.
#pragma once
#include "common.h"
#include "dense.h"
#include "select.h"
#include "sparse.h"

namespace hbmrag {

constexpr int kFinishThreads = 512;        // 8 waves: fits beside a resident scan block (<= 80 VGPRs per lane)
constexpr int kFinishMaxSlots = 4096;      // candidate rows per query whose keys fit LDS (32 KiB)
constexpr int kFinishMaxCand = 256;        // candidate groups (kFinishMaxSlots / 16)
constexpr int kFinishMaxBuckets = 12288;   // bucket maxima kept in LDS (48 KiB)

struct FinishMod {
    int kind;                  // 0 = dense, 1 = sparse
    int group_rows;
    GroupSelArgs sel;          // sel.bmax / sel.cand unused: both live in LDS
    TopkArgs topk;             // topk.cscore / crow unused: the keys live in LDS
    const uint8_t* rowmask;
    // dense
    const chunk_t* tiles;
    int KT, dim, cosine, dtype;
    const float* q;
    const double* qn2;
    const double* norm2;
    // sparse
    const int64_t* indptr;
    const int32_t* idx;
    const float* val;
    const int64_t* q_indptr;
    const int32_t* q_idx;
    const float* q_val;
    int q_cap;
    int64_t n_rows;            // rows (dense) or docs (sparse) of the shard
};
struct FinishPair {
    FinishMod m[2];
    int n;
    int key_slots;             // max over the modalities of C * group_rows: size of the key region of the dynamic LDS
};

// LDS the kernel needs beyond its static arrays: the keys of every candidate row, then a region that first holds the
// query's bucket maxima and later (sparse) the staged query and its membership filter.
__host__ __device__ inline bool finish_sparse_hashed(int q_cap) { return q_cap <= kHashMaxTerms; }
inline size_t finish_lds_bytes(const FinishPair& p) {
    size_t region_b = 0;
    for (int i = 0; i < p.n; ++i) {
        const FinishMod& m = p.m[i];
        if (m.sel.two_level) region_b = region_b > (size_t)m.sel.n_buckets * 4 ? region_b : (size_t)m.sel.n_buckets * 4;
        if (m.kind == 1) {
            const size_t s = (size_t)kFilterBits / 8 +
                             (finish_sparse_hashed(m.q_cap) ? (size_t)sparse_hash_slots(m.q_cap) * 8 : (size_t)m.q_cap * 8);
            region_b = region_b > s ? region_b : s;
        } else {
            const size_t s = (size_t)m.dim * 4;   // the query, staged for the refine
            region_b = region_b > s ? region_b : s;
        }
    }
    return (size_t)p.key_slots * 8 + region_b;
}

// Bucket maxima of one query into LDS (what bucket_max_kernel writes to the global table).  A wave takes
// kFinishBucketsPerWave buckets per trip and issues all their (coalesced, 256-byte) loads before the first reduction:
// with 8 waves per block the pass is bound by memory latency, not bytes (312 KB per query at 1.25M rows).
// (64 buckets per trip — 16 KiB in flight per wave — leave the phase as long as it is alone and cost the step 4 % in the
// pipeline: 88 instead of 80 registers per lane beside the scan's waves; profiles/r4_experiments/finish_variants_ab.txt)
constexpr int kFinishBucketsPerWave = 32;
__device__ inline void bucket_max_block(const GroupSelArgs& a, int q, float* s_bmax) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const float* gm = a.gmax + (int64_t)q * a.n_groups;
    for (int64_t b0 = (int64_t)wid * kFinishBucketsPerWave; b0 < a.n_buckets; b0 += (int64_t)nw * kFinishBucketsPerWave) {
        float v[kFinishBucketsPerWave];
#pragma unroll
        for (int i = 0; i < kFinishBucketsPerWave; ++i) {
            const int64_t g = (b0 + i) * kBucketGroups + lane;
            v[i] = g < a.n_groups ? gm[g] : -__builtin_inff();
        }
        float mine = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < kFinishBucketsPerWave; ++i) {
            const float m = wave_max_dpp(v[i]);
            if (lane == i) mine = m;
        }
        if (lane < kFinishBucketsPerWave && b0 + lane < a.n_buckets) s_bmax[b0 + lane] = mine;
    }
}

// <= 88 VGPRs: two of these waves fit on a SIMD beside the two 168-register waves of a resident dense-scan block
#ifdef HR_STAMP  // diagnostic build only (make stamp): s_memtime at the phase boundaries of block (query 0, modality m)
__device__ unsigned long long hr_finish_stamps[2][8];
#define FINISH_STAMP(i)                                                                      \
    do {                                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) hr_finish_stamps[blockIdx.y][i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FINISH_STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(kFinishThreads) void finish_kernel(FinishPair p) {
    extern __shared__ uint64_t finish_lds[];
    __shared__ SelectScratch sh;
    __shared__ int32_t sel_bucket[HR_MAX_TOPK * 2];
    __shared__ uint64_t sel[HR_MAX_TOPK];
    __shared__ int32_t s_cand[kFinishMaxCand];
    const FinishMod& a = p.m[blockIdx.y];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n_slots = a.sel.C * a.group_rows;
    uint64_t* s_key = finish_lds;
    float* s_bmax = reinterpret_cast<float*>(finish_lds + p.key_slots);

    FINISH_STAMP(0);
    if (a.sel.two_level) bucket_max_block(a.sel, q, s_bmax);
    __syncthreads();
    FINISH_STAMP(1);
    select_groups_block(a.sel, q, s_bmax, s_cand, sh, sel_bucket);
    __syncthreads();
    // the candidates that survived the trim sit at the front (GroupSelArgs::K_trim): only their rows are refined
    const int n_live_slots = min(sh.n_live, a.sel.C) * a.group_rows;
    for (int slot = n_live_slots + tid; slot < n_slots; slot += kFinishThreads) s_key[slot] = 0ull;
    FINISH_STAMP(2);
#ifdef HR_STAMP
    if (blockIdx.x == 0 && tid == 0) {
        hr_finish_stamps[blockIdx.y][5] = (unsigned long long)n_live_slots;
        hr_finish_stamps[blockIdx.y][6] = 0ull;
    }
#endif

    if (a.kind == 0) {
        // the query in LDS (the bucket maxima are no longer needed): 64 query values per round as LDS reads instead of
        // global loads that would compete with the row chunks for registers and memory queue
        float* qq = s_bmax;
        for (int k = tid; k < a.dim; k += kFinishThreads) qq[k] = a.q[(int64_t)q * a.dim + k];
        __syncthreads();
        const double qn2 = a.qn2[q];
        for (int slot = tid; slot < n_live_slots; slot += kFinishThreads) {
#ifdef HR_STAMP
            if (blockIdx.x == 0) atomicAdd(&hr_finish_stamps[blockIdx.y][6], 1ull);   // rows that take the canonical chain
#endif
            float sc = 0.f;
            int32_t row = -1;
            const bool ok = a.dtype == HR_F16
                                ? refine_dense_slot<_Float16>(a.tiles, a.KT, a.dim, qq, qn2, a.norm2, a.rowmask, s_cand,
                                                              a.group_rows, a.n_rows, a.cosine, slot, &sc, &row)
                                : refine_dense_slot<float>(a.tiles, a.KT, a.dim, qq, qn2, a.norm2, a.rowmask, s_cand,
                                                           a.group_rows, a.n_rows, a.cosine, slot, &sc, &row);
            s_key[slot] = ok ? rank_key(sc, (uint32_t)row) : 0ull;
        }
    } else {
        // the docs in chains of equal length, a multiple of 8 of them: every wave walks the same number of docs (chains of
        // 64 left half of the waves with twice the docs of the others whenever the trim kept 9 .. 15 x 64 of them)
        constexpr int NW = kFinishThreads / 64;
        const int n_chains = NW * ((n_live_slots + NW * 64 - 1) / (NW * 64));
        const int chain_len = (n_live_slots + n_chains - 1) / n_chains;
        const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto emit = [&](int slot, bool keep, float score, int64_t doc) { s_key[slot] = keep ? rank_key(score, (uint32_t)doc) : 0ull; };
        if (finish_sparse_hashed(a.q_cap)) {
            const int hs = sparse_hash_slots(a.q_cap);
            unsigned int* s_filter = reinterpret_cast<unsigned int*>(s_bmax);   // the bucket maxima are no longer needed
            int32_t* s_hkey = reinterpret_cast<int32_t*>(s_filter + kFilterBits / 32);
            float* s_hval = reinterpret_cast<float*>(s_hkey + hs);
            refine_sparse_stage_query_hash(a.q_indptr, a.q_idx, a.q_val, q, a.q_cap, hs, s_filter, s_hkey, s_hval);
            for (int chain = wid; chain < n_chains; chain += NW) {
                const int slot0 = chain * chain_len;
                const int n_here = min(chain_len, n_live_slots - slot0);
                if (n_here > 0)
                    refine_sparse_chain(a.indptr, a.idx, a.val, a.rowmask, s_cand, a.group_rows, a.n_rows, slot0, n_here,
                                        SparseLookupHash{s_filter, s_hkey, s_hval, hs}, emit);
            }
        } else {
            unsigned int* s_filter = reinterpret_cast<unsigned int*>(s_bmax);
            int32_t* s_idx = reinterpret_cast<int32_t*>(s_filter + kFilterBits / 32);
            float* s_val = reinterpret_cast<float*>(s_idx + a.q_cap);
            const int nt = refine_sparse_stage_query(a.q_indptr, a.q_idx, a.q_val, q, a.q_cap, s_filter, s_idx, s_val);
            for (int chain = wid; chain < n_chains; chain += NW) {
                const int slot0 = chain * chain_len;
                const int n_here = min(chain_len, n_live_slots - slot0);
                if (n_here > 0)
                    refine_sparse_chain(a.indptr, a.idx, a.val, a.rowmask, s_cand, a.group_rows, a.n_rows, slot0, n_here,
                                        SparseLookupSorted{s_filter, s_idx, s_val, nt}, emit);
            }
        }
    }
    __syncthreads();
    FINISH_STAMP(3);
    TopkArgs t = a.topk;
    t.n = n_slots;
    select_topk_block(t, q, [&](int64_t i) -> uint64_t { return s_key[i]; }, sh, sel);
    __syncthreads();
    FINISH_STAMP(4);
}

}  // namespace hbmrag
