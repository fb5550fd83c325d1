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
// A term that at least half of a range's docs have is stored as a DENSE run: one fp16 weight per doc of the range, at
// dense_run_pos(doc) (kRangeDocs / 2 posting words; 0x8000 = the doc does not have the term), instead of a 4-byte posting per doc that
// has it — never longer than the sparse form, and the scan applies it without LDS atomics and without a slot table (the
// thread that owns 32 consecutive docs adds the 32 products to its own accumulators).  On Zipfian postings these few runs are
// most of a query's postings.  A run is dense iff its length is exactly kDenseRunWords: sparse runs are shorter.
constexpr unsigned kDenseRunWords = kRangeDocs / 2;
constexpr unsigned short kDenseAbsent = 0x8000u;      // -0.0: a stored weight of +/-0 is canonicalised to +0
// Where local doc d sits inside a dense run (in fp16 units).  The scan thread that owns docs 32 t .. 32 t + 31 applies a run
// as four 16-byte pieces (8 docs each); piece q of ALL 512 threads is stored contiguously, so a wave's load of its piece q
// is one fully coalesced 1 KiB (in doc order a lane's piece sat 64 bytes from its neighbour's: sixteen 64-byte sectors
// touched per 256 bytes used, four times over for the four pieces).
__host__ __device__ inline unsigned dense_run_pos(unsigned local_doc) {
    const unsigned t = local_doc >> 5, i = local_doc & 31u;
    return (((i >> 3) * (unsigned)(kRangeDocs / 32) + t) << 3) | (i & 7u);
}

// ---- build: CSR (doc-major) -> range-major postings ---------------------------
// rt_off[range][t] counts, then (after the per-range exclusive scan) offsets of
// term t's run inside the range's posting block.
// All three build kernels work on docs [doc0, n_docs): an append rebuilds only the ranges from the one that holds
// the first new doc onwards (doc0 = that range's first doc).
__global__ void sparse_count_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                    int64_t doc0, int64_t n_docs, int64_t V1, unsigned int* __restrict__ rt_off) {
    int64_t d = doc0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_docs) return;
    unsigned int* row = rt_off + (d / kRangeDocs) * V1;
    for (int64_t e = indptr[d]; e < indptr[d + 1]; ++e) atomicAdd(&row[idx[e]], 1u);
}

// One block per range: in-place exclusive scan of the V run lengths, each rounded up to a multiple of
// 4 postings (the scan fetches 16 bytes per lane and applies all four without a validity test; the
// round-up slots hold filler postings, see sparse_pad_kernel); slot V gets the total.
__global__ __launch_bounds__(1024) void sparse_scan_offsets_kernel(unsigned int* __restrict__ rt_off, int64_t V1,
                                                                   int64_t range0,
                                                                   unsigned long long* __restrict__ range_total) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry, n_dense;
    unsigned int* row = rt_off + (range0 + (int64_t)blockIdx.x) * V1;
    const int64_t V = V1 - 1;
    if (threadIdx.x == 0) { carry = 0; n_dense = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = 0; base < V; base += blockDim.x) {
        int64_t i = base + threadIdx.x;
        unsigned int v = (i < V) ? (row[i] + 3u) & ~3u : 0u;
        if (v >= kDenseRunWords) {   // half of the range's docs or more: the dense form
            v = kDenseRunWords;
            atomicAdd(&n_dense, 1u);
        }
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
        range_total[blockIdx.x] = (unsigned long long)carry | ((unsigned long long)n_dense << 32);   // words | dense runs
    }
}

// A posting is 4 bytes: fp16 weight (high half) | uint16 accumulator slot of the doc inside its range.
// The scan is only the candidate generator (the refine recomputes from the fp32
// CSR), so the weight may be rounded; a positive weight never rounds to zero, so
// "accumulator > 0  <=>  some positive product" still holds.
// Accumulator slot of local doc d = d + 2 * (d / 32): the scan's LDS accumulators carry two pad words per
// 32 docs (a thread's slice of 32 docs then starts 8-byte aligned and a half-wave's slices fall on distinct
// banks), and storing the padded index here saves the scan two instructions per posting.
constexpr int kAccPadShift = 5;
constexpr int kAccSlice = (1 << kAccPadShift) + 2;                 // words per 32-doc slice
constexpr int kAccWords = (kRangeDocs >> kAccPadShift) * kAccSlice;
__host__ __device__ inline int acc_slot(int local_doc) { return local_doc + 2 * (local_doc >> kAccPadShift); }
__device__ inline unsigned short posting_weight_bits(float w) {
    union { _Float16 h; unsigned short u; } cv;
    cv.h = (_Float16)w;  // round to nearest even
    unsigned short hb = cv.u;
    if ((hb & 0x7FFFu) == 0 && w != 0.f) hb = (unsigned short)((w < 0.f ? 0x8000u : 0u) | 1u);  // keep the sign, min subnormal
    return hb;
}
__device__ inline uint32_t pack_posting(uint16_t local_doc, float w) {
    return ((uint32_t)posting_weight_bits(w) << 16) | (uint32_t)acc_slot(local_doc);
}
__device__ inline float posting_weight(uint32_t p) {
    union { _Float16 h; unsigned short u; } cv;
    cv.u = (unsigned short)(p >> 16);
    return (float)cv.h;
}

// Scatter postings.  cursor = copy of rt_off; order inside a run follows the
// atomics (the scan's sums are order-independent up to fp32 rounding, which
// the refine step makes irrelevant).
// cursor holds the rows of ranges >= range0 only.
// Thread i takes doc perm(i): inside every aligned block of 128 docs, thread 4l + j takes doc 32j + l.  The slots of a
// run are handed out in arrival order, which for the lanes of a wave is lane order, so a DENSE run (a frequent term:
// nearly every doc has it) ends up stored so that postings 4l + j, l = 0..31 — what the 32 lanes of a half-wave apply
// with their j-th LDS atomic — are 32 consecutive docs = 32 distinct banks; stored in doc order they would be docs
// 4 apart, a 4-way bank conflict on every atomic (Zipfian corpora: 0.29 of the HBM peak before, see DESIGN).
__global__ void sparse_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                   const float* __restrict__ val, int64_t doc0, int64_t n_docs, int64_t V1,
                                   unsigned int* __restrict__ cursor, const unsigned int* __restrict__ rt_off,
                                   const int64_t* __restrict__ range_base, uint32_t* __restrict__ post) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t d = doc0 + (i & ~127ll) + 32 * (i & 3) + ((i >> 2) & 31);
    if (d >= n_docs) return;
    const int64_t range = d / kRangeDocs;
    unsigned int* cur = cursor + (range - doc0 / kRangeDocs) * V1;
    const unsigned int* off = rt_off + range * V1;
    const int64_t base = range_base[range];
    const uint16_t local = (uint16_t)(d - range * kRangeDocs);
    for (int64_t e = indptr[d]; e < indptr[d + 1]; ++e) {
        const int32_t t = idx[e];
        const unsigned int lo = off[t];
        if (off[t + 1] - lo == kDenseRunWords) {   // dense run: the doc's weight at its own place (the block was preset to "absent")
            unsigned short hb = posting_weight_bits(val[e]);
            if (hb == kDenseAbsent) hb = 0;         // a stored -0 counts like +0 (one unit, as in the sparse form)
            reinterpret_cast<unsigned short*>(post + base + lo)[dense_run_pos(local)] = hb;
        } else {
            unsigned int slot = atomicAdd(&cur[t], 1u);
            post[base + slot] = pack_posting(local, val[e]);
        }
    }
}

// Filler postings behind the last real posting of every run (up to 3): weight 0, aimed at one of the
// accumulators' pad words, which nobody reads.  cursor = where sparse_fill_kernel stopped.
__device__ __host__ inline uint32_t filler_posting(unsigned int k) { return (uint32_t)(kAccSlice * (k & 511u) + 32u); }
__global__ void sparse_pad_kernel(const unsigned int* __restrict__ rt_off, const unsigned int* __restrict__ cursor,
                                  int64_t V1, int64_t range0, int64_t n_ranges, const int64_t* __restrict__ range_base,
                                  uint32_t* __restrict__ post) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (range - range0, term)
    const int64_t V = V1 - 1;
    if (i >= (n_ranges - range0) * V) return;
    const int64_t rel = i / V, t = i - rel * V, range = range0 + rel;
    const unsigned int end = rt_off[range * V1 + t + 1];
    if (end - rt_off[range * V1 + t] == kDenseRunWords) return;   // a dense run has no filler postings
    for (unsigned int p = cursor[rel * V1 + t]; p < end; ++p) post[range_base[range] + p] = filler_posting((unsigned int)t);
}

// ---- query prep: fixed-point scale per query -------------------------------------
// The scan accumulates in 32-bit FIXED POINT (LDS integer atomics run ~4x the
// rate of LDS float atomics on gfx950, and integer sums do not depend on the
// order the waves arrive in).  With S = sum |w_q| and M = max |doc weight| every
// partial sum is bounded by S*M, so scale = 2^30/(S*M) leaves int32 a factor 2 of headroom.
// A posting contributes trunc(product * scale + 1): never below the exact value, at most 1 above it for
// products >= -1/scale and less than 2 above it otherwise, and a positive product always counts at
// least 1.  Hence 0 <= fixed/scale - exact < 2 * nnz / scale: that bound (with nnz + 1) is written to
// q_eps for the exactness check of select_topk.
// It also re-lays the CSR queries out at a fixed stride (pq_idx / pq_w = weight*scale, zero padded,
// pq_n = term count), so a scan block can fetch its query's terms without first waiting for q_indptr:
// one dependent round trip less per block.
__global__ __launch_bounds__(256) void sparse_query_prep_kernel(const int64_t* __restrict__ q_indptr,
                                                                const int32_t* __restrict__ q_idx,
                                                                const float* __restrict__ q_val, float max_doc_w,
                                                                int stride, int sparse_dim, float* __restrict__ q_scale,
                                                                float* __restrict__ q_eps, int32_t* __restrict__ pq_n,
                                                                int32_t* __restrict__ pq_idx,
                                                                float* __restrict__ pq_w) {
    // q_eps = absolute part of the scan's error bound: fixed-point rounding 2*(nnz+1)/scale plus the
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
                                       : (scale > 0.f ? 2.0f * (float)(t1 - t0 + 1) / scale + part[0] * 6.0e-8f : 0.f);
        pq_n[qi] = (int32_t)min((int64_t)stride, t1 - t0);
    }
    for (int i = tid; i < stride; i += 256) {
        const bool in = t0 + i < t1;
        // the scan indexes the run-bound rows with these: an index outside [0, sparse_dim) (only a caller that bypasses
        // pack_sparse_queries can produce one) is clamped instead of being allowed to read out of bounds
        pq_idx[(int64_t)qi * stride + i] = in ? min(max(q_idx[t0 + i], 0), sparse_dim - 1) : 0;
        pq_w[(int64_t)qi * stride + i] = in ? q_val[t0 + i] * scale : 0.f;
    }
}

// ---- scan, pipelined over ranges: grid (B, ceil(n_ranges / rpb)), 512 threads, two blocks per CU ----
// The block owns ONE query over `rpb` consecutive doc ranges.  Accumulators for one range (16 384
// docs) live in LDS; every posting of the query's terms inside the range is applied with an LDS
// integer atomic, and only the per-group maxima leave the CU.  The ranges are walked as a software
// pipeline, so that posting loads are in flight all the time:
//   * P stage (wave 0, lane = 4 term slots): the query's terms are fetched once per block; the run
//     bounds (rt_off lookups) and the posting-block base of the NEXT range are requested when the
//     bounds of the current one start being consumed, a whole range ahead;
//   * T stage (wave 0): bounds -> slot table of one STEP.  A slot is 64 consecutive postings of one
//     run, served by 16 lanes with one 16-byte load each; a step is as many whole runs as fit
//     kSlots slots (greedy), so a short query needs one step per range and a long or skewed one
//     (a df = 50 % term is a run of 8 192 postings = 128 slots) simply takes more steps — there is
//     no second code path.  T(s+2) is built while step s is applied;
//   * L stage (all waves): wave w owns items w, w + 8, ... (an item = 4 slots = one load instruction);
//     it keeps K items in registers and, right after applying item u of step s, re-issues that
//     register quad for item u of step s+1: the loads of the next step fly under this step's LDS
//     atomics, the group maxima, the zeroing and both barriers;
//   * group maxima and zeroing are one pass (a thread zeroes the words it has just reduced).
// blockIdx.x = query: blocks that run together work on the same ranges, so the run-bound lookups (a
// 40 KB row per range) and the postings of shared terms are served by L2 / the Infinity Cache.
// Round 1's form (one block per (range, query), every fetch on the block's critical path) ran at
// 0.30 of the HBM peak; an intermediate form of this pipeline with 256-posting items and ~30
// instructions per posting was instruction-issue bound (no posting loads at all: 0.30 ms of 0.37 ms
// at 2.5M docs, B = 128), which is why a posting now costs 7 vector instructions.
// Algorithmic HBM bytes per (query, range): sum over query terms of run_len * 4 (packed uint16 slot +
// fp16 weight) + 2*4 per term for the run bounds + group maxima out.
constexpr int kScanThreads = 512;
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kScanK = 8;                               // items a wave keeps in flight
constexpr int kScanKDense = 4;                          // ... in the variant for shards with dense runs: its 16 registers hold four
                                                        // 16-byte pieces of dense runs in flight instead (the longest SPARSE run of
                                                        // such a shard is < 8 189 postings = 128 slots: the smaller table holds it)
constexpr int kSlotPostings = 64;                       // postings per slot (16 lanes x 4)
constexpr int kSlots = kScanK * kScanWaves * 4;         // slots per step = everything one round of loads covers
constexpr int kDocsPerThread = kRangeDocs / kScanThreads;  // consecutive docs a thread reduces in the group-max pass
constexpr int kMaxDenseStep = 32;                       // dense runs one step may carry (more: the unit simply takes more steps)
static_assert(kDocsPerThread == 32 && (1 << kAccPadShift) == kDocsPerThread, "one padded slice per thread");
static_assert(kSlots >= kRangeDocs / kSlotPostings, "the longest possible run must fit an empty slot table");
static_assert(kScanKDense * kScanWaves * 4 >= (int)kDenseRunWords / kSlotPostings, "the longest sparse run must fit the dense variant's table");

struct ScanTab {
    unsigned e0[kSlots];      // per slot: first posting, relative to the range's block
    unsigned n[kSlots];       // per slot: postings (a multiple of 4, at most 64; 0 = unused slot)
    float ws[kSlots];         // per slot: query weight of its run * fixed-point scale
    unsigned long long base;  // first posting of the range's block
    int n_slots;              // slots in use
    int range;                // range - r0
    int range_done;           // this step completes its range: group maxima follow
    int end;                  // no such step: the block is done
};

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load at 4-byte alignment

template <bool DENSE>   // DENSE: the shard has dense runs (kDenseRunWords); without any, the code for them is compiled out
__global__ __launch_bounds__(kScanThreads, kScanThreads / 128) void sparse_scan_kernel(
    const unsigned int* __restrict__ rt_off, int64_t V1, const int64_t* __restrict__ range_base,
    const uint32_t* __restrict__ post, const int32_t* __restrict__ pq_n, const int32_t* __restrict__ pq_idx,
    const float* __restrict__ pq_w, int stride, const float* __restrict__ q_scale,
    const uint8_t* __restrict__ rowmask, int64_t n_docs, int64_t n_groups, int group_docs, int64_t n_ranges,
    int rpb, const uint32_t* __restrict__ idle_postings, float* __restrict__ gmax) {
    constexpr int K = DENSE ? kScanKDense : kScanK, NW = kScanWaves, DPT = kDocsPerThread;
    constexpr int kSlotsV = K * NW * 4;   // slots per step of this variant (the tables are sized for the larger one)
    __shared__ int acc[kAccWords];
    __shared__ ScanTab tab[2];  // T(s) lives in tab[s & 1]
    __shared__ unsigned run_first[kScanTermChunk], run_lo[kScanTermChunk], run_hi[kScanTermChunk];  // wave 0's scratch for long runs
    __shared__ float run_w[kScanTermChunk];
    // dense runs of a step (see kDenseRunWords): offset inside the range's block and scaled query weight, in a ring of three
    // steps — T(s + 2) is built while step s still reads its own
    __shared__ unsigned dl_lo[3][kMaxDenseStep];
    __shared__ float dl_w[3][kMaxDenseStep];
    __shared__ int dl_n[3];
    // Block (x, y) = chunk y of query (x + y) mod B.  Workgroups go to the eight XCDs strictly round-robin in linear order
    // (tests/probes/xcd_dispatch_census.hip: workgroup L runs on XCD L % 8, also when the grid oversubscribes the chip) and
    // B is usually a multiple of 8: without the rotation XCD i would run queries i, i + 8, ... of EVERY chunk, and the XCD
    // that drew the expensive queries (frequent terms, long queries) would finish last — Zipfian postings at 10M docs,
    // B = 128: 4.15 ms without the rotation, 3.38 ms with it (uniform postings: 0.75 ms either way).
    const int qi = (int)((blockIdx.x + blockIdx.y) % gridDim.x);
    const int64_t r0 = (int64_t)blockIdx.y * rpb;
    const int n_r = (int)min((int64_t)rpb, n_ranges - r0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        int4* a4 = reinterpret_cast<int4*>(acc);
        for (int i = tid; i < kAccWords / 4; i += kScanThreads) a4[i] = make_int4(0, 0, 0, 0);
    }

    // ---- P / T stages: wave 0 only.  A "unit" is (range, chunk of 256 query terms); lane l owns term slots 4l .. 4l+3.
    const int n_terms = pq_n[qi];
    const int n_chunks = max(1, (n_terms + kScanTermChunk - 1) / kScanTermChunk);
    const int G = n_r * n_chunks;  // units of this block
    int t4[4] = {0, 0, 0, 0};
    float wc[4] = {0.f, 0.f, 0.f, 0.f}, wn[4] = {0.f, 0.f, 0.f, 0.f};          // weights of the current / next unit
    unsigned loc[4] = {0, 0, 0, 0}, hic[4] = {0, 0, 0, 0}, lon[4] = {0, 0, 0, 0}, hin[4] = {0, 0, 0, 0};  // run bounds
    unsigned long long base_c = 0, base_n = 0;                                // posting-block base of the unit's range
    int g_cur = 0, p_pos = 0;  // unit being consumed, first term slot of it not yet in a step
    auto p_terms = [&](int chunk, float (&w)[4]) {  // the chunk's term ids and scaled weights (slots past the query hold 0 / 0.f)
        const int slot = chunk * kScanTermChunk + 4 * lane;
        int4 ti = make_int4(0, 0, 0, 0);
        float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < stride) {
            ti = *reinterpret_cast<const int4*>(pq_idx + (int64_t)qi * stride + slot);
            tw = *reinterpret_cast<const float4*>(pq_w + (int64_t)qi * stride + slot);
        }
        t4[0] = ti.x; t4[1] = ti.y; t4[2] = ti.z; t4[3] = ti.w;
        w[0] = tw.x; w[1] = tw.y; w[2] = tw.z; w[3] = tw.w;
    };
    auto p_issue = [&](int g, unsigned (&lo)[4], unsigned (&hi)[4], float (&w)[4], unsigned long long& base) {
        const int range = g / n_chunks, chunk = g - range * n_chunks;
        if (n_chunks > 1) p_terms(chunk, w);
        const unsigned int* offs = rt_off + (r0 + range) * V1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = offs[t4[j]];
            hi[j] = offs[t4[j] + 1];
        }
        base = (unsigned long long)range_base[r0 + range];
    };
    auto p_build = [&](ScanTab& T, int step_id) {  // the next step: as many whole runs of the current unit as fit the slot table
        const int ring = step_id % 3;
        if (g_cur >= G) {
#pragma unroll 1
            for (int i = lane; i < kSlotsV; i += 64) T.n[i] = 0u;
            if (lane == 0) { T.n_slots = 0; T.range_done = 0; T.end = 1; T.base = 0; dl_n[ring] = 0; }
            return;
        }
        const int range = g_cur / n_chunks, chunk = g_cur - range * n_chunks;
        const int nt = min(kScanTermChunk, max(n_terms - chunk * kScanTermChunk, 0));
        unsigned cnt[4], dns[4], mine = 0, mine_d = 0;   // slots / dense-list entries of the lane's four runs
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 4 * lane + j;
            const bool in_unit = t >= p_pos && t < nt;
            const unsigned len = hic[j] - loc[j];
            dns[j] = (DENSE && in_unit && len == kDenseRunWords) ? 1u : 0u;   // applied by the docs' owner threads, not through slots
            cnt[j] = (in_unit && !dns[j]) ? (len + kSlotPostings - 1) / kSlotPostings : 0u;
            mine += cnt[j];
            mine_d += dns[j];
        }
        const unsigned incl = wave_scan_add(mine);
        unsigned first = incl - mine;
        unsigned first_d = wave_scan_add(mine_d) - mine_d;
        unsigned taken = 0;  // term slots of this lane that go into the step
        unsigned used = 0;   // slots this lane's runs add to the step
        unsigned used_d = 0; // dense-list entries this lane's runs add to the step
        unsigned longest = 0;
        // run j of this lane is in range and still fits the step's slot table and dense list; both prefixes are monotone
        auto run_fits = [&](int j, unsigned start, unsigned dstart) {
            const int t = 4 * lane + j;
            return t >= p_pos && t < nt && start + cnt[j] <= (unsigned)kSlotsV && dstart + dns[j] <= (unsigned)kMaxDenseStep;
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (run_fits(j, first, first_d)) {
                ++taken;
                used += cnt[j];
                used_d += dns[j];
                longest = max(longest, cnt[j]);
                if (dns[j]) {
                    dl_lo[ring][first_d] = loc[j];
                    dl_w[ring][first_d] = wc[j];
                }
            }
            first += cnt[j];
            first_d += dns[j];
        }
        first -= mine;
        first_d -= mine_d;
        // the prefix is monotone, so the runs that fit are exactly the first ones: their slot counts simply add up
        taken = wave_sum(taken);
        used = wave_sum(used);
        used_d = wave_sum(used_d);
        if (!__any(longest > 8u)) {
            // short runs (the usual case: ~3 slots each): every lane writes the slots of its own runs
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (run_fits(j, first, first_d)) {
                    unsigned e = loc[j];
#pragma unroll 1
                    for (unsigned c = 0; c < cnt[j]; ++c, e += kSlotPostings) {
                        T.e0[first + c] = e;
                        T.n[first + c] = min((unsigned)kSlotPostings, hic[j] - e);
                        T.ws[first + c] = wc[j];
                    }
                }
                first += cnt[j];
                first_d += dns[j];
            }
        } else {
            // a long run (a frequent term: up to 256 slots) would keep ONE lane writing for thousands of cycles between
            // the step's barriers: publish the runs, then every lane fills slots lane, lane + 64, ... by looking its run up
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * lane + j;
                run_first[t] = first;
                run_lo[t] = loc[j];
                run_hi[t] = hic[j];
                run_w[t] = wc[j];
                first += cnt[j];
            }
            const int t_end = p_pos + (int)taken;  // runs [p_pos, t_end) are in the step
#pragma unroll 1
            for (unsigned sl = (unsigned)lane; sl < used; sl += 64) {
                int lo_t = p_pos, hi_t = t_end - 1;  // largest t with run_first[t] <= sl (empty runs share their successor's start)
#pragma unroll 1
                while (lo_t < hi_t) {
                    const int mid = (lo_t + hi_t + 1) >> 1;
                    if (run_first[mid] <= sl) lo_t = mid; else hi_t = mid - 1;
                }
                const unsigned e = run_lo[lo_t] + (sl - run_first[lo_t]) * kSlotPostings;
                T.e0[sl] = e;
                T.n[sl] = min((unsigned)kSlotPostings, run_hi[lo_t] - e);
                T.ws[sl] = run_w[lo_t];
            }
        }
#pragma unroll 1
        for (int i = (int)used + lane; i < kSlotsV; i += 64) T.n[i] = 0u;  // unused slots fetch idle postings
        const int p_end = p_pos + (int)taken;
        const bool unit_done = p_end >= nt;
        if (lane == 0) {
            T.n_slots = (int)used;
            T.range = range;
            T.base = base_c;
            T.range_done = unit_done && chunk == n_chunks - 1;
            T.end = 0;
            dl_n[ring] = (int)used_d;
        }
        p_pos = p_end;
        if (unit_done) {  // on to the next unit; request the bounds of the one after it
            ++g_cur;
            p_pos = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { loc[j] = lon[j]; hic[j] = hin[j]; wc[j] = wn[j]; }
            base_c = base_n;
            if (g_cur + 1 < G) p_issue(g_cur + 1, lon, hin, wn, base_n);
        }
    };

    // ---- L stage: lane group q = lane / 16 serves slot 4 * item + q with one 16-byte load per lane
    const int sub = lane >> 4, l16 = lane & 15;
    // Lanes with nothing to fetch (behind the end of a run, unused slots) read 16 bytes of IDLE postings
    // instead: weight 0, each lane's aimed at an accumulator pad word of its own.  So there is no branch
    // around the load and no validity test per posting: every lane applies four postings per item.
    const uint32_t* idle = idle_postings + 4 * lane;
    auto load_item = [&](const ScanTab& T, const uint32_t* pp, int item) -> u32x4_a4 {
        const int slot = 4 * item + sub;
        const uint32_t* a = (unsigned)(4 * l16) < T.n[slot] ? pp + T.e0[slot] + 4 * l16 : idle;
#if defined(HR_ABLATE) && HR_ABLATE == 2  // timing-only build: no posting loads, synthetic postings instead
        const unsigned x = (T.e0[slot] + 4 * l16) * 2654435761u;
        return (u32x4_a4){(x >> 7 & 0x3FFFu) | 0x3C000000u, (x >> 11 & 0x3FFFu) | 0x3C000000u, (x >> 5 & 0x3FFFu) | 0x3C000000u, (x >> 13 & 0x3FFFu) | 0x3C000000u};
#else
        return *reinterpret_cast<const u32x4_a4*>(a);   // (a non-temporal load here: 0.76 -> 0.91 ms, profiles/r4_experiments/sparse_layout_nt_probe.txt)
#endif
    };
    auto apply_item = [&](const ScanTab& T, int item, const u32x4_a4& v) {
        const float wr = T.ws[4 * item + sub];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t p = v[j];
            const int c = (int)fmaf(posting_weight(p), wr, 1.0f);  // trunc(product + 1): see sparse_query_prep_kernel
#if defined(HR_ABLATE) && HR_ABLATE == 1  // timing-only build: postings are fetched but not applied
            asm volatile("" ::"v"(c), "v"(p & 0xFFFFu));
#else
            atomicAdd(&acc[p & 0xFFFFu], c);
#endif
        }
    };

    // ---- prologue: T(0), T(1), round-0 items of step 0 in flight
    if (wave == 0) {
        if (n_chunks == 1) {
            p_terms(0, wc);
#pragma unroll
            for (int j = 0; j < 4; ++j) wn[j] = wc[j];
        }
        p_issue(0, loc, hic, wc, base_c);
        if (G > 1) p_issue(1, lon, hin, wn, base_n);
#pragma unroll 1
        for (int i = 0; i < 2; ++i) p_build(tab[i], i);
    }
    __syncthreads();
    u32x4_a4 pk[K];
    {
        const uint32_t* pp0 = post + tab[0].base;
#pragma unroll
        for (int u = 0; u < K; ++u) pk[u] = load_item(tab[0], pp0, wave + NW * u);
    }
    const float scale = q_scale[qi];
    const float inv = scale > 0.f ? 1.0f / scale : 0.f;
#pragma unroll 1
    for (int s = 0;; ++s) {
        const ScanTab& Tc = tab[s & 1];
        const ScanTab& Tn = tab[(s + 1) & 1];
        if (Tc.end) break;
        const int n_c = Tc.n_slots;
        const bool range_done = Tc.range_done != 0;
        const int64_t range = r0 + Tc.range;
        const uint32_t* pp_c = post + Tc.base;   // read now: T(s) is rebuilt as T(s + 2) behind the barrier
        const uint32_t* pp_n = post + Tn.base;
        // apply item u of step s, refill the quad with item u of step s+1
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int item = wave + NW * u;
            if (4 * item < n_c) apply_item(Tc, item, pk[u]);
            pk[u] = load_item(Tn, pp_n, item);  // unconditional (an item past the step reads the block's first bytes): no branch around the load
        }
        __syncthreads();  // every posting of step s is in the accumulators; T(s) is free
        // dense runs of the step: no atomics are in flight between the two barriers, so the thread that owns 32 consecutive
        // docs adds their products to its own accumulators with plain read-modify-writes (two docs per 64-bit word), 16
        // bytes (8 docs) at a time to keep the registers of the posting pipeline where they are
        if (DENSE && dl_n[s % 3] > 0) {
            const int ring = s % 3, nd = dl_n[ring];
            unsigned long long* mine = reinterpret_cast<unsigned long long*>(acc + tid * kAccSlice);
            // four 16-byte pieces (the quarters of a run) in flight: quarter q of the next run is requested right after
            // quarter q of this one has been applied — the registers are the ones the posting pipeline gave up (kScanKDense)
            auto piece = [&](int d, int q) -> u32x4_a4 {
                return *reinterpret_cast<const u32x4_a4*>(pp_c + dl_lo[ring][d] + 4 * (q * kScanThreads + tid));   // dense_run_pos
            };
            u32x4_a4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = piece(0, q);
#pragma unroll 1
            for (int d = 0; d < nd; ++d) {
                const float wr = dl_w[ring][d];
                const int dn = d + 1 < nd ? d + 1 : d;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t w2 = r[q][k];
                        const unsigned short h0 = (unsigned short)(w2 & 0xFFFFu), h1 = (unsigned short)(w2 >> 16);
                        const int c0 = h0 != kDenseAbsent ? (int)fmaf(posting_weight((uint32_t)h0 << 16), wr, 1.0f) : 0;
                        const int c1 = h1 != kDenseAbsent ? (int)fmaf(posting_weight((uint32_t)h1 << 16), wr, 1.0f) : 0;
                        const unsigned long long x = mine[4 * q + k];
                        mine[4 * q + k] = (unsigned long long)(unsigned)((int)(unsigned)x + c0) |
                                          ((unsigned long long)(unsigned)((int)(unsigned)(x >> 32) + c1) << 32);
                    }
                    r[q] = piece(dn, q);
                }
            }
        }
        if (wave == 0) p_build(tab[s & 1], s + 2);  // T(s+2)
        if (range_done) {
            // per-group maxima of the finished range, and zero for the next one: a thread owns DPT consecutive docs
            const int64_t doc0 = range * kRangeDocs + (int64_t)tid * DPT;
            // one LDS exchange per pair of docs reads the accumulators and leaves zeros behind
            unsigned long long* mine = reinterpret_cast<unsigned long long*>(acc + tid * kAccSlice);
            int v[DPT];
#pragma unroll
            for (int i = 0; i < DPT / 2; ++i) {
                const unsigned long long x = __hip_atomic_exchange(mine + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                v[2 * i] = (int)(unsigned)x;
                v[2 * i + 1] = (int)(unsigned)(x >> 32);
            }
            if (rowmask) {
                unsigned alive = 0u;  // filter bits of the thread's docs (bit i = doc0 + i)
#pragma unroll
                for (int b = 0; b < DPT / 8; ++b)  // docs past the shard have empty accumulators: their bits do not matter
                    alive |= (doc0 + 8 * b < n_docs ? (unsigned)rowmask[(doc0 >> 3) + b] : 0u) << (8 * b);
#pragma unroll
                for (int i = 0; i < DPT; ++i) v[i] = (alive >> i) & 1u ? v[i] : 0;
            }
            int m0 = 0, m1 = 0;  // maxima of docs 0..15 / 16..31 of the thread's slice
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                m0 = max(m0, v[i]);
                m1 = max(m1, v[16 + i]);
            }
            if (group_docs == 16) {
                const int64_t group = range * (kRangeDocs / 16) + tid * (DPT / 16);
                if (group < n_groups) gmax[(int64_t)qi * n_groups + group] = (float)m0 * inv;
                if (group + 1 < n_groups) gmax[(int64_t)qi * n_groups + group + 1] = (float)m1 * inv;
            } else {
                int mm = max(m0, m1);
                mm = max(mm, __shfl_xor(mm, 1));  // 64-doc group = two threads
                const int64_t group = range * (kRangeDocs / 64) + (tid >> 1);
                if ((tid & 1) == 0 && group < n_groups) gmax[(int64_t)qi * n_groups + group] = (float)mm * inv;
            }
        }
        __syncthreads();  // zeroed accumulators and T(s+2) are visible
    }
}

// ---- refine: one WAVE per candidate doc at a time (group_docs = 16 or 64 docs per group) ----
// Canonical score: walk the doc's CSR entries in stored order, look each index
// up in the query's sorted terms, accumulate exact products in fp64.
// Restated in oracle/oracle.c:oracle_sparse_scores().
// The block (4 waves x 64 candidate docs of one query) stages the query in LDS
// together with a 32768-bit hashed membership filter, so the ~99 % of entries
// that cannot match cost one LDS read instead of a binary search.
// Each wave walks its docs (docs_per_wave of them) one after the other with lane = entry: the doc's indices
// and values arrive as two coalesced loads per 64 entries (the entries of the next
// kRefinePrefetch docs are already in flight), every lane tests its own entry, and the
// few matching products are added in ENTRY ORDER (ballot, lowest lane first), which is
// the canonical order.  (One thread per doc, the first form of this kernel, read each
// row with a 400-byte stride between lanes and thrashed the L1: 0.97 ms per 128-query
// batch against the scans it shares the chip with.)
constexpr int kFilterBits = 32768;
constexpr int kRefinePrefetch = 12;  // docs whose entries a wave has in flight: the chain is bound by memory latency (two
                                     // dependent round trips per doc), so depth buys time almost linearly until the
                                     // per-doc filter test + ballot (~100 cycles) is what is left (2 -> 8: r3 probe; 8 -> 12: the last 2 %, same registers)

// Stage query qi in LDS for the refine: s_filter (kFilterBits / 32 words), s_idx / s_val (q_cap entries each).
// All threads of the block take part; returns the number of staged terms.
__device__ inline int refine_sparse_stage_query(const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
                                                const float* __restrict__ q_val, int qi, int q_cap,
                                                unsigned int* s_filter, int32_t* s_idx, float* s_val) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t t0 = q_indptr[qi];
    // a query longer than q_cap was already marked "never proven" by the prep kernel (q_eps = inf)
    const int nt = min((int)(q_indptr[qi + 1] - t0), q_cap);
    for (int i = tid; i < kFilterBits / 32; i += nthr) s_filter[i] = 0u;
    __syncthreads();
    for (int i = tid; i < nt; i += nthr) {
        const int32_t t = q_idx[t0 + i];
        s_idx[i] = t;
        s_val[i] = q_val[t0 + i];
        atomicOr(&s_filter[(t & (kFilterBits - 1)) >> 5], 1u << (t & 31));
    }
    __syncthreads();
    return nt;
}

// The fused finishing kernel's form for queries of up to kHashMaxTerms terms: the same membership filter, and for the
// entries that pass it an open-addressing hash table in LDS (keys s_hkey[hs], values s_hval[hs], hs a power of two >= 4 x
// the terms: load <= 0.25, linear probing, -1 = empty) instead of the lower-bound search — one or two dependent LDS reads
// where the search costs the wave ~7 as soon as ONE lane passes the filter.  A term that occurs twice in the query keeps
// its FIRST value, as the search does.  (The hash alone, without the filter in front, lost: every lane probes, and the
// longest probe sequence of 64 lanes is 3 - 4 reads — profiles/r4_experiments/finish_prerank_hash_*.txt.)
constexpr int kHashMaxTerms = 256;
__host__ __device__ inline int sparse_hash_slots(int q_cap) { return q_cap <= 128 ? 512 : 1024; }
__device__ inline unsigned sparse_hash(int32_t t, int hs) { return ((unsigned)t * 0x9E3779B1u) >> (hs == 512 ? 23 : 22); }
__device__ inline void refine_sparse_stage_query_hash(const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
                                                      const float* __restrict__ q_val, int qi, int q_cap, int hs,
                                                      unsigned int* s_filter, int32_t* s_hkey, float* s_hval) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t t0 = q_indptr[qi];
    const int nt = min((int)(q_indptr[qi + 1] - t0), q_cap);
    for (int i = tid; i < kFilterBits / 32; i += nthr) s_filter[i] = 0u;
    for (int i = tid; i < hs; i += nthr) s_hkey[i] = -1;
    __syncthreads();
    for (int i = tid; i < nt; i += nthr) {
        const int32_t t = q_idx[t0 + i];
        if (t < 0 || (i > 0 && q_idx[t0 + i - 1] == t)) continue;   // the terms are sorted: a repeat sits next to its first
        atomicOr(&s_filter[(t & (kFilterBits - 1)) >> 5], 1u << (t & 31));
        unsigned h = sparse_hash(t, hs);
        while (atomicCAS(&s_hkey[h], -1, t) != -1) h = (h + 1) & (hs - 1);
        s_hval[h] = q_val[t0 + i];
    }
    __syncthreads();
}
struct SparseLookupSorted {   // membership filter, then lower-bound search in the sorted terms
    const unsigned int* s_filter;
    const int32_t* s_idx;
    const float* s_val;
    int nt;
    __device__ inline bool operator()(int32_t tt, float* qv) const {
        if (!((s_filter[(tt & (kFilterBits - 1)) >> 5] >> (tt & 31)) & 1u)) return false;
        int lo = 0, hi = nt;  // first position with s_idx[pos] >= tt
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_idx[mid] < tt) lo = mid + 1; else hi = mid;
        }
        if (lo < nt && s_idx[lo] == tt) {
            *qv = s_val[lo];
            return true;
        }
        return false;
    }
};
struct SparseLookupHash {
    const unsigned int* s_filter;
    const int32_t* s_hkey;
    const float* s_hval;
    int hs;
    __device__ inline bool operator()(int32_t tt, float* qv) const {
        if (!((s_filter[(tt & (kFilterBits - 1)) >> 5] >> (tt & 31)) & 1u)) return false;
        unsigned h = sparse_hash(tt, hs);
        for (;;) {
            const int32_t k = s_hkey[h];
            if (k == tt) {
                *qv = s_hval[h];
                return true;
            }
            if (k == -1) return false;
            h = (h + 1) & (hs - 1);
        }
    }
};

// One wave walks the n_here (<= 64) candidate doc slots slot0 .. slot0 + n_here - 1 of a query whose terms are staged in
// LDS (`lookup`); emit(slot, keep, score, doc) is called once per slot by the lane that owns it.
template <typename Lookup, typename Emit>
__device__ inline void refine_sparse_chain(const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                           const float* __restrict__ val, const uint8_t* __restrict__ rowmask,
                                           const int32_t* cand_q, int group_docs, int64_t n_docs, int slot0, int n_here,
                                           Lookup lookup, Emit emit) {
    const int lane = threadIdx.x & 63;
    // lane l describes doc slot0 + l
    const int slot = slot0 + lane;
    int64_t doc = -1, p0 = 0;
    int len = 0;
    bool valid = false;
    if (lane < n_here) {
        const int32_t group = cand_q[slot / group_docs];
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
        double prod = 0.0;
        float qv = 0.f;
        const bool hit = tt >= 0 && lookup(tt, &qv);
        if (hit) prod = __dmul_rn((double)vv, (double)qv);
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
    if (lane < n_here) emit(slot, valid && my_score > 0.f, my_score, doc);
}

template <bool HASH>
__global__ __launch_bounds__(256) void refine_sparse_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
    const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_idx,
    const float* __restrict__ q_val, const uint8_t* __restrict__ rowmask,
    const int32_t* __restrict__ cand, int C, int group_docs, int64_t n_docs, int q_cap, int docs_per_wave,
    float* __restrict__ out_score, int32_t* __restrict__ out_row) {
    // dynamic LDS: filter words, then q_cap query indices and q_cap query values (q_cap = the batch's
    // longest query rounded up, so short queries leave the CU room for many blocks)
    extern __shared__ unsigned int refine_lds[];
    unsigned int* s_filter = refine_lds;
    const int hs = sparse_hash_slots(q_cap);    // HASH: keys and values of the table instead of the sorted terms
    int32_t* s_idx = reinterpret_cast<int32_t*>(refine_lds + kFilterBits / 32);
    float* s_val = reinterpret_cast<float*>(s_idx + (HASH ? hs : q_cap));
    const int qi = blockIdx.y, tid = threadIdx.x;
    int nt = 0;
    if constexpr (HASH) refine_sparse_stage_query_hash(q_indptr, q_idx, q_val, qi, q_cap, hs, s_filter, s_idx, s_val);
    else nt = refine_sparse_stage_query(q_indptr, q_idx, q_val, qi, q_cap, s_filter, s_idx, s_val);
    const int n_slots = C * group_docs;
    // A wave's docs are a serial chain (each waits for its entries): docs_per_wave (8..64, the host's choice) trades the
    // length of that chain against the number of blocks that rebuild the filter.
    const int slot0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (tid >> 6)) * docs_per_wave);  // first doc slot of this wave
    if (slot0 >= n_slots) return;
    const int n_here = min(docs_per_wave, n_slots - slot0);
    auto emit = [&](int slot, bool keep, float score, int64_t doc) {
        const int64_t o = (int64_t)qi * n_slots + slot;
        out_score[o] = keep ? score : -__builtin_inff();
        out_row[o] = keep ? (int32_t)doc : -1;
    };
    if constexpr (HASH)
        refine_sparse_chain(indptr, idx, val, rowmask, cand + (int64_t)qi * C, group_docs, n_docs, slot0, n_here,
                            SparseLookupHash{s_filter, s_idx, s_val, hs}, emit);
    else
        refine_sparse_chain(indptr, idx, val, rowmask, cand + (int64_t)qi * C, group_docs, n_docs, slot0, n_here,
                            SparseLookupSorted{s_filter, s_idx, s_val, nt}, emit);
}

}  // namespace hbmrag
