// Dense shard kernels: ingest re-tiling, row norms, query prep, the MFMA scan
// that streams the shard once per query batch, and the canonical fp64 refine.
//
// Replaces the server-side HNSW/COSINE search behind Collection.search
// (reference src/advanced_rag/indexing.py:503-525) with an exact FLAT scan.
#pragma once
#include <type_traits>

#include "common.h"

namespace hbmrag {

// ---------------------------------------------------------------------------
// ingest: row-major rows -> tiled shard layout (see chunk_index)
// SRC = float (convert to the store type) or the store type itself.
template <typename STORE, typename SRC>
__global__ void tile_rows_kernel(const SRC* __restrict__ src, int64_t n, int dim, int KT,
                                 int64_t row0, chunk_t* __restrict__ tiles) {
    constexpr int EPC = kChunkBytes / (int)sizeof(STORE);
    const int kchunks = KT * 4;
    int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n * kchunks) return;
    int64_t i = tid / kchunks;
    int kc = (int)(tid - i * kchunks);
    union { chunk_t v; STORE e[EPC]; } out;
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
        int k = kc * EPC + j;
        out.e[j] = (k < dim) ? (STORE)src[i * dim + k] : (STORE)0;
    }
    tiles[chunk_index(row0 + i, kc, KT)] = out.v;
}

// One thread per row: sum of squares in fp64, k-ordered (the canonical norm the
// oracle restates), and the fp32 scan scale (1/||x|| for COSINE, 1 for IP).
template <typename STORE>
__global__ void row_norms_kernel(const chunk_t* __restrict__ tiles, int KT, int64_t row0, int64_t n,
                                 int cosine, double* __restrict__ norm2, float* __restrict__ scale,
                                 unsigned int* __restrict__ max_norm_bits) {
    constexpr int EPC = kChunkBytes / (int)sizeof(STORE);
    int64_t r = row0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= row0 + n) return;
    double s = 0.0;
    for (int kc = 0; kc < KT * 4; ++kc) {
        union { chunk_t v; STORE e[EPC]; } c;
        c.v = tiles[chunk_index(r, kc, KT)];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            double x = (double)c.e[j];
            s = __dadd_rn(s, __dmul_rn(x, x));
        }
    }
    norm2[r] = s;
    float nrm = (float)sqrt(s);
    scale[r] = cosine ? (s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.0f) : 1.0f;
    atomicMax(max_norm_bits, __float_as_uint(nrm));  // non-negative floats order as uints
}

// ---------------------------------------------------------------------------
// query prep: one block per query slot.  Writes the unit-normalised query in
// the B-operand fragment layout of the scan (fp16 or fp32 to match the shard)
// and the canonical fp64 |q|^2.  Slots >= B are zero-filled padding.
template <typename STORE>
__global__ void prep_queries_kernel(const float* __restrict__ q, int B, int dim, int KT,
                                    chunk_t* __restrict__ qfrag, double* __restrict__ qn2) {
    constexpr int EPC = kChunkBytes / (int)sizeof(STORE);
    const int slot = blockIdx.x;
    const int g = slot >> 4, col = slot & 15;
    __shared__ float s_inv;
    __shared__ float s_q[HR_MAX_DIM];
    if (slot < B)
        for (int k = threadIdx.x; k < dim; k += blockDim.x) s_q[k] = q[(int64_t)slot * dim + k];
    __syncthreads();
    if (threadIdx.x == 0) {
        // canonical |q|^2: k-ordered fp64 sum (the oracle does the same); the
        // operands come from LDS so only the add chain is serial
        double s = 0.0;
        if (slot < B) {
            for (int k = 0; k < dim; ++k) {
                double x = (double)s_q[k];
                s = __dadd_rn(s, __dmul_rn(x, x));
            }
            qn2[slot] = s;
        }
        s_inv = (s > 0.0) ? (float)(1.0 / sqrt(s)) : 0.0f;
    }
    __syncthreads();
    const float inv = s_inv;
    for (int kc = threadIdx.x; kc < KT * 4; kc += blockDim.x) {
        union { chunk_t v; STORE e[EPC]; } out;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
            int k = kc * EPC + j;
            float x = (slot < B && k < dim) ? s_q[k] * inv : 0.0f;
            out.e[j] = (STORE)x;
        }
        int kt = kc >> 2, c = kc & 3;
        qfrag[((int64_t)g * KT + kt) * kTileChunks + col + 16 * c] = out.v;
    }
}

// ---------------------------------------------------------------------------
// The scan.  A wave owns candidate groups (64 consecutive rows = 4 row blocks)
// in a grid-stride loop and streams their tiles straight from HBM into VGPRs
// (no LDS round trip: each byte of the shard is used by exactly one wave).
// The query batch (16*G queries) sits in LDS in fragment order and is re-read
// per k-step with conflict-free ds_read_b128.  Per MFMA result the lane holds
// 4 rows x 1 query; the epilogue keeps only the maximum over the group, so the
// kernel's output is one float per (query, 64 rows): N/64 * B * 4 bytes.
//
// HBM traffic per launch (algorithmic): n_rows * Dpad * sizeof(STORE) + 4 * n_rows.
template <typename STORE>
struct Mfma;
template <>
struct Mfma<_Float16> {
    __device__ static inline void run(const chunk_t& a, const chunk_t& b, f32x4_t& c) {
        union { chunk_t v; half8_t h; } ua, ub;
        ua.v = a; ub.v = b;
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, c, 0, 0, 0);
    }
};
template <>
struct Mfma<float> {
    // One 16 B chunk = 4 k-values; element j of every lane feeds MFMA step j.
    // Any k order is fine as long as A and B agree, which the shared chunk
    // layout guarantees.
    __device__ static inline void run(const chunk_t& a, const chunk_t& b, f32x4_t& c) {
        union { chunk_t v; float f[4]; } ua, ub;
        ua.v = a; ub.v = b;
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(ua.f[j], ub.f[j], c, 0, 0, 0);
    }
};

// PF = depth of the per-wave register prefetch ring (k-steps in flight); KT is
// padded to a multiple of PF at hr_create so ring slots stay compile-time.
template <typename STORE, int G, int RS, int PF, int NRB>
__global__ __launch_bounds__(512) void dense_scan_kernel(
    const chunk_t* __restrict__ tiles, const chunk_t* __restrict__ qfrag, const float* __restrict__ scale,
    const uint8_t* __restrict__ rowmask, float* __restrict__ gmax, int nq, int KT, int64_t n_rows,
    int64_t n_super) {
    // NRB = row blocks per candidate group: 4 (64-row groups) or 1 (16-row groups).
    // `group` below walks SUPER-groups of 4 row blocks either way; gmax is [nq][n_super*4/NRB].
    static_assert(NRB == 1 || NRB == kRowBlocksPerSuper, "group = one row block or one super-group");
    const int64_t n_groups = n_super;  // loop bound (super-groups)
    const int64_t gmax_stride = n_super * (kRowBlocksPerSuper / NRB);
    extern __shared__ chunk_t lds_q[];  // [G][KT][64] chunks
    constexpr int kPairs = kRowBlocksPerSuper / RS;
    const int nq_chunks = G * KT * kTileChunks;
    for (int i = threadIdx.x; i < nq_chunks; i += blockDim.x) lds_q[i] = qfrag[i];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int waves_per_block = blockDim.x >> 6;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * waves_per_block + (threadIdx.x >> 6)));
    const int64_t total_waves = (int64_t)gridDim.x * waves_per_block;
    const int quad = lane >> 4;
    const float NEG_INF = -__builtin_inff();

    // Prefetch cursor: walks (group, pair, kt) exactly PF steps ahead of the
    // consumer, across pair and group boundaries, so the wave's HBM stream
    // never drains.  All of it is wave-uniform (scalar) state.
    int64_t pf_group = wave;
    int pf_pair = 0, pf_kt = 0;
    chunk_t ring[PF][RS];
    auto prefetch = [&](int slot) {
        // Unconditional load (no control flow around VMEM, so the compiler keeps
        // counted vmcnt waits): past the wave's last group re-read its first one.
        // (a wave WITHOUT any group — the last block may hold up to three — re-reads group 0: `wave` itself is past the end of
        // the shard there, and with 48 KiB super-groups at D = 384 that read crossed into an unmapped page: round 4)
        const int64_t grp = pf_group < n_groups ? pf_group : (wave < n_groups ? wave : 0);
        const int64_t rb = grp * kRowBlocksPerSuper + pf_pair * RS;
#pragma unroll
        for (int s = 0; s < RS; ++s)
            ring[slot][s] = __builtin_nontemporal_load(tiles + ((rb + s) * KT + pf_kt) * kTileChunks + lane);
        const bool wrap_kt = (pf_kt + 1 == KT);
        pf_kt = wrap_kt ? 0 : pf_kt + 1;
        const bool wrap_pair = wrap_kt && (pf_pair + 1 == kPairs);
        pf_pair = wrap_kt ? (wrap_pair ? 0 : pf_pair + 1) : pf_pair;
        pf_group = wrap_pair ? pf_group + total_waves : pf_group;
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) prefetch(j);

    for (int64_t group = wave; group < n_groups; group += total_waves) {
        float m[G];
#pragma unroll
        for (int g = 0; g < G; ++g) m[g] = NEG_INF;
        const bool tail = (group + 1) * kSuperRows > n_rows || rowmask != nullptr;

#pragma unroll 1
        for (int pair = 0; pair < kPairs; ++pair) {
            f32x4_t acc[RS][G];
#pragma unroll
            for (int s = 0; s < RS; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[s][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
            for (int kt0 = 0; kt0 < KT; kt0 += PF) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const chunk_t b = lds_q[(g * KT + kt0 + j) * kTileChunks + lane];
#pragma unroll
                        for (int s = 0; s < RS; ++s) Mfma<STORE>::run(ring[j][s], b, acc[s][g]);
                    }
                    prefetch(j);  // refill the slot just consumed: PF-1 steps of lookahead
                }
            }
            // epilogue: lane holds rows row0..row0+3 of row block s for query 16g+(lane&15)
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const int64_t row0 = (group * kRowBlocksPerSuper + pair * RS + s) * kRowsPerBlock + quad * 4;
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(scale + row0);
                float ok[4] = {1.f, 1.f, 1.f, 1.f};
                if (tail) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int64_t row = row0 + r;
                        bool v = row < n_rows;
                        if (v && rowmask) v = (rowmask[row >> 3] >> (row & 7)) & 1;
                        ok[r] = v ? 1.f : 0.f;
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float mr = NEG_INF;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[s][g][r] * sc[r];
                        v = (ok[r] != 0.f) ? v : NEG_INF;
                        mr = fmaxf(mr, v);
                    }
                    if (NRB == 1) {  // this row block is a candidate group of its own
                        mr = col4_max(mr);
                        if (lane < 16 && 16 * g + lane < nq)
                            gmax[(int64_t)(16 * g + lane) * gmax_stride + group * kRowBlocksPerSuper + pair * RS + s] = mr;
                    } else {
                        m[g] = fmaxf(m[g], mr);
                    }
                }
            }
        }
        if (NRB != 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float v = m[g];
                v = col4_max(v);
                if (lane < 16 && 16 * g + lane < nq) gmax[(int64_t)(16 * g + lane) * gmax_stride + group] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Large-batch scan: 16*GQ = 128 queries per pass (GQ = 8 is what is instantiated: at GQ = 16 the
// 2 x 16 x 4 accumulators push hipcc into scratch and the pass runs at half the HBM rate, i.e. no
// better than two 128-query passes — measured 5.79 ms vs 2 x 2.76 ms at 10M x 768).  The whole query
// tile no longer fits LDS (128 x 1024 fp16 = 256 KiB), so it is streamed through LDS in
// k-chunks of BKT = 2 MFMA steps, double buffered and shared by the block's 8
// waves, which therefore walk k in lockstep (one __syncthreads per chunk) while
// each wave keeps its own 2 row blocks' accumulators (2 x GQ x 4 registers) and
// still streams its corpus tiles straight from HBM into the register ring.
// Per round a block reads 8 x 2 row blocks from HBM and the query tile once from
// L2: L2 : HBM traffic = 1 : 1.  MFMA work per corpus KiB is GQ x 16 cycles per
// SIMD (GQ = 16: 64 cycles per KiB per CU against ~100 cycles per KiB of HBM
// supply), so the pass stays HBM-bound.
template <typename STORE, int GQ, int NRB>
__global__ __launch_bounds__(512) void dense_scan_bigq_kernel(
    const chunk_t* __restrict__ tiles, const chunk_t* __restrict__ qfrag, const float* __restrict__ scale,
    const uint8_t* __restrict__ rowmask, float* __restrict__ gmax, int nq, int KT, int64_t n_rows,
    int64_t n_super) {
    constexpr int RS = 2, BKT = 2, PF = 4;  // query chunk = 2 k-steps (16 staging registers), corpus ring = 4 k-steps
    constexpr int kPairs = kRowBlocksPerSuper / RS;
    constexpr int kFrags = GQ * BKT;           // 1 KiB query fragments per k-chunk
    constexpr int kStage = kFrags / 8;         // fragments each of the 8 waves stages per chunk
    static_assert(kFrags % 8 == 0, "8 waves share the staging");
    static_assert(NRB == 1 || NRB == kRowBlocksPerSuper, "group = one row block or one super-group");
    extern __shared__ chunk_t lds_q[];         // [2][GQ][BKT][64]
    static_assert(PF == 2 * BKT, "two query chunks per trip of the ring");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 8 + wid));
    const int64_t total_waves = (int64_t)gridDim.x * 8;
    const int64_t gmax_stride = n_super * (kRowBlocksPerSuper / NRB);
    const int quad = lane >> 4;
    const float NEG_INF = -__builtin_inff();
    const int n_chunks = KT / BKT;
    const int64_t n_rounds = (n_super + total_waves - 1) / total_waves;  // the same for every wave: lockstep

    // corpus prefetch cursor (as in dense_scan_kernel); past the end (and for idle waves) re-read a valid group
    const int64_t safe_group = wave % n_super;
    int64_t pf_group = wave;
    int pf_pair = 0, pf_kt = 0;
    chunk_t ring[PF][RS];
    auto prefetch = [&](int slot) {
        const int64_t grp = pf_group < n_super ? pf_group : safe_group;
        const int64_t rb = grp * kRowBlocksPerSuper + pf_pair * RS;
#pragma unroll
        for (int s = 0; s < RS; ++s)
            ring[slot][s] = __builtin_nontemporal_load(tiles + ((rb + s) * KT + pf_kt) * kTileChunks + lane);
        const bool wrap_kt = (pf_kt + 1 == KT);
        pf_kt = wrap_kt ? 0 : pf_kt + 1;
        const bool wrap_pair = wrap_kt && (pf_pair + 1 == kPairs);
        pf_pair = wrap_kt ? (wrap_pair ? 0 : pf_pair + 1) : pf_pair;
        pf_group = wrap_pair ? pf_group + total_waves : pf_group;
    };
    // query chunk staging: wave `wid` moves fragments wid*kStage .. +kStage of a chunk (f = g*BKT + kk)
    auto q_src = [&](int chunk, int f) {
        const int g = f / BKT, kk = f % BKT;
        return qfrag + ((int64_t)g * KT + chunk * BKT + kk) * kTileChunks + lane;
    };
    // prologue: chunk 0 into buffer 0
#pragma unroll
    for (int i = 0; i < kStage; ++i) {
        const int f = wid * kStage + i;
        lds_q[f * kTileChunks + lane] = *q_src(0, f);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) prefetch(j);
    __syncthreads();
    int cur = 0;

    for (int64_t round = 0; round < n_rounds; ++round) {
        const int64_t group = round * total_waves + wave;
        const bool live = group < n_super;
        float m[GQ];
#pragma unroll
        for (int g = 0; g < GQ; ++g) m[g] = NEG_INF;
        const bool tail = live && ((group + 1) * kSuperRows > n_rows || rowmask != nullptr);
#pragma unroll 1
        for (int pair = 0; pair < kPairs; ++pair) {
            f32x4_t acc[RS][GQ];
#pragma unroll
            for (int s = 0; s < RS; ++s)
#pragma unroll
                for (int g = 0; g < GQ; ++g) acc[s][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            // one query chunk: stage the NEXT chunk (wrapping to chunk 0 for the next pair / round) into
            // registers, run this chunk's MFMAs out of LDS buffer `cur`, then publish the staged chunk.
            // SB = first ring slot of the chunk (compile-time: the ring holds two chunks).
            auto chunk_body = [&](auto SB, int kc) {
                const int kn = (kc + 1 == n_chunks) ? 0 : kc + 1;
                chunk_t stage[kStage];
#pragma unroll
                for (int i = 0; i < kStage; ++i) stage[i] = *q_src(kn, wid * kStage + i);
                const chunk_t* qb = lds_q + cur * kFrags * kTileChunks;
#pragma unroll
                for (int j = 0; j < BKT; ++j) {
                    // query fragments one group ahead of their MFMAs; the scheduling barriers stop hipcc from
                    // hoisting all GQ fragment reads (64 VGPRs at GQ = 16) above the MFMA chain, which spills
                    chunk_t bn = qb[(0 * BKT + j) * kTileChunks + lane];
#pragma unroll
                    for (int g = 0; g < GQ; ++g) {
                        const chunk_t b = bn;
                        if (g + 1 < GQ) bn = qb[((g + 1) * BKT + j) * kTileChunks + lane];
#pragma unroll
                        for (int s = 0; s < RS; ++s) Mfma<STORE>::run(ring[decltype(SB)::value + j][s], b, acc[s][g]);
                        if (GQ > 8) __builtin_amdgcn_sched_barrier(0);
                    }
                    prefetch(decltype(SB)::value + j);
                }
                chunk_t* qn = lds_q + (cur ^ 1) * kFrags * kTileChunks;
#pragma unroll
                for (int i = 0; i < kStage; ++i) qn[(wid * kStage + i) * kTileChunks + lane] = stage[i];
                __syncthreads();
                cur ^= 1;
            };
#pragma unroll 1
            for (int kc = 0; kc < n_chunks; kc += 2) {  // KT is a multiple of 4 -> n_chunks is even
                chunk_body(std::integral_constant<int, 0>{}, kc);
                chunk_body(std::integral_constant<int, BKT>{}, kc + 1);
            }
            if (live) {
#pragma unroll
                for (int s = 0; s < RS; ++s) {
                    const int64_t row0 = (group * kRowBlocksPerSuper + pair * RS + s) * kRowsPerBlock + quad * 4;
                    const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(scale + row0);
                    float ok[4] = {1.f, 1.f, 1.f, 1.f};
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int64_t row = row0 + r;
                            bool v = row < n_rows;
                            if (v && rowmask) v = (rowmask[row >> 3] >> (row & 7)) & 1;
                            ok[r] = v ? 1.f : 0.f;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GQ; ++g) {
                        float mr = NEG_INF;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[s][g][r] * sc[r];
                            v = (ok[r] != 0.f) ? v : NEG_INF;
                            mr = fmaxf(mr, v);
                        }
                        if (NRB == 1) {
                            mr = col4_max(mr);
                            if (lane < 16 && 16 * g + lane < nq)
                                gmax[(int64_t)(16 * g + lane) * gmax_stride + group * kRowBlocksPerSuper + pair * RS + s] = mr;
                        } else {
                            m[g] = fmaxf(m[g], mr);
                        }
                    }
                }
            }
        }
        if (NRB != 1 && live) {
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                float v = m[g];
                v = col4_max(v);
                if (lane < 16 && 16 * g + lane < nq) gmax[(int64_t)(16 * g + lane) * gmax_stride + group] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 256-query scan, queries stationary in REGISTERS (fp16 shards with KT = 24 tiles per row, D = 768).
//
// dense_scan_bigq_kernel keeps the corpus in registers and re-streams the query tile from L2
// through LDS every round; that extra traffic shares the CU's vector-memory path with the
// corpus stream, and a 256-query form of it gains nothing.  Here the
// roles are swapped:
//   * wave w of the block owns 32 queries (GW = 2 groups of 16) and holds ALL their B fragments
//     in registers (2 x 24 x 4 VGPRs) for the whole kernel: 8 waves = 256 queries per pass;
//   * the block streams its corpus tiles HBM -> LDS with global_load_lds_dwordx4 (no VGPRs,
//     one 1 KiB tile per wave instruction) into a ring of 64 tiles; every wave reads every tile
//     once from LDS (ds_read_b128, conflict-free) and issues GW MFMAs per tile;
//   * tiles move in stages of one row block (KT = 24 tiles, three per wave): per stage a wave waits for
//     its own tiles of that stage (counted s_waitcnt, the loads of three newer stages stay in flight),
//     the block meets at an LDS-only barrier, the slot set everybody has just finished with is
//     refilled, and the 24 tiles are consumed four at a time.  Nothing but corpus bytes crosses the
//     vector-memory path.
// The 64 row scales of the NEXT super-group travel the same way (one 256-byte LDS-DMA load by wave 0,
// a whole super-group ahead): an ordinary register load inside the loop would make the compiler wait
// for it with vmcnt(0), i.e. drain the ring.  Epilogue and output (per-group maxima) as in the other scans.
// Measured at 10M x 768: 3.77 ms per 256 queries (4.09 TB/s) against 2 x 2.6 ms for two 128-query passes (stages of
// 8 tiles, i.e. three barriers per row block: 4.0 ms; a stage's refill loads issued in one burst: 3.82 ms).  The DMA stream alone (no LDS reads, no MFMAs) runs at
// 5.06 TB/s with nontemporal loads (4.3 TB/s without the hint; ring depth changes nothing), and the LDS reads
// (8 waves x 1 KiB per tile, ~50 B/clk/CU) are what the rest of the time goes to.  A 128-query form (4 waves x 32 queries, two
// blocks per CU) measures 2.63 ms alone and the same step time as dense_scan_bigq_kernel inside the pipeline, so
// batches up to 128 queries keep that kernel (any D, fp32 too) and only the 8-wave form is instantiated.
constexpr int kQregStages = 5;        // ring depth in stages of one row block (KT tiles): 5 x 24 KiB of LDS, 4 stages in flight

typedef __attribute__((address_space(1))) const void* hr_gptr_t;
typedef __attribute__((address_space(3))) void* hr_lptr_t;

template <int KT, int NRB, int GW, int NW>  // NW waves x GW groups of 16 queries per pass
__global__ __launch_bounds__(64 * NW) void dense_scan_qreg_kernel(
    const chunk_t* __restrict__ tiles, const chunk_t* __restrict__ qfrag, const float* __restrict__ scale,
    const uint8_t* __restrict__ rowmask, float* __restrict__ gmax, int nq, int64_t n_rows, int64_t n_super) {
    constexpr int T = KT, NS = kQregStages;       // one stage = the KT tiles of one row block (one barrier per row block)
    constexpr int kRing = NS * T;
    constexpr int L = T / NW;                     // tiles a wave loads per stage
    constexpr int TPS = kRowBlocksPerSuper * KT;  // tiles of one 64-row super-group, contiguous in the shard
    constexpr int H = 4;                          // tiles whose LDS reads are issued / awaited together
    static_assert(KT % T == 0 && T % NW == 0 && T % H == 0 && (T / H) % L == 0, "whole stages per row block, the same number of tiles per wave and stage");
    static_assert(NRB == 1 || NRB == kRowBlocksPerSuper, "group = one row block or one super-group");
    __shared__ chunk_t ring[kRing * kTileChunks];
    __shared__ f32x4_t sc_lds[2][kSuperRows / 4];  // row scales of the current / next super-group
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int quad = lane >> 4;
    const float NEG_INF = -__builtin_inff();
    const unsigned ring_lds = (unsigned)(uintptr_t)(hr_lptr_t)ring;        // LDS byte offsets for the asm reads
    const unsigned sc_lds_addr = (unsigned)(uintptr_t)(hr_lptr_t)&sc_lds[0][0];
    const int64_t first = blockIdx.x, step = gridDim.x;
    if (first >= n_super) return;  // whole block
    const int64_t n_my = (n_super - first + step - 1) / step;
    const int64_t gmax_stride = n_super * (kRowBlocksPerSuper / NRB);

    chunk_t qf[GW][KT];  // this wave's 16 * GW queries, every k-step
#pragma unroll
    for (int gq = 0; gq < GW; ++gq)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            qf[gq][kt] = qfrag[((int64_t)(wid * GW + gq) * KT + kt) * kTileChunks + lane];
    // make the compiler settle these loads HERE (it waits before this "use"); otherwise it places an
    // s_waitcnt vmcnt(0) in front of the first MFMA inside the loop, which would drain the ring every row block
#pragma unroll
    for (int gq = 0; gq < GW; ++gq)
#pragma unroll
        for (int kt = 0; kt < KT; kt += 8)
            asm volatile("" : "+v"(qf[gq][kt]), "+v"(qf[gq][kt + 1]), "+v"(qf[gq][kt + 2]), "+v"(qf[gq][kt + 3]),
                              "+v"(qf[gq][kt + 4]), "+v"(qf[gq][kt + 5]), "+v"(qf[gq][kt + 6]), "+v"(qf[gq][kt + 7]));

    // loader: wave w brings tile w of every stage
    int64_t ld_sg = first;
    int ld_within = wid;
    // tile wid + l * NW of the stage the loader is at (all L of them lie inside the same super-group)
    auto issue_one = [&](int stage_slot, int l) {
        const int64_t sg = ld_sg < n_super ? ld_sg : first;  // past the end: harmless re-read, keeps the count of loads in flight fixed
        const chunk_t* src = tiles + (sg * TPS + ld_within + l * NW) * kTileChunks + lane;
        __builtin_amdgcn_global_load_lds((hr_gptr_t)src, (hr_lptr_t)(ring + (stage_slot * T + wid + l * NW) * kTileChunks), 16,
                                         0, 2 /* nt: each byte is read once */);
    };
    auto advance_loader = [&]() {
        ld_within += T;
        if (ld_within >= TPS) {
            ld_within -= TPS;
            ld_sg += step;
        }
    };
    auto issue = [&](int stage_slot) {
#pragma unroll
        for (int l = 0; l < L; ++l) issue_one(stage_slot, l);
        advance_loader();
    };
    auto issue_scale = [&](int64_t sg, int slot) {  // wave 0 only: one more (older) operation on its counter
        const int64_t sgc = sg < n_super ? sg : first;
        __builtin_amdgcn_global_load_lds((hr_gptr_t)(scale + sgc * kSuperRows + lane), (hr_lptr_t)&sc_lds[slot][0], 4, 0, 0);
    };
    if (wid == 0) issue_scale(first, 0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);

    int st = 0;  // stage slot of the stage about to be consumed
    int64_t sg = first;
    for (int64_t g = 0; g < n_my; ++g, sg += step) {
        float m[GW];
#pragma unroll
        for (int gq = 0; gq < GW; ++gq) m[gq] = NEG_INF;
        if (wid == 0) issue_scale(sg + step, (int)((g + 1) & 1));  // lands >= 7 stages before its first use
        const bool tail = (sg + 1) * kSuperRows > n_rows || rowmask != nullptr;
#pragma unroll 1
        for (int rbi = 0; rbi < kRowBlocksPerSuper; ++rbi) {
            f32x4_t acc[GW][2];
#pragma unroll
            for (int gq = 0; gq < GW; ++gq) acc[gq][0] = acc[gq][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ph = 0; ph < KT / T; ++ph) {
                // own tiles of this stage have landed: at least (NS - 2) * L newer loads were issued after them
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * L) : "memory");
                lds_barrier();
                const int refill = st == 0 ? NS - 1 : st - 1;  // the slots of the stage everybody has just left
                // LDS reads as inline asm: a ds_read the compiler can see after an LDS-DMA load makes it insert
                // s_waitcnt vmcnt(0) ("may alias the DMA destination"), which would drain the ring every stage
                const unsigned addr = ring_lds + (unsigned)(st * T * kTileChunks + lane) * 16u;
                chunk_t a[H];
#pragma unroll
                for (int half = 0; half < T / H; ++half) {  // GW = 2 leaves registers for four tiles at a time
#pragma unroll
                    for (int j = 0; j < H; ++j)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(addr), "n"((half * H + j) * 1024) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
#pragma unroll
                    for (int j = 0; j < H; ++j)
#pragma unroll
                        for (int gq = 0; gq < GW; ++gq)
                            Mfma<_Float16>::run(a[j], qf[gq][ph * T + half * H + j], acc[gq][j & 1]);
                    // the stage's L refill loads are spread over the stage instead of issued in one burst
                    if ((half + 1) % ((T / H) / L) == 0) issue_one(refill, (half + 1) / ((T / H) / L) - 1);
                }
                advance_loader();
                st = (st + 1 == NS) ? 0 : st + 1;
            }
            // epilogue of the row block: lane holds rows quad*4..+3 of the block for query (lane & 15) of each group
            float ok[4] = {1.f, 1.f, 1.f, 1.f};
            const int64_t row0 = (sg * kRowBlocksPerSuper + rbi) * kRowsPerBlock + quad * 4;
            f32x4_t sc;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(sc)
                         : "v"(sc_lds_addr + (unsigned)(((g & 1) * (kSuperRows / 4) + rbi * (kRowsPerBlock / 4) + quad) * 16))
                         : "memory");
            if (tail) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + r;
                    bool v = row < n_rows;
                    if (v && rowmask) v = (rowmask[row >> 3] >> (row & 7)) & 1;
                    ok[r] = v ? 1.f : 0.f;
                }
            }
#pragma unroll
            for (int gq = 0; gq < GW; ++gq) {
                const int q = 16 * (wid * GW + gq) + (lane & 15);
                float mr = NEG_INF;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = (acc[gq][0][r] + acc[gq][1][r]) * sc[r];
                    v = (ok[r] != 0.f) ? v : NEG_INF;
                    mr = fmaxf(mr, v);
                }
                if (NRB == 1) {
                    mr = col4_max(mr);
                    if (lane < 16 && q < nq) gmax[(int64_t)q * gmax_stride + sg * kRowBlocksPerSuper + rbi] = mr;
                } else {
                    m[gq] = fmaxf(m[gq], mr);
                }
            }
        }
        if (NRB != 1) {
#pragma unroll
            for (int gq = 0; gq < GW; ++gq) {
                const int q = 16 * (wid * GW + gq) + (lane & 15);
                float v = m[gq];
                v = col4_max(v);
                if (lane < 16 && q < nq) gmax[(int64_t)q * gmax_stride + sg] = v;
            }
        }
    }
    // LDS DMA still in flight must land before the block's LDS is handed to another block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// 256-query scan, second form (round 4): dense_scan_qreg_kernel with HALF the LDS reads — 4 waves x 64 queries, ONE wave
// per SIMD, the 4 x KT query fragments of a wave spread over the unified 512-entry register file (groups 0, 1 in the
// accumulator half, groups 2, 3 in the vector half: v_mfma takes A / B / C from either), the corpus streamed HBM -> LDS by
// LDS-DMA exactly as there (stages of one row block, ring of kQregStages, one barrier per stage).  Every 1 KiB tile is read
// from LDS by 4 waves instead of 8 and feeds 4 MFMAs per read instead of 2.  (A first form of this kernel streamed the
// corpus into registers and let the four waves share it through the caches: the unique bytes in flight per compute unit
// are then one wave's ring — 4 KiB — and the pass ran at 2.4 TB/s.)
//
// hipcc, left to itself, copies the AGPR-resident query fragments to VGPRs before every MFMA (and, with 384 query
// registers, sees no room for a pipeline), so the k loop is written out: LDS reads, counted waits and MFMAs as inline asm.
// What the asm owes the hardware: an accumulator is touched every 4th MFMA (no back-to-back dependence), its first MFMA of
// a row block takes the constant 0 as C, and the epilogue's reads sit behind s_nop (the compiler's hazard recogniser
// does not look into asm).
template <bool FIRST>
__device__ inline void q64_mfma_a(f32x4_t& acc, const chunk_t& tile, const chunk_t& q) {   // query fragment in an AGPR
    if (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(tile), "a"(q));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(tile), "a"(q));
}
template <bool FIRST>
__device__ inline void q64_mfma_v(f32x4_t& acc, const chunk_t& tile, const chunk_t& q) {   // query fragment in a VGPR
    if (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(tile), "v"(q));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(tile), "v"(q));
}

template <int KT, int NRB>
__global__ __launch_bounds__(256, 1) void dense_scan_q64_kernel(
    const chunk_t* __restrict__ tiles, const chunk_t* __restrict__ qfrag, const float* __restrict__ scale,
    const uint8_t* __restrict__ rowmask, float* __restrict__ gmax, int nq, int64_t n_rows, int64_t n_super) {
    constexpr int GW = 4, NW = 4, T = KT, NS = kQregStages;
    constexpr int L = T / NW;                     // tiles a wave loads per stage
    constexpr int TPS = kRowBlocksPerSuper * KT;  // tiles of one 64-row super-group, contiguous in the shard
    constexpr int H = 4;                          // tiles whose LDS reads are issued / awaited together
    static_assert(T % NW == 0 && T % H == 0 && (T / H) == L, "one refill tile per group of four");
    static_assert(NRB == 1 || NRB == kRowBlocksPerSuper, "group = one row block or one super-group");
    extern __shared__ chunk_t q64_lds[];          // ring [NS * T tiles], then the row scales of the current / next super-group
    chunk_t* ring = q64_lds;
    float* sc_lds = reinterpret_cast<float*>(q64_lds + NS * T * kTileChunks);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int quad = lane >> 4;
    const float NEG_INF = -__builtin_inff();
    const unsigned ring_lds = (unsigned)(uintptr_t)(hr_lptr_t)ring;
    const unsigned sc_lds_addr = (unsigned)(uintptr_t)(hr_lptr_t)sc_lds;
    const int64_t first = blockIdx.x, step = gridDim.x;
    if (first >= n_super) return;  // whole block
    const int64_t n_my = (n_super - first + step - 1) / step;
    const int64_t gmax_stride = n_super * (kRowBlocksPerSuper / NRB);

    chunk_t qa[2][KT], qv[2][KT];   // this wave's 64 queries, every k-step: groups 0, 1 (AGPRs) and 2, 3 (VGPRs)
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            qa[gq][kt] = qfrag[((int64_t)(wid * GW + gq) * KT + kt) * kTileChunks + lane];
            qv[gq][kt] = qfrag[((int64_t)(wid * GW + 2 + gq) * KT + kt) * kTileChunks + lane];
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int gq = 0; gq < 2; ++gq)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) asm volatile("" : "+a"(qa[gq][kt]), "+v"(qv[gq][kt]));   // settled, and where they belong

    // loader (as dense_scan_qreg_kernel): wave w brings tiles w, w + NW, ... of every stage
    int64_t ld_sg = first;
    int ld_within = wid;
    auto issue_one = [&](int stage_slot, int l) {
        const int64_t sg = ld_sg < n_super ? ld_sg : first;  // past the end: harmless re-read, keeps the count of loads in flight fixed
        const chunk_t* src = tiles + (sg * TPS + ld_within + l * NW) * kTileChunks + lane;
        __builtin_amdgcn_global_load_lds((hr_gptr_t)src, (hr_lptr_t)(ring + (stage_slot * T + wid + l * NW) * kTileChunks), 16,
                                         0, 2 /* nt: each byte is read once */);
    };
    auto advance_loader = [&]() {
        ld_within += T;
        if (ld_within >= TPS) {
            ld_within -= TPS;
            ld_sg += step;
        }
    };
    auto issue_scale = [&](int64_t sg, int slot) {  // wave 0 only: one more (older) operation on its counter
        const int64_t sgc = sg < n_super ? sg : first;
        __builtin_amdgcn_global_load_lds((hr_gptr_t)(scale + sgc * kSuperRows + lane), (hr_lptr_t)(sc_lds + slot * kSuperRows), 4, 0, 0);
    };
    if (wid == 0) issue_scale(first, 0);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int l = 0; l < L; ++l) issue_one(s, l);
        advance_loader();
    }

    int st = 0;  // stage slot of the stage about to be consumed
    int64_t sg = first;
    for (int64_t g = 0; g < n_my; ++g, sg += step) {
        float m[GW];
#pragma unroll
        for (int gq = 0; gq < GW; ++gq) m[gq] = NEG_INF;
        if (wid == 0) issue_scale(sg + step, (int)((g + 1) & 1));  // lands >= NS - 1 stages before its first use
        const bool tail = (sg + 1) * kSuperRows > n_rows || rowmask != nullptr;
#pragma unroll 1
        for (int rbi = 0; rbi < kRowBlocksPerSuper; ++rbi) {
            f32x4_t acc[GW];
            // own tiles of this stage have landed: at least (NS - 2) * L newer loads were issued after them
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * L) : "memory");
            lds_barrier();
            const int refill = st == 0 ? NS - 1 : st - 1;  // the slots of the stage everybody has just left
            const unsigned addr = ring_lds + (unsigned)(st * T * kTileChunks + lane) * 16u;
            chunk_t a[2][H];
            auto read4 = [&](chunk_t (&d)[H], int half) {
                asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                             : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                             : "v"(addr), "n"((half * H) * 1024), "n"((half * H + 1) * 1024), "n"((half * H + 2) * 1024), "n"((half * H + 3) * 1024)
                             : "memory");
            };
            read4(a[0], 0);
#pragma unroll
            for (int half = 0; half < T / H; ++half) {
                chunk_t (&cur)[H] = a[half & 1];
                if (half + 1 < T / H) {
                    read4(a[(half + 1) & 1], half + 1);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
                }
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    const int kt = half * H + j;
                    if (kt == 0) {
                        q64_mfma_a<true>(acc[0], cur[j], qa[0][kt]);
                        q64_mfma_a<true>(acc[1], cur[j], qa[1][kt]);
                        q64_mfma_v<true>(acc[2], cur[j], qv[0][kt]);
                        q64_mfma_v<true>(acc[3], cur[j], qv[1][kt]);
                    } else {
                        q64_mfma_a<false>(acc[0], cur[j], qa[0][kt]);
                        q64_mfma_a<false>(acc[1], cur[j], qa[1][kt]);
                        q64_mfma_v<false>(acc[2], cur[j], qv[0][kt]);
                        q64_mfma_v<false>(acc[3], cur[j], qv[1][kt]);
                    }
                }
                issue_one(refill, half);   // one refill tile per group of four: the stage's L loads spread over the stage
            }
            advance_loader();
            st = (st + 1 == NS) ? 0 : st + 1;
            // the accumulators are complete 16 cycles after the last MFMA issued; the compiler does not know they came from MFMAs
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
            // epilogue of the row block: lane holds rows quad*4..+3 of the block for query (lane & 15) of each group
            float ok[4] = {1.f, 1.f, 1.f, 1.f};
            const int64_t row0 = (sg * kRowBlocksPerSuper + rbi) * kRowsPerBlock + quad * 4;
            f32x4_t sc;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(sc)
                         : "v"(sc_lds_addr + (unsigned)(((g & 1) * kSuperRows + rbi * kRowsPerBlock + quad * 4) * 4))
                         : "memory");
            if (tail) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + r;
                    bool v = row < n_rows;
                    if (v && rowmask) v = (rowmask[row >> 3] >> (row & 7)) & 1;
                    ok[r] = v ? 1.f : 0.f;
                }
            }
#pragma unroll
            for (int gq = 0; gq < GW; ++gq) {
                const int q = 16 * (wid * GW + gq) + (lane & 15);
                float mr = NEG_INF;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[gq][r] * sc[r];
                    v = (ok[r] != 0.f) ? v : NEG_INF;
                    mr = fmaxf(mr, v);
                }
                if (NRB == 1) {
                    mr = col4_max(mr);
                    if (lane < 16 && q < nq) gmax[(int64_t)q * gmax_stride + sg * kRowBlocksPerSuper + rbi] = mr;
                } else {
                    m[gq] = fmaxf(m[gq], mr);
                }
            }
        }
        if (NRB != 1) {
#pragma unroll
            for (int gq = 0; gq < GW; ++gq) {
                const int q = 16 * (wid * GW + gq) + (lane & 15);
                float v = m[gq];
                v = col4_max(v);
                if (lane < 16 && q < nq) gmax[(int64_t)q * gmax_stride + sg] = v;
            }
        }
    }
    // LDS DMA still in flight must land before the block's LDS is handed to another block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// 256-query scan as a tiled contraction (any KT >= 8, fp16 shards): BASELINE config 5's shape (D = 1024, B = 256).
// (GQ = 8 is the same kernel for 65..128 queries: 64 accumulator registers per wave, half the query ring.)
//
// 256 queries x 1024 dims of fp16 are 512 KiB — as much as a CU's whole register file — so for this shape the queries
// can be stationary neither in registers (dense_scan_qreg_kernel: D = 768 only) nor in LDS; two 128-query passes read
// the corpus twice (0.37 of the HBM peak per step however fast each pass is).  Here BOTH operands stream through LDS
// and the block keeps a 256-row x 256-query accumulator tile in registers, the structure of a dense GEMM:
//   * block = 8 waves as 2 (row halves) x 4 (query quarters): a wave owns 128 rows x 64 queries = 8 x 4 MFMA tiles of
//     16 x 16, 128 accumulator registers, for the whole k loop of a row tile;
//   * per k-step (one 1 KiB tile = 32 dims) the block needs 16 corpus tiles (HBM) and 16 query fragments (L2: the
//     fragment set is re-read once per 256 rows, 1 byte of L2 traffic per byte of HBM traffic);
//   * waves 0-3 (row half 0) are the corpus loaders, waves 4-7 (row half 1) the query loaders (global_load_lds_dwordx4,
//     4 pieces of 1 KiB per wave and step): the two streams need different run-ahead (HBM: 5 steps, L2: 2) and a wave's
//     vmcnt retires in issue order, so one wave must not carry both; every wave computes;
//   * the two waves of a SIMD (wave w and w + 4) run half a step apart ("ping-pong"): while one issues its 32 MFMAs of
//     a step, the other does everything else — 12 fragment reads (inline asm, see dense_scan_qreg_kernel), their wait,
//     its four refill pieces, the landed-check of its own loads.  Two LDS-only barriers per step
//     keep the halves in anti-phase.  With one barrier per step all eight waves wanted the matrix pipe at the same time
//     and left it idle at the same time (stamp build: 40 % busy, the older wave of a SIMD waiting 600 cycles per step at
//     the barrier for the younger one);
//   * the fragment registers are single-buffered (the reads of a step do not overlap the wave's own MFMAs, they overlap
//     the other wave's);
//   * epilogue per row tile as in the other scans (scale, mask, maximum per candidate group), row scales by LDS-DMA.
// MFMA time per step is 2 waves x 32 x 16 cycles per SIMD for 16 KiB of corpus: the matrix pipe, not HBM, is the
// nearer bound (SURVEY §7 "batch = 256 is a GEMM"); bench.py reports both fractions.
#ifdef HR_STAMP  // diagnostic build only (make stamp): where a k-step of dense_scan_gemm_kernel spends its cycles
__device__ unsigned long long hr_gemm_stamps[2][8];
#define GEMM_STAMP(i)                                                                                   \
    do {                                                                                                \
        unsigned long long t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        seg[i] += t_ - t_last;                                                                          \
        t_last = t_;                                                                                    \
    } while (0)
#else
#define GEMM_STAMP(i)
#endif
template <bool B> struct BoolConst { static constexpr bool value = B; };
constexpr int kGemmRowBlocks = 16;               // row blocks per block tile (256 rows)
#ifndef HR_GEMM_DA
#define HR_GEMM_DA 6
#define HR_GEMM_DB 3
#endif
constexpr int kGemmDA = HR_GEMM_DA, kGemmDB = HR_GEMM_DB;  // ring depth (k-steps) of the corpus / query stream
#ifndef HR_GEMM_DA128
#define HR_GEMM_DA128 7
#endif
#ifndef HR_GEMM_READ_BURST
#define HR_GEMM_READ_BURST 3
#endif
constexpr int kGemmReadBurst = HR_GEMM_READ_BURST;  // next-step fragment reads issued together between the MFMAs
constexpr int kGemmDA128 = HR_GEMM_DA128;        // corpus ring depth of the 128-query form (its query ring is half the size)

template <int GQ, int NRB>                       // GQ query groups of 16 per pass: 16 (256 queries) or 8 (128)
__global__ __launch_bounds__(512) void dense_scan_gemm_kernel(
    const chunk_t* __restrict__ tiles, const chunk_t* __restrict__ qfrag, const float* __restrict__ scale,
    const uint8_t* __restrict__ rowmask, float* __restrict__ gmax, int nq, int KT, int64_t n_rows, int64_t n_super) {
    static_assert(NRB == 1 || NRB == kRowBlocksPerSuper, "group = one row block or one super-group");
    static_assert(GQ == 16 || GQ == 8, "four query quarters of 4 or 2 groups");
    constexpr int DA = GQ == 16 ? kGemmDA : kGemmDA128, DB = kGemmDB, RB = kGemmRowBlocks;
    constexpr int WA = RB / 2, WB = GQ / 4;      // fragments a wave reads per step: its row blocks, its query groups
    constexpr int PB = GQ / 4;                   // 1 KiB pieces a query loader moves per step (a corpus loader: 4)
    __shared__ chunk_t ringA[DA * RB * kTileChunks];
    __shared__ chunk_t ringB[DB * GQ * kTileChunks];
    __shared__ f32x4_t sc_lds[2][RB * kRowsPerBlock / 4];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wid >> 2, wq = wid & 3;       // row half, query quarter this wave computes
    const bool loads_a = wid < 4;
    const int lw = wid & 3;                      // loader index within its stream
    const int quad = lane >> 4;
    const float NEG_INF = -__builtin_inff();
    const unsigned a_lds = (unsigned)(uintptr_t)(hr_lptr_t)ringA, b_lds = (unsigned)(uintptr_t)(hr_lptr_t)ringB;
    const unsigned sc_addr = (unsigned)(uintptr_t)(hr_lptr_t)&sc_lds[0][0];
    const int64_t n_rb = n_super * kRowBlocksPerSuper;                  // row blocks that exist in the shard
    const int64_t n_tiles = (n_rb + RB - 1) / RB;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first >= n_tiles) return;  // whole block
    const int64_t my_tiles = (n_tiles - first + stride - 1) / stride;
    const int total_steps = (int)(my_tiles * KT);  // rows < 2^31 and 256 rows per tile: at most 2^23 tiles x 128 k-steps
    const int64_t gmax_stride = n_super * (kRowBlocksPerSuper / NRB);

    // ---- loaders.  One piece of code for both roles (a role is data: source, ring, depth), so that the pieces can sit
    // between the MFMAs of a step without control flow.  Everything but the lane's 16 bytes is wave-uniform: `src` is the
    // wave's piece 0 of the loader's current step and advances by 1 KiB per step; pieces 1-3 lie 4 row blocks / query
    // groups further each (rel[]).  Row blocks past the shard (last tile only; row blocks come in fours) are replaced by
    // piece 0 — harmless re-reads, the epilogue masks the rows.  Past the block's last tile the corpus loader wraps to
    // its first one, which keeps the number of loads in flight fixed.
    static_assert(RB == 16, "four pieces per step and corpus loader wave");
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned pstride = (unsigned)KT * 4096u;   // 4 row blocks / query groups further, same k-step
    const int depth = loads_a ? DA : DB;
    const int slot_tiles = loads_a ? RB : GQ;
    chunk_t* const ring_w = (loads_a ? ringA : ringB) + lw * kTileChunks;
    const char* src;
    unsigned rel[4] = {0u, pstride, 2u * pstride, 3u * pstride};
    int l_kt = 0, l_slot = 0, l_sc = 0;
    int64_t l_tile = first, l_tiles_left = my_tiles;
    auto tile_setup = [&]() {  // corpus loaders: first addresses of a row tile
        src = reinterpret_cast<const char*>(tiles) + (unsigned long long)(l_tile * RB + lw) * (unsigned long long)KT * 1024ull;
        const int64_t fours = (n_rb - l_tile * RB) >> 2;  // >= 1
#pragma unroll
        for (int l = 1; l < 4; ++l) rel[l] = l < fours ? (unsigned)l * pstride : 0u;
    };
    if (loads_a) tile_setup();
    else src = reinterpret_cast<const char*>(qfrag) + (unsigned long long)lw * (unsigned long long)KT * 1024ull;
    auto piece = [&](int l) {  // 1 KiB of the loader's current step into its ring slot
        if (l < PB || loads_a)
            __builtin_amdgcn_global_load_lds((hr_gptr_t)(src + rel[l] + lane16),
                                             (hr_lptr_t)(ring_w + (l_slot * slot_tiles + 4 * l) * kTileChunks), 16, 0, 0);
    };
    auto advance = [&]() {
        if (loads_a && l_kt == 0) {  // the tile's 256 row scales (64 per loader wave), needed KT steps from now
            int64_t row = l_tile * (RB * kRowsPerBlock) + lw * 64 + lane;
            if (row >= n_rb * kRowsPerBlock) row = 0;
            __builtin_amdgcn_global_load_lds((hr_gptr_t)(scale + row),
                                             (hr_lptr_t)(reinterpret_cast<float*>(&sc_lds[l_sc][0]) + lw * 64), 4, 0, 0);
        }
        l_slot = l_slot + 1 == depth ? 0 : l_slot + 1;
        src += 1024;
        if (++l_kt == KT) {
            l_kt = 0;
            if (loads_a) {
                l_sc ^= 1;
                l_tile = --l_tiles_left > 0 ? l_tile + stride : first;
                tile_setup();
            } else {
                src -= (unsigned long long)KT * 1024ull;  // the query fragments again, for the next row tile
            }
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int l = 0; l < 4; ++l) piece(l);
        advance();
    };
    // fragments of one step: this wave's WA corpus tiles and WB query fragments (inline asm, see dense_scan_qreg_kernel)
    auto read_frags = [&](int sa, int sb, chunk_t (&a)[WA], chunk_t (&b)[WB]) {
        const unsigned aa = a_lds + (unsigned)(((sa * RB) + wr * WA) * kTileChunks + lane) * 16u;
        const unsigned ba = b_lds + (unsigned)(((sb * GQ) + wq * WB) * kTileChunks + lane) * 16u;
#pragma unroll
        for (int g = 0; g < WB; ++g)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[g]) : "v"(ba), "n"(g * 1024) : "memory");
#pragma unroll
        for (int r = 0; r < WA; ++r)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[r]) : "v"(aa), "n"(r * 1024) : "memory");
    };
    auto settle = [&](chunk_t (&a)[WA], chunk_t (&b)[WB]) {  // the reads above have returned
        __builtin_amdgcn_sched_barrier(0);  // the MFMAs issued before this point stay there: they are what the wait hides behind
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])::"memory");
#pragma unroll
        for (int g = 0; g < WB; ++g) asm volatile("" : "+v"(b[g]));
        __builtin_amdgcn_sched_barrier(0);
    };
    // Own loads of the step that is read after the next barrier have landed.  Leading half, after issuing step t + DA - 1:
    // step t + 1, so DA - 2 steps may still fly; trailing half, having issued step t + DB: step t + 2, DB - 2 steps.
    auto wait_landed = [&]() {
        if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 2) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DB - 2) * PB) : "memory");
    };

    // Run-ahead before the first step: after its reads of step t the leading half (corpus loaders) refills the slot of
    // step t - 1, the trailing half (query loaders) the slot of step t — at that point both halves have read it.
#pragma unroll 1
    for (int s = 0; s < (loads_a ? DA - 1 : DB); ++s) issue();
    f32x4_t acc[WA][WB];     // written by the first k-step of every row tile
    chunk_t fa0[WA], fb0[WB], fa1[WA], fb1[WB];  // fragments of the step being multiplied / of the next one

    int kt = 0, sa = 0, sb = 0;  // k-tile and ring slots of the step being multiplied
    int64_t tile = first;
    int sci = 0;
#ifdef HR_STAMP
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memtime();
#endif
    // Epilogue of a row tile.  The accumulators are not cleared here: the first k-step of a tile multiplies into zero.
    // tail_c: the tile reaches past the shard's rows or a row mask is set (the lane's four rows of a row block are four
    // neighbouring mask bits: one byte load per row block, all issued before the one wait).
    auto epilogue = [&](auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        GEMM_STAMP(7);
        // lane holds rows quad*4..+3 of each of its WA row blocks for query (lane & 15) of each of its WB groups
        const int64_t rb0 = tile * RB + wr * WA;   // first row block of this wave
        f32x4_t sc[WA];
        const unsigned sc_a =
            sc_addr + (unsigned)((sci * (RB * kRowsPerBlock / 4) + wr * WA * (kRowsPerBlock / 4) + quad) * 16);
#pragma unroll
        for (int r = 0; r < WA; ++r)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sc[r]) : "v"(sc_a), "n"(r * (kRowsPerBlock / 4) * 16) : "memory");
        unsigned ok[WA];
        if (TAIL) {
#pragma unroll
            for (int r = 0; r < WA; ++r) {
                const int64_t row0 = (rb0 + r) * kRowsPerBlock + quad * 4;
                const int64_t left = n_rows - row0;  // how many of the lane's four rows exist
                unsigned bits = left >= 4 ? 0xFu : left > 0 ? (1u << (int)left) - 1u : 0u;
                if (rowmask != nullptr && left > 0) bits &= (unsigned)rowmask[row0 >> 3] >> (unsigned)(row0 & 7);
                ok[r] = bits;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]), "+v"(sc[6]), "+v"(sc[7])::"memory");
        float m[2][WB];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int g = 0; g < WB; ++g) m[h2][g] = NEG_INF;
#pragma unroll
        for (int r = 0; r < WA; ++r) {
#pragma unroll
            for (int g = 0; g < WB; ++g) {
                f32x4_t v = acc[r][g] * sc[r];
                if (TAIL) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = ((ok[r] >> x) & 1u) ? v[x] : NEG_INF;
                }
                float mr = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                if (NRB == 1) {
                    mr = col4_max(mr);
                    const int q = 16 * (wq * WB + g) + (lane & 15);
                    if (lane < 16 && q < nq && rb0 + r < n_rb) gmax[(int64_t)q * gmax_stride + rb0 + r] = mr;
                } else {
                    m[r / kRowBlocksPerSuper][g] = fmaxf(m[r / kRowBlocksPerSuper][g], mr);
                }
            }
        }
        if (NRB != 1) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int64_t sg = rb0 / kRowBlocksPerSuper + h2;
#pragma unroll
                for (int g = 0; g < WB; ++g) {
                    float v = m[h2][g];
                    v = col4_max(v);
                    const int q = 16 * (wq * WB + g) + (lane & 15);
                    if (lane < 16 && q < nq && sg < n_super) gmax[(int64_t)q * gmax_stride + sg] = v;
                }
            }
        }
        tile += stride;
        sci ^= 1;
        GEMM_STAMP(3);  // epilogue of a row tile
    };
    // Barrier sequence (every wave passes every barrier): B2(-1) | B1(0) B2(0) | B1(1) B2(1) | ...
    //   leading half:   [B2(s-1)] reads of step s returned, refill, own loads of step s+1 landed [B1(s)]
    //                             MFMAs of step s with the reads of step s+1 between them [B2(s)]
    //   trailing half:  [B1(s)]   reads of step s returned, refill [B2(s)]
    //                             MFMAs of step s with the reads of step s+1 between them, own loads of step s+2 landed [B1(s+1)]
    // B1(s) publishes "every piece of step s+1 has landed"; a ring slot is refilled only after both halves have waited
    // for their reads of it (corpus slot of step s-1 in the leading half's read phase of step s, query slot of step s in
    // the trailing half's).
    auto m_phase = [&](const bool ZERO, chunk_t (&ca)[WA], chunk_t (&cb)[WB], chunk_t (&na)[WA], chunk_t (&nb)[WB], int nsa,
                       int nsb) __attribute__((always_inline)) {
        // ZERO (a constant at both call sites): first k-step of a row tile, C = 0 instead of clearing 128 registers
        const unsigned aa = a_lds + (unsigned)(((nsa * RB) + wr * WA) * kTileChunks + lane) * 16u;
        const unsigned ba = b_lds + (unsigned)(((nsb * GQ) + wq * WB) * kTileChunks + lane) * 16u;
        // The next step's WA + WB fragments go into the other register set in bursts of kGemmReadBurst reads after every
        // NM * kGemmReadBurst / NR MFMAs (inline asm, see dense_scan_qreg_kernel).  Measured at 12.5M x 1024: bursts of
        // 1-2 reads per 4 MFMAs 6.17 ms, single reads spread evenly (one per 2-3 MFMAs) 6.68 ms.
        constexpr int NR = WA + WB, NM = WA * WB, NB = (NR + kGemmReadBurst - 1) / kGemmReadBurst;  // NB bursts
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int r = i / WB, g = i % WB;
            if (ZERO) acc[r][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            Mfma<_Float16>::run(ca[r], cb[g], acc[r][g]);
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if ((k + 1) * NM / NB - 1 == i) {  // burst k follows this MFMA
#pragma unroll
                    for (int j = k * kGemmReadBurst; j < (k + 1) * kGemmReadBurst && j < NR; ++j) {
                        if (j < WB) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nb[j]) : "v"(ba), "n"(j * 1024) : "memory");
                        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(na[j - WB]) : "v"(aa), "n"((j - WB) * 1024) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // One k-step of a wave (a macro, not a lambda: hipcc keeps the accumulators of a lambda that calls the epilogue
    // lambda in scratch).  ca/cb: fragments of this step; na/nb: where the next step's go.
#define GEMM_HALF_STEP(ca, cb, na, nb)                                                                                  \
    do {                                                                                                                \
        settle(ca, cb);                                                                                                 \
        GEMM_STAMP(0); /* the reads of this step, issued a half-step ago, have returned */                              \
        /* the refill belongs to this half-step, not between the MFMAs: a piece holds its wave for ~60-100 cycles */    \
        /* wherever it is issued, and here the SIMD's other wave has the matrix pipe */                                 \
        piece(0);                                                                                                       \
        piece(1);                                                                                                       \
        piece(2);                                                                                                       \
        piece(3);                                                                                                       \
        advance();                                                                                                      \
        if (loads_a) wait_landed();                                                                                     \
        GEMM_STAMP(1); /* refill issue (leading half: + own loads of the next step) */                                  \
        lds_barrier();                                                                                                  \
        GEMM_STAMP(2); /* barrier before the MFMAs */                                                                   \
        const int nsa = sa + 1 == DA ? 0 : sa + 1, nsb = sb + 1 == DB ? 0 : sb + 1;                                     \
        if (__builtin_expect(kt == 0, 0)) m_phase(true, ca, cb, na, nb, nsa, nsb);                                      \
        else m_phase(false, ca, cb, na, nb, nsa, nsb);                                                                  \
        GEMM_STAMP(4); /* 32 MFMAs + 12 reads (issue) */                                                                \
        sa = nsa;                                                                                                       \
        sb = nsb;                                                                                                       \
        if (__builtin_expect(++kt == KT, 0)) { /* out of line: the k loop stays one contiguous run of code */           \
            kt = 0;                                                                                                     \
            if ((tile * RB + wr * WA + WA) * kRowsPerBlock > n_rows || rowmask != nullptr) epilogue(BoolConst<true>{}); \
            else epilogue(BoolConst<false>{});                                                                          \
        }                                                                                                               \
        GEMM_STAMP(7); /* bookkeeping after the MFMAs */                                                                \
        if (!loads_a) wait_landed();                                                                                    \
        GEMM_STAMP(5); /* trailing half: own loads of the step after next */                                            \
        lds_barrier();                                                                                                  \
        GEMM_STAMP(6); /* barrier after the MFMAs */                                                                    \
    } while (0)
    // steps 0 and 1 have landed (the leading half has issued one step less than wait_landed assumes)
    if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 3) * 4) : "memory");
    else wait_landed();
    lds_barrier();                 // B2(-1)
    read_frags(0, 0, fa0, fb0);
    if (!loads_a) lds_barrier();   // B1(0): the trailing half starts half a step later
#pragma unroll 1
    for (int S = 0; S < total_steps; S += 2) {  // KT is even, so is the number of steps
        GEMM_HALF_STEP(fa0, fb0, fa1, fb1);
        GEMM_HALF_STEP(fa1, fb1, fa0, fb0);
    }
#undef GEMM_HALF_STEP
    if (loads_a) lds_barrier();    // the trailing half's last B1
#ifdef HR_STAMP
    if (blockIdx.x == 7 && lane == 0 && (wid == 0 || wid == 5))
        for (int i = 0; i < 8; ++i) hr_gemm_stamps[wid == 0 ? 0 : 1][i] = seg[i];
#endif
    // LDS DMA still in flight must land before the block's LDS is handed to another block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// Canonical refine: 64 candidate rows per wave (group_rows = 16 or 64 rows per candidate group); lane = row.
// score = (float) S with S the k-ordered fp64 sum of exact products.  The same
// arithmetic is restated in oracle/oracle.c:dense_score().
// One candidate row slot of query qi: returns false (invalid) or the canonical score and the row.
template <typename STORE>
__device__ inline bool refine_dense_slot(const chunk_t* __restrict__ tiles, int KT, int dim, const float* __restrict__ qq,
                                         double qn2_q, const double* __restrict__ norm2,
                                         const uint8_t* __restrict__ rowmask, const int32_t* cand_q, int group_rows,
                                         int64_t n_rows, int cosine, int slot, float* score, int32_t* row_out) {
    constexpr int EPC = kChunkBytes / (int)sizeof(STORE);
    const int32_t group = cand_q[slot / group_rows];
    const int64_t row = (int64_t)group * group_rows + slot % group_rows;
    bool valid = group >= 0 && row < n_rows;
    if (valid && rowmask) valid = (rowmask[row >> 3] >> (row & 7)) & 1;
    if (!valid) return false;
    double s = 0.0;
    // The fp64 add chain is serial by definition (canonical k order); the loads are not: one whole 1 KiB-tile row
    // slice (4 chunks) x 2 tiles per round trip, and the NEXT round's 8 chunks are requested before this round's are
    // consumed (the fused finishing kernel runs 8 waves per compute unit: nothing else hides the round trip).
    // (the next round is requested in two halves, each right after the half of the current round it replaces has been
    // consumed: 8 to 12 chunks in flight on 48 registers instead of 64)
    constexpr int U = 8, H = U / 2;
    const int n_kc = KT * 4;   // a multiple of 16
    union Chunk { chunk_t v; STORE e[EPC]; };
    Chunk lo[H], hi[H], nlo[H];
    auto consume = [&](const Chunk (&c)[H], int kc_first) {
#pragma unroll
        for (int u = 0; u < H; ++u) {
#pragma unroll
            for (int j = 0; j < EPC; ++j) {
                const int k = (kc_first + u) * EPC + j;
                if (k < dim) s = __dadd_rn(s, __dmul_rn((double)c[u].e[j], (double)qq[k]));
            }
        }
    };
#pragma unroll
    for (int u = 0; u < H; ++u) lo[u].v = tiles[chunk_index(row, u, KT)];
#pragma unroll
    for (int u = 0; u < H; ++u) hi[u].v = tiles[chunk_index(row, H + u, KT)];
    for (int kc0 = 0; kc0 < n_kc; kc0 += U) {
        const int kn = kc0 + U < n_kc ? kc0 + U : kc0;   // the last round re-reads itself (no branch around the loads)
#pragma unroll
        for (int u = 0; u < H; ++u) nlo[u].v = tiles[chunk_index(row, kn + u, KT)];
        __builtin_amdgcn_sched_barrier(0);   // keep the compiler from hoisting all 16 loads above the first use
        consume(lo, kc0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < H; ++u) lo[u].v = tiles[chunk_index(row, kn + H + u, KT)];   // next round's upper half, parked in lo
        __builtin_amdgcn_sched_barrier(0);
        consume(hi, kc0 + H);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < H; ++u) {
            hi[u].v = lo[u].v;
            lo[u].v = nlo[u].v;
        }
    }
    if (cosine) {
        double d = norm2[row] * qn2_q;
        s = (d > 0.0) ? s / sqrt(d) : 0.0;
    }
    *score = (float)s;
    *row_out = (int32_t)row;
    return true;
}

template <typename STORE>
__global__ __launch_bounds__(64) void refine_dense_kernel(
    const chunk_t* __restrict__ tiles, int KT, int dim, const float* __restrict__ q,
    const double* __restrict__ qn2, const double* __restrict__ norm2,
    const uint8_t* __restrict__ rowmask, const int32_t* __restrict__ cand, int C, int group_rows,
    int64_t n_rows, int cosine, float* __restrict__ out_score, int32_t* __restrict__ out_row) {
    const int qi = blockIdx.y, lane = threadIdx.x;
    const int slot = blockIdx.x * 64 + lane;          // candidate row slot of this query
    const int n_slots = C * group_rows;
    if (slot >= n_slots) return;
    const int64_t o = (int64_t)qi * n_slots + slot;
    float sc = -__builtin_inff();
    int32_t row = -1;
    if (!refine_dense_slot<STORE>(tiles, KT, dim, q + (int64_t)qi * dim, qn2[qi], norm2, rowmask, cand + (int64_t)qi * C,
                                  group_rows, n_rows, cosine, slot, &sc, &row)) {
        sc = -__builtin_inff();
        row = -1;
    }
    out_score[o] = sc;
    out_row[o] = row;
}

}  // namespace hbmrag
