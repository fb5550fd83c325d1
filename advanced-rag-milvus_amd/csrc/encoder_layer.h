// The GEMM-shaped half of an encoder / cross-encoder layer as hand-written MFMA kernels (round 4): every linear map of a
// post-LN BERT layer (reference hooks: retrieval.py:651-685 `CrossEncoder.predict`, indexing.py:610-620 `encode_semantic`)
// with the elementwise work that follows it folded in, so that a layer is three launches —
//
//   linear_rows_kernel     qkv = x W_qkv^T + b                                  (written in the layout the attention reads)
//   attention_kernel       a   = softmax(q k^T / sqrt(d)) v                      (attention.h)
//   encoder_tail_kernel    x'  = LN2(x1 + W_down gelu(W_up x1 + b_up) + b_down),  x1 = LN1(x + W_out a + b_out)
//
// — instead of four vendor GEMMs, an attention and two add + LayerNorm passes, with the [tokens, 4H] FFN intermediate,
// the output projection's result and x1 never leaving the compute unit (HBM traffic of a layer ~8 KB per token at
// H = 384 instead of ~20 KB).
//
// TOKEN-STATIONARY, WEIGHT-STREAMING.  The products are computed transposed, y^T = W x^T: the weight matrix is the MFMA A
// operand (16 output features x 32 inputs per v_mfma_f32_16x16x32_f16), the activations are the B operand (32 inputs x 16
// tokens), and a result tile has its TOKEN in the lane column and four output features in the lane's registers.  A wave
// owns 16 * TT tokens for the whole kernel and keeps ALL their activations in registers (H = 384, TT = 2: 96 registers of
// B fragments + 192 of fp32 accumulators; one wave per SIMD, the unified 512-register file is what makes this possible);
// the four waves of a block (64 * TT tokens) share one stream of weights, brought HBM/L2 -> LDS by LDS-DMA in 1 KiB
// pieces = one A fragment in register-image order (lane l's 16 bytes at l * 16: conflict-free ds_read_b128, and exactly
// what global_load_lds writes), in stages of SP = H / 16 pieces through a ring of NS stages with ONE workgroup barrier per
// stage (counted vmcnt: the DMA of NS - 2 later stages stays in flight across it).  Every weight byte is read from LDS
// once per wave and feeds TT MFMAs: 1 KiB / TT per MFMA per wave — half of what a 256 x 256 tiled GEMM moves through LDS
// per MFMA (128 x 64 outputs per wave: 384 B / MFMA for both operands) at TT = 2, and no activation ever goes through LDS.
//
// CHAINING WITHOUT LANE MOVEMENT.  A result tile's layout (column = token, rows 4g .. 4g+3 in the lane's registers) is, two
// tiles at a time, the B-operand layout of the NEXT product up to a permutation of the k index: element j of lane group g
// is input feature 32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4) instead of 32 s + 8 g + j.  The weights of a product that
// consumes an accumulator (W_up, W_down) are packed on the host with that k permutation (advanced_rag/encoder_kernels.py),
// so LN1's output feeds W_up and gelu's output feeds W_down straight from registers — the same trick attention.h plays
// with its probabilities.  LayerNorm statistics are per token = per lane column: a sum over the lane's registers and the
// four lanes l, l ^ 16, l ^ 32, l ^ 48 (col4_sum: two row swaps).
//
// FRAGMENT-ORDER ACTIVATIONS ("FR").  Between the launches of a layer the activations travel in the order the MFMA wants
// them: X_fr[tile = row / 16][s][lane = 16 g + col][j] = X[16 tile + col][32 s + 16 (j >> 2) + 4 g + (j & 3)] — 1 KiB per
// (16-row tile, k-step), which is at the same time the accumulator layout of two neighbouring output tiles and the B operand
// of the next product (accumulator k order).  A wave then loads / stores its rows with 16-byte accesses that are
// contiguous over the wave (1 KiB per instruction) instead of 64-byte (B fragment) or 8-byte (accumulator) pieces 768
// bytes apart: the stamped build showed 18 % of a block's time in those loads and stores.  Row-major stays available per
// operand (flags): the embedding layer writes it, pooling / heads read it.
//
// SKEWED FFN.  The intermediate is produced 32 features (one k-step of W_down) at a time: chunk c = two tiles of
// W_up x1 (+ bias as the accumulator's initial value) -> gelu -> fp16 B fragment -> 2 H / 16 MFMAs of W_down.  The weight
// stream interleaves the stages as  up(0) | up(1) down(0) | up(2) down(1) | ... | down(last), so the vector work of
// gelu(c) has the MFMAs of up(c + 1) beside it — none of which depend on it — instead of sitting between two dependent
// MFMA groups.  (Round 4, second form: the vector work of a chunk's activation is split over the two stages that follow
// its up-projection — see the loop.)
#pragma once
#include "common.h"

namespace hbmrag {

typedef __attribute__((address_space(1))) const void* el_gptr_t;
typedef __attribute__((address_space(3))) void* el_lptr_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

#ifndef EL_ABLATE
#define EL_ABLATE 0     // timing-only probe builds (tests/probes/el_probe.hip); the library is built with 0
#endif
constexpr int kElWaves = 4;        // waves per block, one per SIMD
constexpr int kElRingStages = 4;   // LDS ring depth in stages

// ---- LDS access the compiler must not see (a ds_read it can see after an LDS-DMA makes it wait vmcnt(0)) -------------
__device__ inline void el_ds_read4(chunk_t (&a)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(addr) : "memory");   // early clobber: the address
}                                                                                                 // outlives the first read
template <int N>
__device__ inline void el_wait_lgkm(chunk_t (&a)[4]) {   // the four fragments of a group are usable after this
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}
__device__ inline f32x4_t el_lds_f4(unsigned addr) {     // four floats of a parameter table, synchronously
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ inline half8_t el_as_half8(const chunk_t& c) { return __builtin_bit_cast(half8_t, c); }

// tanh-form GELU as x * sigmoid(2u), 2u = x (c0 + c1 x^2): one exp2, one reciprocal, four multiply-adds
__device__ inline float el_gelu_tanh(float v) {
    constexpr float kC0 = -1.5957691216057308f * 1.4426950408889634f;              // -2 sqrt(2/pi) log2(e)
    constexpr float kC1 = -1.5957691216057308f * 0.044715f * 1.4426950408889634f;
    const float z = v * __builtin_fmaf(v * v, kC1, kC0);
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
__device__ inline float el_gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.7071067811865476f)); }


// This lane's B fragments of 16 rows x (32 KS) features, fragment order or row-major (-> accumulator k order either way
// when `acc_order`, natural k order otherwise; fragment-order buffers are always in accumulator k order).
template <int KS>
__device__ inline void el_load_rows(half8_t (&f)[KS], const _Float16* base, bool fr, bool acc_order, int64_t row0, int64_t M,
                                    int col, int g, int lane) {
    if (fr) {
        const int64_t tile = min(row0, M - 1) >> 4;      // a tile past the end re-reads the last one (its rows are never stored)
        const half8_t* src = reinterpret_cast<const half8_t*>(base) + tile * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) f[s] = src[s * 64];
    } else {
        const _Float16* r = base + min(row0 + col, M - 1) * (32 * KS);
        if (acc_order) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half4_t lo = *reinterpret_cast<const half4_t*>(r + 32 * s + 4 * g);
                const half4_t hi = *reinterpret_cast<const half4_t*>(r + 32 * s + 16 + 4 * g);
                f[s] = (half8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const half8_t*>(r + 32 * s + 8 * g);
        }
    }
}

// The weight stream of a block: `stage` pieces per stage, stage s in ring slot s % NS.  Wave w issues pieces w, w + 4, ...
template <int SP, int NS>
struct ElStream {
    static constexpr int L = SP / kElWaves;   // DMA instructions per wave and stage
    static_assert(SP % kElWaves == 0 && SP % 4 == 0, "whole pieces per wave, groups of four");
    const chunk_t* stream0;
    const chunk_t* src;      // this lane's address inside piece `wid` of the next stage to issue
    const chunk_t* src_last; // the same inside the LAST stage (issues past the end re-read it: the counts stay fixed)
    chunk_t* ring;
    int issue_slot, read_slot;
    int wid;

    __device__ inline void init(const chunk_t* stream, int64_t n_stages, chunk_t* ring_, int wid_, int lane) {
        stream0 = src = stream + (int64_t)wid_ * kTileChunks + lane;
        src_last = src + (n_stages - 1) * SP * kTileChunks;
        ring = ring_;
        wid = wid_;
        issue_slot = read_slot = 0;
    }
    __device__ inline void issue_piece(int l) {     // piece wid + 4 l of the next stage to issue
        if ((EL_ABLATE & 1) && src > stream0 + (int64_t)(NS - 1) * SP * kTileChunks) return;
        const chunk_t* s = src <= src_last ? src : src_last;
        __builtin_amdgcn_global_load_lds((el_gptr_t)(s + l * kElWaves * kTileChunks),
                                         (el_lptr_t)(ring + (issue_slot * SP + wid + l * kElWaves) * kTileChunks), 16, 0, 0);
    }
    __device__ inline void issued() {
        src += (int64_t)SP * kTileChunks;
        issue_slot = issue_slot + 1 == NS ? 0 : issue_slot + 1;
    }
    __device__ inline void issue() {
#pragma unroll
        for (int l = 0; l < L; ++l) issue_piece(l);
        issued();
    }
    // Top of a stage: my own pieces of it have landed (the DMA of the NS - 2 stages issued after it may still fly) and the
    // block meets (everybody's pieces landed; everybody is done with the stage before, whose slot consume() refills).
    // -> LDS byte address of this lane's 16 bytes of piece 0 of the stage
    __device__ inline unsigned begin(int lane) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * L) : "memory");
        if (!(EL_ABLATE & 4)) lds_barrier();
        const unsigned addr = (unsigned)(uintptr_t)(el_lptr_t)(ring + (read_slot * SP) * kTileChunks + lane);
        read_slot = read_slot + 1 == NS ? 0 : read_slot + 1;
        return addr;
    }
    // The SP pieces of the stage, four at a time, the next four requested before the current four are used; after the
    // MFMAs of group l this wave's l-th refill piece is issued (one LDS-DMA per 4 TT MFMAs instead of a burst of L behind
    // the barrier, where nothing hides their issue cost) and side(l) gets its turn: vector work that does not depend on
    // this stage's MFMAs and is meant to issue beside them.
    template <class F, class S>
    __device__ inline void consume(unsigned addr, F&& f, S&& side) {
        static_assert(L == SP / 4, "one refill piece per group of four");
        constexpr int NGR = SP / 4;
        chunk_t a[3][4];   // groups gi + 1 and gi + 2 are requested before group gi is used
        if (EL_ABLATE & 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[0][j] = a[1][j] = a[2][j] = (chunk_t){addr, addr + j, 0x3c003c00u, 0x38003800u};
        } else {
            el_ds_read4(a[0], addr);
            if (NGR > 1) el_ds_read4(a[1], addr + 4096);
        }
#pragma unroll
        for (int gi = 0; gi < NGR; ++gi) {
            if (!(EL_ABLATE & 2)) {
                if (gi + 2 < NGR) {
                    el_ds_read4(a[(gi + 2) % 3], addr + (gi + 2) * 4096);
                    el_wait_lgkm<8>(a[gi % 3]);
                } else if (gi + 1 < NGR) {
                    el_wait_lgkm<4>(a[gi % 3]);
                } else {
                    el_wait_lgkm<0>(a[gi % 3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) f(gi * 4 + j, el_as_half8(a[gi % 3][j]));
            issue_piece(gi);
            side(gi);
        }
        issued();
    }
    template <class F>
    __device__ inline void consume(unsigned addr, F&& f) {
        consume(addr, f, [](int) {});
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// out[M][N] (row stride out_stride) = x[M][K] W^T + bias, K = 32 KS, N a multiple of 32.  Pieces ordered [t][s]; a stage =
// two output tiles = 32 output features.  The ROWS of W are packed in "store order": row r of tile 2 so + u is output
// feature 32 so + 8 (r >> 2) + 4 u + (r & 3), so that the eight results a lane holds after a stage (rows 4 g .. 4 g + 3 of
// both tiles) are eight CONSECUTIVE features 32 so + 8 g .. + 7 of its row: one 16-byte store per token tile and stage
// instead of two 8-byte ones 32 bytes apart.  bias: fp32 [N] in the same order ([so][u][r]).
struct LinearArgs {
    const _Float16* x;
    const chunk_t* w;       // natural k order for row-major x, accumulator k order for fragment-order x
    const float* bias;
    _Float16* out;          // row-major
    int64_t M, out_stride;
    int N;
    int x_fr;               // x in fragment order
};

template <int KS, int TT>
__global__ __launch_bounds__(64 * kElWaves, 1) void linear_rows_kernel(LinearArgs p) {
    constexpr int SP = 2 * KS, NS = kElRingStages + 1;
    extern __shared__ chunk_t el_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, g = lane >> 4;
    const int64_t tok0 = (int64_t)blockIdx.x * (64 * TT) + wid * (16 * TT);
    const int n_stages = p.N / 32;
    ElStream<SP, NS> st;
    st.init(p.w, n_stages, el_lds, wid, lane);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) st.issue();

    float* tbl = reinterpret_cast<float*>(el_lds + NS * SP * kTileChunks);
    const unsigned tbl_lds = (unsigned)(uintptr_t)(el_lptr_t)tbl;
    for (int i = threadIdx.x; i < p.N / 4; i += 64 * kElWaves)
        reinterpret_cast<f32x4_t*>(tbl)[i] = reinterpret_cast<const f32x4_t*>(p.bias)[i];
    half8_t xf[TT][KS];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) el_load_rows<KS>(xf[tt], p.x, p.x_fr != 0, false, tok0 + 16 * tt, p.M, col, g, lane);
    // settle the loads here: a wait the compiler places inside the loop would be vmcnt(0) and drain the ring every stage
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[tt][s]));

    for (int so = 0; so < n_stages; ++so) {
        // bias = the accumulators' initial value, from the LDS copy (a global load in here would make the compiler drain
        // the ring with vmcnt(0) every stage)
        f32x4_t acc[2][TT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4_t b = el_lds_f4(tbl_lds + 4u * (32 * so + 16 * u + 4 * g));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[u][tt] = b;
        }
        const unsigned addr = st.begin(lane);
        st.consume(addr, [&](int pi, half8_t a) {
            const int u = pi / KS, s = pi % KS;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[tt][s], acc[u][tt], 0, 0, 0);
        });
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int64_t row = tok0 + 16 * tt + col;
            if (row < p.M) {
                half8_t w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (_Float16)acc[j >> 2][tt][j & 3];
                *reinterpret_cast<half8_t*>(p.out + row * p.out_stride + 32 * so + 8 * g) = w;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMA still in flight must land before the LDS is handed on
}

// ---------------------------------------------------------------------------------------------------------------------
// x' = LN2(x1 + W_down gelu(W_up x1 + b_up) + b_down),  x1 = LN1(x + W_out a + b_out)   for H = 32 HS, I = 32 IS.
// wstream: stages of SP = 2 HS pieces —  out-projection: HS stages (natural k order, two output tiles each);  then
// up(0) | up(1) down(0) | ... | up(IS-1) down(IS-2) | down(IS-1)   with up(c) = tiles 2c, 2c+1 of W_up and down(c) = k-step
// c of every output tile of W_down, both with the accumulator k permutation (header).
// tables (fp32): b_out[H] g1[H] be1[H] b_down[H] g2[H] be2[H] b_up[I].
struct TailArgs {
    const _Float16* a;     // attention output, FRAGMENT ORDER (hr_attention_*_dev with out_fr)
    const _Float16* x;     // layer input (the residual of LN1): fragment order if x_fr, else row-major [M][H]
    _Float16* out;         // fragment order if out_fr, else row-major [M][H]
    int x_fr, out_fr;
    const chunk_t* wstream;
    const float* tables;
    int64_t M;
    float eps;
#ifdef EL_STAMP
    unsigned long long* stamps;   // [blocks][8] s_memtime at the phase boundaries (probe builds only)
#endif
};
#ifdef EL_STAMP
#define EL_MARK(i) do { if (threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EL_MARK(i) do { } while (0)
#endif

template <int HS, int IS, int TT, bool ERF>
__global__ __launch_bounds__(64 * kElWaves, 1) void encoder_tail_kernel(TailArgs p) {
    constexpr int H = 32 * HS, I = 32 * IS, HT = 2 * HS, SP = 2 * HS, NS = kElRingStages;
    static_assert(IS % 2 == 0, "the chunk loop is unrolled by two");
    extern __shared__ chunk_t el_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, g = lane >> 4;
    const int64_t tok0 = (int64_t)blockIdx.x * (64 * TT) + wid * (16 * TT);
    float* tbl = reinterpret_cast<float*>(el_lds + NS * SP * kTileChunks);
    const unsigned tbl_lds = (unsigned)(uintptr_t)(el_lptr_t)tbl;
    constexpr unsigned kBout = 0, kG1 = 4u * H, kBe1 = 8u * H, kBdown = 12u * H, kG2 = 16u * H, kBe2 = 20u * H, kBup = 24u * H;  // byte offsets

    EL_MARK(0);
    ElStream<SP, NS> st;
    st.init(p.wstream, HS + 2 * IS, el_lds, wid, lane);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) st.issue();

    // parameter tables -> LDS; this wave's attention rows (B fragments, natural k order) and residual rows (accumulator layout)
    for (int i = threadIdx.x; i < (6 * H + I) / 4; i += 64 * kElWaves)
        reinterpret_cast<f32x4_t*>(tbl)[i] = reinterpret_cast<const f32x4_t*>(p.tables)[i];
    half8_t af[TT][HS], res[TT][HS];   // both in accumulator k order: element j = output tile 2 s + (j >> 2), row 4 g + (j & 3)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        el_load_rows<HS>(af[tt], p.a, true, true, tok0 + 16 * tt, p.M, col, g, lane);
        el_load_rows<HS>(res[tt], p.x, p.x_fr != 0, true, tok0 + 16 * tt, p.M, col, g, lane);
    }
    __syncthreads();   // tables visible; every load and the first DMA stages have landed (vmcnt(0) is part of it)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int s = 0; s < HS; ++s) asm volatile("" : "+v"(af[tt][s]), "+v"(res[tt][s]));

    EL_MARK(1);
    // ---- out-projection: o^T = W_out a^T, initial value = residual + bias ------------------------------------------
    f32x4_t o[HT][TT];
#pragma unroll
    for (int so = 0; so < HS; ++so) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4_t b = el_lds_f4(tbl_lds + kBout + 4u * (16 * (2 * so + u) + 4 * g));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[2 * so + u][tt][r] = (float)res[tt][so][4 * u + r] + b[r];
        }
        const unsigned addr = st.begin(lane);
        st.consume(addr, [&](int pi, half8_t a) {
            const int u = pi / HS, s = pi % HS;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                o[2 * so + u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, af[tt][s], o[2 * so + u][tt], 0, 0, 0);
        });
    }

    EL_MARK(2);
    // ---- LayerNorm over the lane column; -> fp16 B fragments of the next product (two tiles per k-step) -------------
    half8_t xf[TT][HS];
    auto layer_norm = [&](f32x4_t (&v)[HT][TT], unsigned g_off, unsigned b_off, float (&mean)[TT], float (&rstd)[TT]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < HT; ++t) s += (v[t][tt][0] + v[t][tt][1]) + (v[t][tt][2] + v[t][tt][3]);
            mean[tt] = col4_sum(s) * (1.0f / H);
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = v[t][tt][r] - mean[tt];
                    q = __builtin_fmaf(d, d, q);
                }
            rstd[tt] = __builtin_amdgcn_rsqf(col4_sum(q) * (1.0f / H) + p.eps);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            const f32x4_t ga = el_lds_f4(tbl_lds + g_off + 4u * (16 * t + 4 * g));
            const f32x4_t be = el_lds_f4(tbl_lds + b_off + 4u * (16 * t + 4 * g));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t][tt][r] = __builtin_fmaf((v[t][tt][r] - mean[tt]) * rstd[tt], ga[r], be[r]);
        }
    };
    {
        float mean[TT], rstd[TT];
        layer_norm(o, kG1, kBe1, mean, rstd);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int s = 0; s < HS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[tt][s][j] = (_Float16)o[2 * s + (j >> 2)][tt][j & 3];
    }

    EL_MARK(3);
    // ---- FFN: y^T = x1 + b_down + W_down gelu(W_up x1 + b_up), 32 intermediate features at a time, skewed ------------
    f32x4_t y[HT][TT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f32x4_t b = el_lds_f4(tbl_lds + kBdown + 4u * (16 * t + 4 * g));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][tt][r] = (float)xf[tt][t >> 1][4 * (t & 1) + r] + b[r];
    }
    constexpr int NG = SP / 4;       // groups of four pieces per stage
    constexpr int EP = 4 * TT;       // activation values per lane handled beside ONE stage (half of a chunk's 8 TT)
    // value e (0 .. 8 TT - 1) of a chunk: token tile e / 8, B-fragment element e % 8 = accumulator tile (e % 8) / 4, row e % 4
    auto act_one = [&](const f32x4_t (&h)[2][TT], half8_t (&hf)[TT], int e) {
        const float v = h[(e & 7) >> 2][e >> 3][e & 3];
        hf[e >> 3][e & 7] = (_Float16)((EL_ABLATE & 8) ? v : ERF ? el_gelu_erf(v) : el_gelu_tanh(v));
    };
    auto act_part = [&](const f32x4_t (&h)[2][TT], half8_t (&hf)[TT], int part, int gi) {   // the share of group gi
#pragma unroll
        for (int i = 0; i < EP; ++i)
            if (i * NG / EP == gi) act_one(h, hf, part * EP + i);
    };
    auto act_all = [&](const f32x4_t (&h)[2][TT], half8_t (&hf)[TT], int part) {
#pragma unroll
        for (int i = 0; i < EP; ++i) act_one(h, hf, part * EP + i);
    };
    auto up_stage = [&](int c, f32x4_t (&h)[2][TT], auto&& side) {     // h = b_up + W_up[32 c .. 32 c + 31] x1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4_t b = el_lds_f4(tbl_lds + kBup + 4u * (32 * c + 16 * u + 4 * g));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) h[u][tt] = b;
        }
        const unsigned addr = st.begin(lane);
        st.consume(addr, [&](int pi, half8_t a) {
            const int u = pi / HS, s = pi % HS;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) h[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[tt][s], h[u][tt], 0, 0, 0);
        }, side);
    };
    auto down_stage = [&](const half8_t (&hf)[TT], auto&& side) {      // y += W_down[:, chunk] gelu(h)
        const unsigned addr = st.begin(lane);
        st.consume(addr, [&](int pi, half8_t a) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) y[pi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hf[tt], y[pi][tt], 0, 0, 0);
        }, side);
    };
    // Stage order (= the stream's): up(0) up(1) | down(c) up(c+2) down(c+1) up(c+3) | ... | down(IS-2) down(IS-1).  The
    // activation of chunk c+1 is computed half beside down(c) and half beside up(c+2): neither depends on it.
    f32x4_t hA[2][TT], hB[2][TT];
    half8_t hfA[TT], hfB[TT];
    auto none = [](int) {};
    up_stage(0, hA, none);
    act_all(hA, hfA, 0);
    act_all(hA, hfA, 1);
    up_stage(1, hB, none);
#pragma unroll 1
    for (int c = 0; c + 2 < IS; c += 2) {
        down_stage(hfA, [&](int gi) { act_part(hB, hfB, 0, gi); });
        up_stage(c + 2, hA, [&](int gi) { act_part(hB, hfB, 1, gi); });
        down_stage(hfB, [&](int gi) { act_part(hA, hfA, 0, gi); });
        up_stage(c + 3, hB, [&](int gi) { act_part(hA, hfA, 1, gi); });
    }
    down_stage(hfA, [&](int gi) { act_part(hB, hfB, 0, gi); });
    act_all(hB, hfB, 1);
    down_stage(hfB, none);

    EL_MARK(4);
    // ---- LN2 and out -------------------------------------------------------------------------------------------------
    {
        float mean[TT], rstd[TT];
        layer_norm(y, kG2, kBe2, mean, rstd);
    }
    EL_MARK(5);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int64_t row = tok0 + 16 * tt + col;
        if (row < p.M) {
            if (p.out_fr) {     // 16 bytes per (k-step, lane): a wave's store is 1 KiB contiguous
                half8_t* dst = reinterpret_cast<half8_t*>(p.out) + ((tok0 + 16 * tt) >> 4) * HS * 64 + lane;
#pragma unroll
                for (int s = 0; s < HS; ++s) {
                    half8_t w;
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (_Float16)y[2 * s + (j >> 2)][tt][j & 3];
                    dst[s * 64] = w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    half4_t w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (_Float16)y[t][tt][r];
                    *reinterpret_cast<half4_t*>(p.out + row * H + 16 * t + 4 * g) = w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EL_MARK(6);
}

}  // namespace hbmrag
