// Shared device helpers for libhbmrag (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hbmrag.h"

namespace hbmrag {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int chunk_t __attribute__((ext_vector_type(4)));  // one 16-byte chunk

constexpr int kWave = 64;
constexpr int kRowsPerBlock = 16;   // rows of one MFMA A-tile ("row block")
constexpr int kRowBlocksPerSuper = 4;  // row blocks a scan wave takes per step of its grid-stride loop
constexpr int kSuperRows = kRowsPerBlock * kRowBlocksPerSuper;  // 64 rows
// Candidate-group size (rows whose scan maximum is kept as one value) is a per-shard
// setting: 16 (one row block) or 64 (one super-group).  Smaller groups mean 4x less
// refine traffic per query at the price of 4x more group maxima written by the scan.
constexpr int kChunkBytes = 16;     // bytes one lane contributes to a tile
constexpr int kTileChunks = 64;     // chunks (lanes) per 1 KiB tile

// Order-preserving map float -> uint32 (larger float <=> larger key).
__host__ __device__ inline uint32_t ord_f32(float f) {
    union { float f; uint32_t u; } c; c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float unord_f32(uint32_t k) {
    union { float f; uint32_t u; } c;
    c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return c.f;
}
// Unique 64-bit ranking key: (score desc, row asc)  <=>  key desc.
__host__ __device__ inline uint64_t rank_key(float score, uint32_t row) {
    return ((uint64_t)ord_f32(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__host__ __device__ inline float key_score(uint64_t key) { return unord_f32((uint32_t)(key >> 32)); }
__host__ __device__ inline uint32_t key_row(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu); }

// Tiled shard layout.  A row block holds 16 rows; along k it is cut into
// tiles of 4 chunks (16 B each: 8 halfs or 4 floats).  Tile (rb, kt) is 1 KiB,
// lane l of a wave owns chunk l: row rb*16 + (l & 15), k-chunk kt*4 + (l >> 4).
// That is exactly the A-operand register image of v_mfma_f32_16x16x32_f16
// (and, element j at a time, of v_mfma_f32_16x16x4_f32), so the scan streams
// the shard with one fully coalesced 16 B/lane load per MFMA step.
__host__ __device__ inline int64_t chunk_index(int64_t row, int kchunk, int KT) {
    int64_t rb = row >> 4;
    int rr = (int)(row & 15);
    int kt = kchunk >> 2, c = kchunk & 3;
    return (rb * KT + kt) * kTileChunks + rr + 16 * c;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + s_barrier and the
// fence may drain every outstanding memory operation (s_waitcnt vmcnt(0)); a kernel that keeps
// global loads in flight across the barrier wants to wait for its own LDS/scalar operations only.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts: six vector instructions and no
// LDS round trip (each __shfl_up step is a ds_bpermute, ~100 cycles on the critical path of a lone wave).
__device__ inline unsigned wave_scan_add(unsigned x) {
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ inline unsigned wave_sum(unsigned x) { return (unsigned)__builtin_amdgcn_readlane((int)wave_scan_add(x), 63); }

// Maximum / sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the lanes that hold one COLUMN of a 16-wide MFMA tile),
// result in all four: two gfx950 row swaps (v_permlane32_swap / v_permlane16_swap: VALU, no LDS round trip) instead of two
// ds_bpermute — the scans' epilogues run this once per (row block, query group) with their loads waiting behind it.
// The swaps are issued as inline asm: through the builtins hipcc folded max(result[0], result[1]) into result[0] (ROCm
// 7.2: it does not model that the swap makes the two results differ when both operands hold the same value) — the column
// maxima were then maxima over one lane.  s_nop on both sides: the hazard recogniser does not see into the asm
// (VALU write -> permlane swap read, permlane swap write -> VALU read).
__device__ inline void permlane32_swap(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ inline void permlane16_swap(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ inline float col4_max(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    permlane32_swap(a, b);   // a = [lo, lo], b = [hi, hi]
    v = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
    a = b = __builtin_bit_cast(unsigned, v);
    permlane16_swap(a, b);   // a = [r0, r0, r2, r2], b = [r1, r1, r3, r3]
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ inline float col4_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    permlane32_swap(a, b);
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    a = b = __builtin_bit_cast(unsigned, v);
    permlane16_swap(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

// Maximum over the 64 lanes with DPP row shifts / broadcasts (VALU only: each __shfl_xor step of wave_max below is a
// ds_bpermute, ~100 cycles of LDS latency); the result is uniform (read from lane 63).
__device__ inline float wave_max_dpp(float v) {
#define HR_DPP_MAX(ctrl, rows)                                                                                      \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),                  \
                                                                        __builtin_bit_cast(int, v), ctrl, rows, 0xf, false)))
    HR_DPP_MAX(0x111, 0xf);   // row_shr:1   (lanes without a source keep v)
    HR_DPP_MAX(0x112, 0xf);   // row_shr:2
    HR_DPP_MAX(0x114, 0xf);   // row_shr:4
    HR_DPP_MAX(0x118, 0xf);   // row_shr:8  -> lane 15 of each row holds the row's maximum
    HR_DPP_MAX(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    HR_DPP_MAX(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef HR_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ inline float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

}  // namespace hbmrag
