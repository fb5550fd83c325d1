// Elementwise pieces of the encoder / cross-encoder forward that PyTorch leaves unfused (the GEMMs and the attention
// stay with hipBLASLt / SDPA, as BASELINE's north_star prescribes: "MFMA only for the encoder/reranker GEMMs").
//
// The reference's CrossEncoderReranker (reference retrieval.py:651-685) names cross-encoder/ms-marco-MiniLM-L-6-v2, a
// post-LN BERT: every layer computes LayerNorm(x + sublayer(x)) twice.  In PyTorch that is an add kernel plus a
// layer-norm kernel per use — at 2560 pairs x 128 tokens x 384 dims the layer norm alone ran at ~1 TB/s (13 launches
// of 0.47 ms per forward, 24 % of the forward: profiles/r2_config4_cross_encoder_kernel_stats.csv).  Here: one pass,
// two reads and one write per element, 16 lanes per row with 16-byte loads, fp32 statistics.
#pragma once
#include "common.h"

namespace hbmrag {

// Sum over the 16 lanes of a DPP row, result in every lane of the row (four row rotates, no LDS).
__device__ inline float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));  // row_ror:1
    return v;
}

// out[r] = LayerNorm(x[r] (+ res[r])) * gamma + beta over rows of `chunks` 16-byte chunks (hidden = 8 * chunks halves).
// The sum x + res is rounded to fp16 before the statistics, as the unfused fp16 module does.  out may alias x or res.
// Block = 256 threads = 16 rows; lane j of a row's 16 lanes owns chunks j, j + 16, ... (NC of them at most).
template <int NC>
__global__ __launch_bounds__(256) void add_layernorm_f16_kernel(const half8_t* __restrict__ x, const half8_t* res,
                                                                const half8_t* __restrict__ gamma,
                                                                const half8_t* __restrict__ beta, half8_t* out,
                                                                int64_t rows, int chunks, float eps) {
    const int j = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;  // a whole DPP row leaves together
    const half8_t* xr = x + row * chunks;
    const half8_t* rr = res ? res + row * chunks : nullptr;
    half8_t h[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = j + 16 * i;
        if (c < chunks) {
            h[i] = xr[c];
            if (rr) h[i] = h[i] + rr[c];  // fp16 add, round to nearest
        } else {
            h[i] = (half8_t)(_Float16)0;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)h[i][e];
    const float inv_n = 1.0f / (float)(chunks * 8);
    const float mean = row16_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (j + 16 * i < chunks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)h[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(row16_sum(q) * inv_n + eps);
    half8_t* orow = out + row * chunks;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = j + 16 * i;
        if (c < chunks) {
            const half8_t g = gamma[c], b = beta[c];
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)h[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            orow[c] = o;
        }
    }
}

// The embedding layer of the same models in one pass: out[s, t] = LayerNorm(word[ids[s, t]] + pos[t] + seg[types[s, t]]).
// PyTorch runs two gathers, a broadcast add and the add + LayerNorm above — ~1.8 GB of traffic at 2560 x 128 tokens x 384
// dims for 0.25 GB of output; here the three table rows come from L2 and only the output goes to HBM.  The two adds are
// rounded to fp16 one after the other, as the unfused fp16 module rounds them.  Ids / types outside the tables are clamped
// to the tables' first / last row (no read out of bounds, no host-side check that would synchronise the stream).
template <int NC>
__global__ __launch_bounds__(256) void embed_layernorm_f16_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ types,
                                                                  const half8_t* __restrict__ word, const half8_t* __restrict__ pos,
                                                                  const half8_t* __restrict__ seg, const half8_t* __restrict__ gamma,
                                                                  const half8_t* __restrict__ beta, half8_t* __restrict__ out,
                                                                  int64_t rows, int T, int chunks, float eps, int64_t n_word, int64_t n_seg) {
    const int j = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;  // a whole DPP row leaves together
    const half8_t* wr = word + min(max(ids[row], (int64_t)0), n_word - 1) * chunks;
    const half8_t* pr = pos + (row % T) * chunks;
    const half8_t* sr = seg + min(max(types[row], (int64_t)0), n_seg - 1) * chunks;
    half8_t h[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = j + 16 * i;
        if (c < chunks) {
            h[i] = wr[c] + pr[c];     // fp16 adds, each rounded to nearest
            h[i] = h[i] + sr[c];
        } else {
            h[i] = (half8_t)(_Float16)0;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)h[i][e];
    const float inv_n = 1.0f / (float)(chunks * 8);
    const float mean = row16_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        if (j + 16 * i < chunks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)h[i][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(row16_sum(q) * inv_n + eps);
    half8_t* orow = out + row * chunks;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = j + 16 * i;
        if (c < chunks) {
            const half8_t g = gamma[c], b = beta[c];
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)h[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            orow[c] = o;
        }
    }
}

}  // namespace hbmrag
