// Self-attention of the encoder / cross-encoder forward for head dimension 32 (MiniLM-L6-H384: 12 heads x 32 —
// the model class the reference's CrossEncoderReranker names, retrieval.py:651-662), straight from the fused QKV
// projection's output to the [tokens, hidden] layout the output projection reads.  PyTorch's SDPA spends 0.53 ms per
// layer on 2560 sequences x 128 tokens (15x its memory bound: a 32-wide head fills a quarter of a generic flash
// tile), plus the permute / transpose copies around it.
//
// One block (4 waves) per (sequence, head, 128 queries); a wave owns 32 queries (two 16-wide tiles) and walks the keys
// in chunks of 32 with an online softmax.  Everything a query needs stays in ITS lane column:
//   S^T[keys x queries] = K . Q^T     A = K rows (16 keys x 32 dims: ONE v_mfma_f32_16x16x32_f16 k-step),
//                                     B = Q rows  -> lane (query = l & 15, g = l >> 4) holds keys 4g .. 4g+3 of a tile
//   O^T[dims x queries] += V^T . P^T  A = V^T (16 dims x 32 keys), B = P^T: the lane's own eight probabilities of the
//                                     chunk ARE its B fragment (k index j <-> key 4g+j of tile 0, 4g+j-4 of tile 1),
//                                     so P never moves between lanes; V is staged once per block in LDS, transposed
//                                     and in that same key order, so an A fragment is one ds_read_b128
// and the running maximum / sum / output of a query are rescaled by a per-lane scalar.
// K and Q fragments are 16-byte global loads (a key's / query's 32 dims are 64 contiguous bytes of the QKV row).
// Keys at or beyond the sequence's length (padding at the tail, as HashTokenizer.batch pads) get probability 0.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kAttnHeadDim = 32;
constexpr int kAttnQueriesPerBlock = 128;

__device__ inline float wave_col_max(float v) {   // over the four lane groups that share a query column
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

// qkv: [n_seq][T][3][heads][32] halves; lengths: [n_seq] valid tokens (null = T); out: [n_seq][T][heads * 32] halves.
__global__ __launch_bounds__(256) void attention_hd32_kernel(const _Float16* __restrict__ qkv, const int32_t* __restrict__ lengths,
                                                             _Float16* __restrict__ out, int T, int heads, float scale_log2e) {
    extern __shared__ half8_t attn_vt[];   // [chunks][2 dim tiles][64 lanes] fragments of V^T
    const int seq = blockIdx.x / heads, head = blockIdx.x % heads;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int H = heads * kAttnHeadDim;
    const int len = lengths ? min(lengths[seq], T) : T;
    const int n_chunks = (T + 31) / 32;
    const _Float16* base = qkv + (int64_t)seq * T * 3 * H + head * kAttnHeadDim;   // + t * 3H (+ H for K, + 2H for V)

    // Q and the first K fragments are requested before V is staged: three dependent round trips become two
    const int q0 = blockIdx.y * kAttnQueriesPerBlock + wid * 32;
    half8_t qf[2], kf[2], kn[2];
    auto load_k = [&](int c, half8_t (&kfr)[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = min(32 * c + 16 * kt + col, T - 1);
            kfr[kt] = *reinterpret_cast<const half8_t*>(base + (int64_t)key * 3 * H + H + 8 * g);
        }
    };
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = min(q0 + 16 * qt + col, T - 1);
        qf[qt] = *reinterpret_cast<const half8_t*>(base + (int64_t)q * 3 * H + 8 * g);
    }
    load_k(0, kf);
    // the softmax scale (x log2 e) goes into Q once — 8 multiplies per query tile instead of one per score; the
    // product is rounded to fp16 like Q itself (|scale| < 1: no overflow), well inside the kernel's fp16 tolerance
    const _Float16 qs = (_Float16)scale_log2e;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[qt][i] *= qs;

    // ---- stage V^T: thread takes (key, 8 dims) pieces; dim d of key k goes to fragment (chunk, d >> 4), lane (d & 15) + 16 g', slot j
    _Float16* vt = reinterpret_cast<_Float16*>(attn_vt);
    for (int piece = threadIdx.x; piece < n_chunks * 32 * 4; piece += 256) {
        const int key = piece >> 2, e = piece & 3;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) v = *reinterpret_cast<const half8_t*>(base + (int64_t)key * 3 * H + 2 * H + 8 * e);
        const int c = key >> 5, kk = key & 31;
        const int gg = (kk & 15) >> 2, j = (kk & 3) + (kk >= 16 ? 4 : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = 8 * e + i;
            vt[(((c * 2 + (d >> 4)) * 64) + (d & 15) + 16 * gg) * 8 + j] = v[i];
        }
    }
    __syncthreads();
    if (q0 >= T) return;
    f32x4_t o[2][2];   // [dim tile][query tile]
    float m[2], l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        m[qt] = -__builtin_inff();
        l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int live_chunks = (len + 31) / 32;   // chunks beyond the sequence's length hold nothing but masked keys
    for (int c = 0; c < live_chunks; ++c) {
        load_k(c + 1 < live_chunks ? c + 1 : c, kn);
        f32x4_t s[2][2];   // [key tile][query tile]
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
                s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf[qt], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // keys at or beyond the length exist only in the sequence's last live chunk: the mask costs nothing elsewhere
        if (32 * c + 32 > len) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (32 * c + 16 * kt + 4 * g + r >= len) s[kt][qt][r] = -__builtin_inff();
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                             fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
            // key 0 is always valid (length >= 1), so the running maximum is finite from the first chunk on
            const float m_new = fmaxf(m[qt], wave_col_max(mx));
            const float alpha = exp2f(m[qt] - m_new);   // first chunk: exp2(-inf) = 0 times a zero accumulator
            m[qt] = m_new;
            float e[8];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) e[4 * kt + r] = exp2f(s[kt][qt][r] - m_new);
            const float sum = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
            union { half8_t v; fp16x2_t h2[4]; } p;
#pragma unroll
            for (int i = 0; i < 4; ++i) p.h2[i] = __builtin_amdgcn_cvt_pkrtz(e[2 * i], e[2 * i + 1]);
            l[qt] = l[qt] * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][qt][r] *= alpha;
                o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_vt[(c * 2 + dt) * 64 + lane], p.v, o[dt][qt], 0, 0, 0);
            }
        }
        kf[0] = kn[0];
        kf[1] = kn[1];
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + col;
        float lt = l[qt];
        lt += __shfl_xor(lt, 16);
        lt += __shfl_xor(lt, 32);
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        if (q < T) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                half4_t w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (_Float16)(o[dt][qt][r] * inv);
                *reinterpret_cast<half4_t*>(out + ((int64_t)seq * T + q) * H + head * kAttnHeadDim + 16 * dt + 4 * g) = w;
            }
        }
    }
}

}  // namespace hbmrag
