// Self-attention of the encoder / cross-encoder forward for head dimensions 32 (MiniLM-L6-H384: 12 heads x 32 —
// the model class the reference's CrossEncoderReranker names, retrieval.py:651-662) and 64 (bge-base 12 x 64, bge-large
// 16 x 64: the sentence encoders BASELINE configs 3-5 name; round 4: template parameter HD, two k-steps per score tile and
// four output dim tiles), straight from the fused QKV projection's output to the [tokens, hidden] layout the output
// projection reads.  The text below describes HD = 32; HD = 64 differs only in those counts.  PyTorch's SDPA spends 0.53 ms per
// layer on 2560 sequences x 128 tokens (15x its memory bound: a 32-wide head fills a quarter of a generic flash
// tile), plus the permute / transpose copies around it.
//
// One block of NW waves per (sequence, head, 32 NW queries) — NW = 4 up to 128 tokens, 8 beyond; a wave owns 32 queries
// (two 16-wide tiles) and walks the keys in chunks of 32 with an online softmax.  K and V of the (sequence, head) are
// staged ONCE per block in LDS (K as it stands, V transposed into fragment order): with K fetched per wave from global
// memory (the first form) every wave re-read all keys as 64-byte pieces 2 304 bytes apart and the kernel waited on those
// loads (T = 512: 2.7 ms per layer against 0.4 ms of MFMA time).  Everything a query needs stays in ITS lane column:
//   S^T[keys x queries] = K . Q^T     A = K rows (16 keys x 32 dims: ONE v_mfma_f32_16x16x32_f16 k-step),
//                                     B = Q rows  -> lane (query = l & 15, g = l >> 4) holds keys 4g .. 4g+3 of a tile
//   O^T[dims x queries] += V^T . P^T  A = V^T (16 dims x 32 keys), B = P^T: the lane's own eight probabilities of the
//                                     chunk ARE its B fragment (k index j <-> key 4g+j of tile 0, 4g+j-4 of tile 1),
//                                     so P never moves between lanes; V is staged once per block in LDS, transposed
//                                     and in that same key order, so an A fragment is one ds_read_b128
// and the reference maximum / sum / output of a query are rescaled by a per-lane scalar — rarely (see the loop).
// Q fragments are 16-byte global loads (a query's 32 dims are 64 contiguous bytes of the QKV row); a K fragment is one
// conflict-free ds_read_b128 (a wave reads 16 keys x 64 bytes = 1 KiB contiguous).
// Keys at or beyond the sequence's length (padding at the tail, as HashTokenizer.batch pads) get probability 0.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kAttnHeadDim = 32;
constexpr int kAttnMaxT = 1024;   // K + V of a (sequence, head) in LDS: 4 * HD bytes per token (HD = 64: 512 tokens)
constexpr int kAttnLdsBytes = kAttnMaxT * 128;

__device__ inline float wave_col_max(float v) { return col4_max(v); }   // over the four lane groups that share a query column

// Operands by pointer and stride (AttnArgs): the fused QKV buffer [n_seq][T][3][heads][32] of an encoder layer, or separate
// query rows and a KV buffer (the cross-encoder's last layer: ONE query row per sequence against [n_seq][T][2][heads][32]).
// lengths: [n_seq] valid keys (null = T); the first n_queries tokens of a sequence are its queries;
// out: [n_seq][n_queries][heads * 32] halves.
//
// Grid: 1-D, XCD-aware.  A head's Q / K / V piece of a token is 64 bytes, so a 128-byte line of the QKV buffer belongs to
// TWO heads, and at T > 128 the blocks of one (sequence, head) all read the same K and V.  Workgroup L runs on XCD L % 8
// (strict round-robin dispatch: tests/probes/xcd_dispatch_census.hip), each XCD with its own L2: the G = 2 x q-blocks
// blocks of a head PAIR are given linear ids (P / 8) * 8 G + i * 8 + P % 8, i < G — consecutive launches on ONE XCD — so a
// line is fetched from HBM once, not once per block that needs it.  (With (sequence x head, q-block) as a 2-D grid the two
// heads of a line sat on different XCDs and the q-blocks of a head 30 720 workgroups apart: 2x the QKV bytes at T = 128,
// up to 5x at T = 512.)
struct AttnArgs {
    const _Float16* q;   // query rows: q + seq * q_seq + token * q_tok + head * 32
    const _Float16* k;   // key rows:   k + seq * kv_seq + token * kv_tok + head * 32 (v likewise)
    const _Float16* v;
    _Float16* out;       // out + (seq * n_queries + token) * heads * 32 + head * 32
    const int32_t* lengths;
    int64_t q_seq, q_tok, kv_seq, kv_tok;   // strides in halves
    int T, n_queries, heads, n_qblocks;     // keys per sequence; the first n_queries tokens are the queries
    int64_t n_pairs;                        // HD = 32: head pairs (two heads share a 128-byte line); HD = 64: heads
    float scale_log2e;
    int out_fr;                             // out in fragment order (encoder_layer.h): what encoder_tail_kernel reads
};

template <int NW, int HD>
__global__ __launch_bounds__(64 * NW, (NW == 4 && HD == 32) ? 2 : 1) void attention_kernel(AttnArgs a) {
    constexpr int KS = HD / 32;   // k-steps of a score tile
    constexpr int DT = HD / 16;   // output dim tiles
    constexpr int PC = HD / 8;    // 16-byte pieces of a key row
    constexpr int HG = HD == 32 ? 2 : 1;   // heads whose blocks are kept on one XCD (they share 128-byte lines)
    const int32_t* __restrict__ lengths = a.lengths;
    _Float16* __restrict__ out = a.out;
    const int T = a.T, heads = a.heads, n_qblocks = a.n_qblocks;
    const int64_t n_pairs = a.n_pairs;
    const float scale_log2e = a.scale_log2e;
    extern __shared__ half8_t attn_vt[];   // [chunks][DT dim tiles][64 lanes] fragments of V^T, then K: [chunks * 32 keys][PC] pieces
    const int G = HG * n_qblocks;
    const int64_t L = blockIdx.x;
    const int64_t P = (L / (8 * G)) * 8 + (L & 7);        // head group: sequence * ceil(heads / HG) + group of the sequence
    const int i_blk = (int)((L / 8) % G);
    if (P >= n_pairs) return;
    const int pairs_per_seq = (heads + HG - 1) / HG;
    const int seq = (int)(P / pairs_per_seq), head = HG * (int)(P % pairs_per_seq) + i_blk / n_qblocks;
    const int q_block = i_blk % n_qblocks;
    if (head >= heads) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int H = heads * HD;
    const int len = lengths ? min(lengths[seq], T) : T;
    const int n_chunks = (T + 31) / 32;
    const _Float16* qb = a.q + seq * a.q_seq + head * HD;
    const _Float16* kb = a.k + seq * a.kv_seq + head * HD;
    const _Float16* vb = a.v + seq * a.kv_seq + head * HD;
    const int NQ = a.n_queries;
    half8_t* ks = attn_vt + n_chunks * DT * 64;
    // K rows in LDS: the PC pieces of a key are XOR-rotated so that the 16 lanes of one ds_read_b128 group (16 different
    // keys, one piece each) hit 16 different 16-byte bank groups: rows of 64 bytes rotate by key / 4 over 4 slots, rows of
    // 128 bytes by key / 2 over 8
    auto rot = [](int key) { return HD == 32 ? ((key >> 2) & 3) : ((key >> 1) & 7); };

    // Q is requested before K and V are staged
    const int q0 = q_block * 32 * NW + wid * 32;
    half8_t qf[2][KS], kf[2][KS];
    auto load_k = [&](int c, half8_t (&kfr)[2][KS]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < KS; ++s) kfr[kt][s] = ks[(32 * c + 16 * kt + col) * PC + ((4 * s + g) ^ rot(col))];
    };
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = min(q0 + 16 * qt + col, NQ - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[qt][s] = *reinterpret_cast<const half8_t*>(qb + (int64_t)q * a.q_tok + 32 * s + 8 * g);
    }
    // the softmax scale (x log2 e) goes into Q once — 8 multiplies per query tile instead of one per score; the
    // product is rounded to fp16 like Q itself (|scale| < 1: no overflow), well inside the kernel's fp16 tolerance
    const _Float16 qs = (_Float16)scale_log2e;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[qt][s][i] *= qs;

    // ---- stage K (as it stands) and V^T: thread takes (key, 8 dims) pieces; dim d of V's key k goes to fragment
    // (chunk, d >> 4), lane (d & 15) + 16 g', slot j
    _Float16* vt = reinterpret_cast<_Float16*>(attn_vt);
    for (int piece = threadIdx.x; piece < n_chunks * 32 * PC; piece += 64 * NW) {
        const int key = piece / PC, e = piece % PC;
        half8_t v = {0, 0, 0, 0, 0, 0, 0, 0}, kk8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) {
            kk8 = *reinterpret_cast<const half8_t*>(kb + (int64_t)key * a.kv_tok + 8 * e);
            v = *reinterpret_cast<const half8_t*>(vb + (int64_t)key * a.kv_tok + 8 * e);
        }
        ks[key * PC + (e ^ rot(key))] = kk8;
        const int c = key >> 5, kk = key & 31;
        const int gg = (kk & 15) >> 2, j = (kk & 3) + (kk >= 16 ? 4 : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = 8 * e + i;
            vt[(((c * DT + (d >> 4)) * 64) + (d & 15) + 16 * gg) * 8 + j] = v[i];
        }
    }
    __syncthreads();
    if (q0 >= NQ) return;
    f32x4_t o[DT][2];   // [dim tile][query tile]
    float m[2], l[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        m[qt] = -__builtin_inff();
        l[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][qt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int live_chunks = (len + 31) / 32;   // chunks beyond the sequence's length hold nothing but masked keys
    // The softmax is what this kernel spends its time on (8 MFMAs per chunk against the vector work of 16 scores per lane),
    // so the common path is kept to: scores straight out of the MFMA already minus the reference maximum (it is the
    // accumulator's initial value), 16 raw v_exp_f32, 8 pack conversions, the partial sums.  The reference maximum m[qt] of
    // a query is NOT the running maximum: it is only moved (and the sum and the output rescaled — a branch the whole wave
    // takes) when some score exceeds it by about 2^8, so every probability stays below 2^8 — well inside fp16 — and the
    // sums are the same sums up to rounding.  The first chunk always takes the branch (m = true maximum of the chunk);
    // maxima are formed only there.
    for (int c = 0; c < live_chunks; ++c) {
        load_k(c, kf);
        f32x4_t s[2][2];   // [key tile][query tile]: score - m[qt]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float neg = c == 0 ? 0.f : -m[qt];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt][qt] = (f32x4_t){neg, neg, neg, neg};
#pragma unroll
                for (int ss = 0; ss < KS; ++ss)
                    s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][ss], qf[qt][ss], s[kt][qt], 0, 0, 0);
            }
        }
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
        float e[2][8], sum8[2];
        auto exps = [&]() {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[qt][4 * kt + r] = __builtin_amdgcn_exp2f(s[kt][qt][r]);
                sum8[qt] = ((e[qt][0] + e[qt][1]) + (e[qt][2] + e[qt][3])) + ((e[qt][4] + e[qt][5]) + (e[qt][6] + e[qt][7]));
            }
        };
        exps();
        // a score more than the gap above its reference maximum shows in the lane's partial sum (every term is >= 0): no
        // maxima are formed on the common path.  (A sum can pass 2^gap without such a score — eight scores just below the
        // gap: the branch is then taken needlessly, which is harmless.)
        if (c == 0 || __any(fmaxf(sum8[0], sum8[1]) > 256.f)) {
            // move the reference maxima to the true maxima seen so far (key 0 is always valid: finite from chunk 0 on)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float mx = fmaxf(fmaxf(fmaxf(s[0][qt][0], s[0][qt][1]), fmaxf(s[0][qt][2], s[0][qt][3])),
                                       fmaxf(fmaxf(s[1][qt][0], s[1][qt][1]), fmaxf(s[1][qt][2], s[1][qt][3])));
                const float up = fmaxf(wave_col_max(mx), c == 0 ? -__builtin_inff() : 0.f);   // relative to m[qt] (chunk 0: absolute)
                const float alpha = c == 0 ? 0.f : __builtin_amdgcn_exp2f(-up);
                m[qt] = c == 0 ? up : m[qt] + up;
                l[qt] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][qt][r] -= up;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][qt][r] *= alpha;
            }
            exps();
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
            union { half8_t v; fp16x2_t h2[4]; } p;
#pragma unroll
            for (int i = 0; i < 4; ++i) p.h2[i] = __builtin_amdgcn_cvt_pkrtz(e[qt][2 * i], e[qt][2 * i + 1]);
            l[qt] += sum8[qt];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(attn_vt[(c * DT + dt) * 64 + lane], p.v, o[dt][qt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 16 * qt + col;
        float lt = l[qt];
        lt = col4_sum(lt);
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        if (q < NQ) {
            if (a.out_fr) {
                // the lane's values ARE a B fragment of the output projection (accumulator k order): k-step s = head (HD = 32) or
                // 2 head + dt / 2 (HD = 64), element j = 4 (dt & 1) + r; row n = seq * NQ + q sits in tile n / 16 at column n % 16
                const int64_t n = (int64_t)seq * NQ + q;
                half8_t* dst = reinterpret_cast<half8_t*>(out) + ((n >> 4) * (H / 32) + head * KS) * 64 + 16 * g + (int)(n & 15);
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    half8_t w;
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = (_Float16)(o[2 * s2 + (j >> 2)][qt][j & 3] * inv);
                    dst[s2 * 64] = w;
                }
            } else {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                    half4_t w;
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = (_Float16)(o[dt][qt][r] * inv);
                    *reinterpret_cast<half4_t*>(out + ((int64_t)seq * NQ + q) * H + head * HD + 16 * dt + 4 * g) = w;
                }
            }
        }
    }
}

}  // namespace hbmrag
