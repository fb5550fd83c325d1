// BM25 document payloads on the device: tokenise, hash, count, weigh — one block per document.
//
// Replaces the per-document Python of the `encode_sparse` hook on the ingest path (reference indexing.py:629-654 calls
// embedding_generator.encode_sparse(text) once per chunk, :379-404 builds the CSR) for the BM25 encoder this build
// ships (advanced_rag/bm25.py, DESIGN section 7: lower-cased `\b\w+\b` tokens, slot = crc32(token) mod sparse_dim,
// weight = tf (k1 + 1) / (tf + k1 (1 - b + b dl / avgdl)) in double, stored as float32).  Byte work, bound by the bytes of the
// text (4 MB per 1 024 documents of ~512 tokens): the point is taking ~0.1 ms of interpreter time per document off the host.
//
// ASCII only: for bytes < 0x80 Python's str.lower() and `\w` are the byte classes below; a document with a byte >= 0x80 is
// FLAGGED and left to the host (Unicode case mapping and categories are not restated here).
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kBm25Threads = 256;
constexpr int kBm25MaxDim = 65536;          // slots whose 16-bit counts fit the dynamic LDS (two per word: 128 KiB)
constexpr int kBm25MaxDocBytes = 65535;     // => at most 32 768 tokens: a 16-bit count cannot overflow

struct Bm25Args {
    const uint8_t* text;      // the documents' bytes, back to back
    const int64_t* off;       // [n_docs + 1] byte offsets
    int n_docs, sparse_dim, cap;
    double k1, b, avgdl;
    int32_t* idx;             // [n_docs][cap] out: slots in ascending order
    float* val;               // [n_docs][cap] out: weights
    int32_t* nnz;             // [n_docs] out
    int32_t* flags;           // [n_docs] out: 0 = done, 1 = not ASCII, 2 = too long / more than cap slots: the host encodes it
};

__device__ inline bool bm25_is_word(unsigned c) {   // `\w` for ASCII, AFTER lower-casing
    return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_';
}
__device__ inline unsigned bm25_lower(unsigned c) { return (c >= 'A' && c <= 'Z') ? c + 32u : c; }

__global__ __launch_bounds__(kBm25Threads) void bm25_encode_kernel(Bm25Args a) {
    extern __shared__ unsigned bm25_hist[];            // sparse_dim 16-bit counts, two per word
    __shared__ unsigned crc_table[256];
    __shared__ unsigned s_tokens, s_bad;
    __shared__ unsigned wsum[kBm25Threads / 64];
    const int d = blockIdx.x, tid = threadIdx.x;
    const uint8_t* p = a.text + a.off[d];
    const int64_t len64 = a.off[d + 1] - a.off[d];
    const int n_words = (a.sparse_dim + 1) / 2;
    for (int i = tid; i < n_words; i += kBm25Threads) bm25_hist[i] = 0u;
    {   // the crc32 table of zlib (reflected polynomial 0xEDB88320)
        unsigned c = (unsigned)tid;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[tid] = c;
    }
    if (tid == 0) { s_tokens = 0; s_bad = 0; }
    __syncthreads();
    if (len64 > kBm25MaxDocBytes) {
        if (tid == 0) { a.nnz[d] = 0; a.flags[d] = 2; }
        return;
    }
    const int len = (int)len64;
    // ---- tokens: the thread that sees a token's first byte hashes the whole token
    unsigned mine = 0, bad = 0;
    for (int i = tid; i < len; i += kBm25Threads) {
        const unsigned raw = p[i];
        bad |= raw >> 7;
        const unsigned c = bm25_lower(raw);
        if (!bm25_is_word(c) || (i > 0 && bm25_is_word(bm25_lower(p[i - 1])))) continue;
        unsigned crc = 0xFFFFFFFFu;
        int j = i;
        unsigned cj = c;
        do {
            crc = crc_table[(crc ^ cj) & 255u] ^ (crc >> 8);
            ++j;
            cj = j < len ? bm25_lower(p[j]) : 0u;
        } while (j < len && bm25_is_word(cj));
        const unsigned slot = (crc ^ 0xFFFFFFFFu) % (unsigned)a.sparse_dim;
        atomicAdd(&bm25_hist[slot >> 1], (slot & 1u) ? 0x10000u : 1u);
        ++mine;
    }
    if (bad) atomicOr(&s_bad, 1u);
    mine = wave_sum(mine);
    if ((tid & 63) == 0 && mine) atomicAdd(&s_tokens, mine);
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { a.nnz[d] = 0; a.flags[d] = 1; }
        return;
    }
    // ---- weights, slots in ascending order: thread t owns a contiguous range of slots
    const double dl = (double)s_tokens;
    const double norm = a.k1 * (1.0 - a.b + a.b * dl / a.avgdl);
    const int per = (a.sparse_dim + kBm25Threads - 1) / kBm25Threads;
    const int s0 = min(tid * per, a.sparse_dim), s1 = min(s0 + per, a.sparse_dim);
    auto count_of = [&](int s) -> unsigned { return (bm25_hist[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu; };
    unsigned n_mine = 0;
    for (int s = s0; s < s1; ++s) n_mine += count_of(s) != 0u;
    const unsigned incl = wave_scan_add(n_mine);
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned before = incl - n_mine, total = 0;
    for (int w = 0; w < kBm25Threads / 64; ++w) {
        if (w < (tid >> 6)) before += wsum[w];
        total += wsum[w];
    }
    if (total > (unsigned)a.cap) {
        if (tid == 0) { a.nnz[d] = 0; a.flags[d] = 2; }
        return;
    }
    int32_t* oi = a.idx + (int64_t)d * a.cap;
    float* ov = a.val + (int64_t)d * a.cap;
    unsigned pos = before;
    for (int s = s0; s < s1; ++s) {
        const unsigned tf = count_of(s);
        if (!tf) continue;
        const double x = (double)tf;
        oi[pos] = s;
        ov[pos] = (float)(x * (a.k1 + 1.0) / (x + norm));   // > 0 for tf >= 1 (bm25.py keeps weights > 0)
        ++pos;
    }
    if (tid == 0) { a.nnz[d] = (int32_t)total; a.flags[d] = 0; }
}

// ---- the hash tokenizer of the encoder hooks on the device ------------------------------------------------------
// advanced_rag/encoders.py::HashTokenizer (the offline stand-in for WordPiece in front of the sentence encoder: reference
// hook embedding_generator.encode_semantic, indexing.py:610-620, batch form :580-587): tokens of the lower-cased text =
// `\w+|[^\w\s]` (a word run, or ONE character that is neither word nor space), id = 1000 + crc32(token) mod (vocab - 1000),
// row = [CLS] ids[: max_len - 2] [SEP], zero padded.  One block per text; a thread owns a contiguous slice of the bytes,
// counts the tokens that START in it, and after a block-wide prefix sum writes their ids at their ordinals.  ASCII only
// (a text with a byte >= 0x80 is flagged and the caller tokenises the batch on the host).
struct TokArgs {
    const uint8_t* text;
    const int64_t* off;       // [n + 1]
    int n, max_len, vocab;    // row width = max_len (CLS / SEP included)
    int64_t* ids;             // [n][max_len] out, 0 = padding
    int32_t* lens;            // [n] out: tokens in the row, CLS and SEP included
    int32_t* flags;           // [n] out: 0 = done, 1 = not ASCII
};
constexpr int kTokThreads = 256;
__device__ inline bool tok_is_space(unsigned c) { return c == 0x20u || (c >= 0x09u && c <= 0x0du) || (c >= 0x1cu && c <= 0x1fu); }

__global__ __launch_bounds__(kTokThreads) void hash_tokenize_kernel(TokArgs a) {
    __shared__ unsigned crc_table[256];
    __shared__ unsigned wsum[kTokThreads / 64];
    __shared__ unsigned s_bad;
    const int d = blockIdx.x, tid = threadIdx.x;
    const uint8_t* p = a.text + a.off[d];
    const int64_t len = a.off[d + 1] - a.off[d];
    {
        unsigned c = (unsigned)tid;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[tid] = c;
    }
    if (tid == 0) s_bad = 0;
    int64_t* row = a.ids + (int64_t)d * a.max_len;
    for (int i = tid; i < a.max_len; i += kTokThreads) row[i] = 0;
    __syncthreads();
    const int64_t per = (len + kTokThreads - 1) / kTokThreads;
    const int64_t b0 = min((int64_t)tid * per, len), b1 = min(b0 + per, len);
    auto starts_token = [&](int64_t i, unsigned c) {   // c = lower-cased byte i
        if (bm25_is_word(c)) return i == 0 || !bm25_is_word(bm25_lower(p[i - 1]));
        return !tok_is_space(c);
    };
    unsigned mine = 0, bad = 0;
    for (int64_t i = b0; i < b1; ++i) {
        const unsigned raw = p[i];
        bad |= raw >> 7;
        mine += starts_token(i, bm25_lower(raw)) ? 1u : 0u;
    }
    if (bad) atomicOr(&s_bad, 1u);
    const unsigned incl = wave_scan_add(mine);
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { a.lens[d] = 0; a.flags[d] = 1; }
        return;
    }
    unsigned ord = incl - mine, total = 0;
    for (int w = 0; w < kTokThreads / 64; ++w) {
        if (w < (tid >> 6)) ord += wsum[w];
        total += wsum[w];
    }
    const unsigned room = (unsigned)(a.max_len - 2);
    const unsigned span = (unsigned)(a.vocab - 1000);
    for (int64_t i = b0; i < b1 && ord < room; ++i) {
        const unsigned c = bm25_lower(p[i]);
        if (!starts_token(i, c)) continue;
        unsigned crc = 0xFFFFFFFFu;
        if (bm25_is_word(c)) {
            int64_t j = i;
            unsigned cj = c;
            do {
                crc = crc_table[(crc ^ cj) & 255u] ^ (crc >> 8);
                ++j;
                cj = j < len ? bm25_lower(p[j]) : 0u;
            } while (j < len && bm25_is_word(cj));
        } else {
            crc = crc_table[(crc ^ c) & 255u] ^ (crc >> 8);
        }
        row[1 + ord] = 1000 + (int64_t)((crc ^ 0xFFFFFFFFu) % span);
        ++ord;
    }
    if (tid == 0) {
        const unsigned kept = total < room ? total : room;
        row[0] = 101;             // CLS
        row[1 + kept] = 102;      // SEP
        a.lens[d] = (int32_t)kept + 2;
        a.flags[d] = 0;
    }
}

}  // namespace hbmrag
