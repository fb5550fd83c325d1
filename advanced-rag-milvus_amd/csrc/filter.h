// Filter expressions on the device: `field OP value and field OP value ...` (the expressions
// HybridRetriever._build_filter_expression emits, reference src/advanced_rag/retrieval.py:565-632, over the scalar
// fields of the collection schema, indexing.py:191-225) evaluated over columns that live in HBM, straight into the
// packed row mask the scans and refines test (bit r%8 of byte r/8) — Milvus evaluates them server-side; nothing
// row-sized crosses PCIe here.
//
// Columns: int64 (chunk_index, token_count), float32 (entropy, redundancy, domain_density), and for string fields
// (doc_id, chunk_id, timestamp) an ORDER-PRESERVING 16-byte prefix key per row (two big-endian uint64 words of the
// zero-padded UTF-8 bytes; advanced_rag/columns.py).  A string term is decided by the key wherever the first 16 bytes
// of row and value differ; rows that tie are reported in a second bit mask ("undecided") and resolved by the host on
// the full strings — a handful of rows, if any.  Comparison rules are numpy's for the same column and literal types
// (advanced_rag/filters.py is the restatement the tests compare with): int64 column vs int -> int64; int64 column vs
// float -> float64; float32 column vs number -> float32.
#pragma once
#include "common.h"

namespace hbmrag {

constexpr int kFilterMaxTerms = 16;

struct FilterArgs {
    hr_filter_term t[kFilterMaxTerms];
    int n_terms;
    int64_t n_rows;
    const uint8_t* deleted;      // tombstones, 1 bit per row (1 = deleted), or null
    unsigned long long* mask;    // out: ceil(n_rows / 64) words
    unsigned long long* undecided;
    int32_t* counts;             // [0] += rows kept, [1] += rows undecided
};

template <typename T>
__device__ inline bool filter_cmp(T a, T b, int op) {
    switch (op) {
        case HR_OP_EQ: return a == b;
        case HR_OP_NE: return a != b;
        case HR_OP_LT: return a < b;
        case HR_OP_LE: return a <= b;
        case HR_OP_GT: return a > b;
        default: return a >= b;
    }
}

// One wave per 64 consecutive rows per trip: lane = row, so every column read is one coalesced wave load and the
// 64 verdicts leave as one 8-byte store (ballot).
__global__ __launch_bounds__(256) void filter_eval_kernel(FilterArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int64_t n_words = (a.n_rows + 63) / 64;
    int kept = 0, und = 0;
    for (int64_t w = wave; w < n_words; w += n_waves) {
        const int64_t row = w * 64 + lane;
        const bool in = row < a.n_rows;
        bool fail = !in, maybe = false;
        if (in && a.deleted) fail = (a.deleted[row >> 3] >> (row & 7)) & 1;
        for (int i = 0; i < a.n_terms; ++i) {
            const hr_filter_term& t = a.t[i];
            bool pass = true, tie = false;
            if (in) {
                switch (t.kind) {
                    case HR_COL_I64: pass = filter_cmp<int64_t>(((const int64_t*)t.col)[row], t.ival, t.op); break;
                    case HR_COL_I64_VS_F64: pass = filter_cmp<double>((double)((const int64_t*)t.col)[row], t.dval, t.op); break;
                    case HR_COL_F32: pass = filter_cmp<float>(((const float*)t.col)[row], t.fval, t.op); break;
                    default: {  // HR_COL_STR16
                        const unsigned long long k0 = ((const unsigned long long*)t.col)[2 * row];
                        const unsigned long long k1 = ((const unsigned long long*)t.col)[2 * row + 1];
                        if (k0 == t.key[0] && k1 == t.key[1]) {
                            tie = true;  // the first 16 bytes agree: the host compares the full strings
                        } else {
                            const bool less = k0 < t.key[0] || (k0 == t.key[0] && k1 < t.key[1]);
                            pass = t.op == HR_OP_EQ ? false : t.op == HR_OP_NE ? true
                                 : (t.op == HR_OP_LT || t.op == HR_OP_LE) ? less : !less;
                        }
                    }
                }
            }
            if (tie) maybe = true;
            else if (!pass) fail = true;
        }
        const bool keep = !fail && !maybe, undecided = !fail && maybe;
        const unsigned long long km = __ballot(keep), um = __ballot(undecided);
        if (lane == 0) {
            a.mask[w] = km;
            a.undecided[w] = um;
            kept += __popcll(km);
            und += __popcll(um);
        }
    }
    if (lane == 0 && (kept | und)) {
        if (kept) atomicAdd(&a.counts[0], kept);
        if (und) atomicAdd(&a.counts[1], und);
    }
}

}  // namespace hbmrag
