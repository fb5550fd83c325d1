// libhbmrag.so — host side of the C ABI declared in include/hbmrag.h.
// Owns the in-HBM shard store (dense tiles + sparse postings) of ONE GPU and
// launches the gfx950 kernels in dense.h / select.h / sparse.h / fuse.h.
// There is deliberately no CPU fallback in this file: without a HIP device
// hr_create fails and every caller sees the error.
#include "../../include/hbmrag.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "attention.h"
#include "encoder_layer.h"
#include "dense.h"
#include "encoder_ops.h"
#include "filter.h"
#include "finish.h"
#include "fuse.h"
#include "text.h"
#include "select.h"
#include "sparse.h"

using namespace hbmrag;

namespace {

thread_local std::string g_last_error;  // for calls without a handle

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // Capacity for `bytes`, KEEPING the first `keep` bytes (device-to-device copy on stream s into a buffer 1.5x the
    // size asked for, so that repeated appends cost amortised O(appended bytes)).
    hipError_t grow(size_t bytes, size_t keep, hipStream_t s) {
        if (bytes <= cap) return hipSuccess;
        void* np_ = nullptr;
        const size_t want = bytes + bytes / 2 + 256;
        hipError_t e = hipMalloc(&np_, want);
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(np_, p, keep, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                (void)hipFree(np_);
                return e;
            }
        }
        if (p) (void)hipFree(p);
        p = np_;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Phase { PH_PREP = 0, PH_SCAN, PH_GSEL, PH_REFINE, PH_TOPK, PH_SSCAN, PH_SGSEL, PH_SREFINE, PH_STOPK, PH_FINISH, PH_COUNT };
static_assert(PH_COUNT == HR_N_PHASES, "phase table and ABI out of step");

struct Workspace {
    std::mutex mu;                 // held while one call enqueues: calls sharing a workspace must not interleave their kernels
    int users = 0;                 // calls that hold or wait for this workspace (guarded by hr_index::pool_mu): never pruned while > 0
    hipStream_t stream = nullptr;  // own stream (host-form calls)
    hipStream_t side = nullptr;    // side stream + events of hr_search_hybrid_dev
    hipEvent_t ev_scan = nullptr, ev_side = nullptr;
    struct Workspace* sparse_ws = nullptr;  // private buffers of the sparse chain when it runs concurrently
    DevBuf qfrag, qn2, gmax, bmax, cand, acut, cscore, crow, flags, qscale, qeps, pq_n, pq_idx, pq_w;
    DevBuf d_q, d_ids, d_scores, d_mask;          // host-form staging
    DevBuf d_qptr, d_qidx, d_qval;                // sparse query staging
    DevBuf f_ids, f_out_ids, f_out_scores, f_out_meth, f_n;  // hr_fuse_rrf staging
    void release() {
        for (DevBuf* b : {&qfrag, &qn2, &gmax, &bmax, &qscale, &qeps, &pq_n, &pq_idx, &pq_w, &cand, &acut, &cscore, &crow, &flags, &d_q, &d_ids, &d_scores,
                          &d_mask, &d_qptr, &d_qidx, &d_qval, &f_ids, &f_out_ids, &f_out_scores, &f_out_meth, &f_n})
            b->release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        if (side) (void)hipStreamDestroy(side);
        side = nullptr;
        if (ev_scan) (void)hipEventDestroy(ev_scan);
        if (ev_side) (void)hipEventDestroy(ev_side);
        ev_scan = ev_side = nullptr;
        if (sparse_ws) {
            sparse_ws->release();
            delete sparse_ws;
            sparse_ws = nullptr;
        }
    }
};

struct EventSpan {
    int phase;
    hipEvent_t a, b;
};

}  // namespace

struct hr_index {
    int device = 0;
    int64_t dim = 0, sparse_dim = 0;
    int dtype = HR_F16, metric = HR_METRIC_COSINE;
    int KT = 0;  // 1 KiB tiles per row block along k
    int group_rows_override = 0;  // hr_debug_option(HR_DEBUG_GROUP_ROWS) at creation pins the candidate-group size (default: by shard size)
    int64_t row_offset = 0;

    // dense shard
    DevBuf tiles, scale, norm2, max_norm;
    int64_t cap_rows = 0, n_rows = 0, n_normed = 0;
    float max_row_norm = 0.f;
    DevBuf stage;  // ingest staging
    hipStream_t ingest_stream = nullptr;

    // sparse shard: the CSR lives on the device; the host only stages the rows appended since the last hr_finalize
    std::vector<int64_t> pend_indptr{0};  // relative to the first pending entry
    std::vector<int32_t> pend_idx;
    std::vector<float> pend_val;
    std::vector<int64_t> h_range_base{0};  // first posting of every range's block (+ total), host copy
    std::vector<unsigned> h_range_dense;   // dense runs per range (sparse.h)
    // n_sparse rows added; n_csr / nnz_csr of them in the device CSR; n_sparse_built of them in the postings
    int64_t n_sparse = 0, n_csr = 0, nnz_csr = 0, n_sparse_built = 0;
    float max_sparse_abs = 0.f;  // max |doc weight|: bounds the scan's fixed-point range
    DevBuf s_indptr, s_idx, s_val, rt_off, range_base, post;  // post: packed (fp16 weight | u16 accumulator slot)
    DevBuf idle_post;  // 64 x 4 idle postings: what scan lanes with nothing to fetch read (sparse.h)
    int64_t n_ranges = 0;
    int64_t n_dense_runs = 0;      // (term, range) runs in the dense form (sparse.h): the scan's variant is chosen by it

    bool finalized = false;
    int profiling = 0;
    int cu_count = 256;
    int scan_cus = 0;              // compute units the scans' stream may use (hr_set_scan_cus); 0 = all
    int fault_inject = 0;          // hr_debug_inject_fault: fail the next build_sparse after its CSR upload (tests)
    bool slot_prepped[HR_MAX_SLOTS] = {};  // hr_hybrid_prep_dev has prepared the slot's queries for the next scan

    mutable std::shared_mutex rw;  // searches shared, add/finalize exclusive
    std::mutex pool_mu;
    std::vector<Workspace*> free_ws;
    std::map<void*, Workspace*> stream_ws;
    Workspace* slot_ws[HR_MAX_SLOTS][2] = {};  // [slot][dense|sparse] of the two-phase forms
    std::mutex prof_mu;
    std::vector<EventSpan> spans;
    std::vector<hipEvent_t> event_pool;
    mutable std::mutex err_mu;
    mutable std::string err;
};

namespace {

int fail(const hr_index* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) {
        std::lock_guard<std::mutex> g(h->err_mu);
        h->err = buf;
    }
    g_last_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                   \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) return fail(h, HR_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define HR_TRY(expr)               \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != HR_OK) return rc__; \
    } while (0)

inline size_t elem_size(int dtype) { return dtype == HR_F16 ? 2 : 4; }
inline int elems_per_chunk(int dtype) { return dtype == HR_F16 ? 8 : 4; }
inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline size_t tile_bytes_for_rows(const hr_index* h, int64_t rows) {
    return (size_t)(rows / kRowsPerBlock) * h->KT * 1024;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

Workspace* take_ws(hr_index* h) {
    std::lock_guard<std::mutex> g(h->pool_mu);
    if (!h->free_ws.empty()) {
        Workspace* w = h->free_ws.back();
        h->free_ws.pop_back();
        return w;
    }
    Workspace* w = new (std::nothrow) Workspace();
    if (!w) return nullptr;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
        delete w;
        return nullptr;
    }
    return w;
}
void give_ws(hr_index* h, Workspace* w) {
    std::lock_guard<std::mutex> g(h->pool_mu);
    h->free_ws.push_back(w);
}
// The workspace of a caller-owned stream, returned LOCKED: `*_dev` calls that share a stream share its workspace
// (query fragments, group maxima, candidates), so one call's enqueue must not interleave with another's — their
// kernels then run in stream order.  The handle-wide pool lock is held only for the map lookup: a call that waits for a
// busy workspace (its owner may be inside hipMalloc) does not stall the other streams of the handle.  The map is
// pruned when it grows — a long-lived handle used from many short-lived streams would otherwise keep every stream's
// buffers for ever — and the pruned workspaces are freed after the pool lock is dropped (hipFree synchronises the device).
constexpr size_t kMaxStreamWorkspaces = 32;
struct StreamWs {
    hr_index* h = nullptr;
    Workspace* w = nullptr;
    std::unique_lock<std::mutex> held;
    StreamWs() = default;
    StreamWs(const StreamWs&) = delete;
    StreamWs& operator=(const StreamWs&) = delete;
    ~StreamWs() {
        if (!w) return;
        if (held.owns_lock()) held.unlock();
        std::lock_guard<std::mutex> g(h->pool_mu);
        --w->users;
    }
};
Workspace* ws_for_stream(hr_index* h, void* stream, StreamWs& out) {
    std::vector<Workspace*> pruned;
    Workspace* w = nullptr;
    {
        std::lock_guard<std::mutex> g(h->pool_mu);
        auto it = h->stream_ws.find(stream);
        if (it != h->stream_ws.end()) {
            w = it->second;
        } else {
            if (h->stream_ws.size() >= kMaxStreamWorkspaces) {
                for (auto jt = h->stream_ws.begin(); jt != h->stream_ws.end();) {
                    if (jt->second->users == 0) {  // nobody holds or waits for it, and nobody can get it while pool_mu is held
                        pruned.push_back(jt->second);
                        jt = h->stream_ws.erase(jt);
                    } else {
                        ++jt;
                    }
                }
            }
            w = new (std::nothrow) Workspace();
            if (w) h->stream_ws[stream] = w;
        }
        if (w) ++w->users;
    }
    for (Workspace* old : pruned) {
        old->release();  // hipFree waits for the work that still uses the buffers
        delete old;
    }
    if (!w) return nullptr;
    out.h = h;
    out.w = w;
    out.held = std::unique_lock<std::mutex>(w->mu);
    return w;
}

// ---- profiling spans -------------------------------------------------------
struct Span {
    hr_index* h;
    hipStream_t s;
    int phase;
    hipEvent_t a = nullptr, b = nullptr;
    Span(hr_index* h_, hipStream_t s_, int phase_) : h(h_), s(s_), phase(phase_) {
        const bool on = h->profiling >= 2 || (h->profiling == 1 && (phase == PH_SCAN || phase == PH_SSCAN));
        if (!on) return;
        std::lock_guard<std::mutex> g(h->prof_mu);
        auto get = [&]() -> hipEvent_t {
            hipEvent_t e = nullptr;
            if (!h->event_pool.empty()) {
                e = h->event_pool.back();
                h->event_pool.pop_back();
            } else if (hipEventCreate(&e) != hipSuccess) {
                e = nullptr;
            }
            return e;
        };
        a = get();
        b = get();
        if (a && b) (void)hipEventRecord(a, s);
    }
    ~Span() {
        if (!a || !b) return;
        (void)hipEventRecord(b, s);
        std::lock_guard<std::mutex> g(h->prof_mu);
        h->spans.push_back({phase, a, b});
    }
};

// Compute units the scans may occupy: their persistent grids are sized to it (hr_set_scan_cus for a masked stream).
inline int scan_cus(const hr_index* h) { return h->scan_cus > 0 ? std::min(h->scan_cus, h->cu_count) : h->cu_count; }

// Rows (docs) per candidate group.  Small shards (a rank of a multi-GPU corpus) use 16-row
// groups: 4x less refine traffic per query, and the 4x larger table of group maxima is
// still small.  Big shards use 64-row groups, where selecting among 4x more maxima would
// cost more than the refine saves (measured at 10M x 768: 3.88 ms/step vs 4.08).
int group_rows_for(const hr_index* h, int64_t n) {
    if (h->group_rows_override) return h->group_rows_override;
    return n > 3000000 ? 64 : 16;
}

// ---- dense launch helpers ----------------------------------------------------
template <typename STORE, int G, int NRB>
hipError_t launch_scan(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                       int nq, int64_t n_groups) {
    constexpr int RS = 2, PF = 4;
    auto kern = dense_scan_kernel<STORE, G, RS, PF, NRB>;
    const size_t lds = (size_t)G * h->KT * 1024;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // One 1024-thread block per CU when the query tile is large, several
    // 256-thread blocks per CU otherwise; waves take groups in a grid-stride loop.
    int threads, per_cu;
    if (lds > 40 * 1024) {
        threads = 512;
        per_cu = 1;
    } else {
        threads = 256;
        per_cu = (int)std::min<size_t>(8, (150 * 1024) / std::max<size_t>(lds, 1));
    }
    const int64_t waves_per_block = threads / 64;
    int64_t blocks = std::min<int64_t>((n_groups + waves_per_block - 1) / waves_per_block,
                                       (int64_t)scan_cus(h) * per_cu);
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds, s, h->tiles.as<chunk_t>(), qfrag,
                       h->scale.as<float>(), mask, gmax, nq, h->KT, h->n_rows, n_groups);
    return hipGetLastError();
}

template <typename STORE>
hipError_t launch_scan_g(const hr_index* h, hipStream_t s, int G, const chunk_t* qfrag, const uint8_t* mask,
                         float* gmax, int nq, int64_t n_groups) {
    // n_groups here counts SUPER-groups (64 rows) = the scan's loop bound
    if (group_rows_for(h, h->n_rows) == 16) {
        switch (G) {
            case 1: return launch_scan<STORE, 1, 1>(h, s, qfrag, mask, gmax, nq, n_groups);
            case 2: return launch_scan<STORE, 2, 1>(h, s, qfrag, mask, gmax, nq, n_groups);
            case 3: return launch_scan<STORE, 3, 1>(h, s, qfrag, mask, gmax, nq, n_groups);
            default: return launch_scan<STORE, 4, 1>(h, s, qfrag, mask, gmax, nq, n_groups);
        }
    }
    switch (G) {
        case 1: return launch_scan<STORE, 1, 4>(h, s, qfrag, mask, gmax, nq, n_groups);
        case 2: return launch_scan<STORE, 2, 4>(h, s, qfrag, mask, gmax, nq, n_groups);
        case 3: return launch_scan<STORE, 3, 4>(h, s, qfrag, mask, gmax, nq, n_groups);
        default: return launch_scan<STORE, 4, 4>(h, s, qfrag, mask, gmax, nq, n_groups);
    }
}

// Large-batch pass (dense_scan_bigq_kernel): GQ query groups streamed through LDS in k-chunks.
template <typename STORE, int GQ, int NRB>
hipError_t launch_scan_bigq(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                            int nq, int64_t n_super) {
    auto kern = dense_scan_bigq_kernel<STORE, GQ, NRB>;
    const size_t lds = (size_t)2 * GQ * 2 * 1024;  // 2 buffers x GQ groups x BKT(2) fragments of 1 KiB
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int64_t blocks = std::min<int64_t>((n_super + 7) / 8, (int64_t)scan_cus(h));
    blocks = std::max<int64_t>(blocks, 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, h->tiles.as<chunk_t>(), qfrag,
                       h->scale.as<float>(), mask, gmax, nq, h->KT, h->n_rows, n_super);
    return hipGetLastError();
}
template <typename STORE>
hipError_t launch_scan_bigq_g(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask,
                              float* gmax, int nq, int64_t n_super) {
    return group_rows_for(h, h->n_rows) == 16 ? launch_scan_bigq<STORE, 8, 1>(h, s, qfrag, mask, gmax, nq, n_super)
                                              : launch_scan_bigq<STORE, 8, 4>(h, s, qfrag, mask, gmax, nq, n_super);
}

// 256-query pass with the queries in registers and the corpus streamed through LDS (dense_scan_qreg_kernel):
// fp16 shards whose rows are 24 tiles long (D = 768 after padding; 2 x 24 x 4 fragment registers per wave).
// hr_debug_option(HR_DEBUG_DENSE_KERNELS) bit mask (tests drive every scan kernel at every shape): 1 = no register-
// resident 256-query pass, 2 = no tiled-contraction pass, 4 = no k-chunked large-batch pass, 8 = prefer the tiled
// contraction to the register-resident pass where both apply
int g_dense_kernels = 0;
int g_sparse_rpb = 0;     // HR_DEBUG_SPARSE_RPB: doc ranges per sparse-scan block (0 = by shard size)
int g_no_trim = 0;        // HR_DEBUG_NO_TRIM: 1 = refine all C candidate groups (A/B of the data-dependent candidate set)
int g_group_rows = 0;     // HR_DEBUG_GROUP_ROWS: candidate-group size of handles created from now on (0 = by shard size)
bool qreg_supported(const hr_index* h) {
    return !(g_dense_kernels & 1) && h->dtype == HR_F16 && h->KT == 24;
}
template <int KT, int NRB, int GW, int NW>
hipError_t launch_scan_qreg(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                            int nq, int64_t n_super) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(n_super, (int64_t)scan_cus(h) * (8 / NW)));
    hipLaunchKernelGGL((dense_scan_qreg_kernel<KT, NRB, GW, NW>), dim3((unsigned)blocks), dim3(64 * NW), 0, s,
                       h->tiles.as<chunk_t>(), qfrag, h->scale.as<float>(), mask, gmax, nq, h->n_rows, n_super);
    return hipGetLastError();
}
// 256 queries per pass: 8 waves x 32 queries, one block per CU
hipError_t launch_scan_qreg_g(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                              int nq, int64_t n_super) {
    return group_rows_for(h, h->n_rows) == 16 ? launch_scan_qreg<24, 1, 2, 8>(h, s, qfrag, mask, gmax, nq, n_super)
                                              : launch_scan_qreg<24, 4, 2, 8>(h, s, qfrag, mask, gmax, nq, n_super);
}
// 256 queries per pass, second form: 4 waves x 64 queries in registers (one wave per SIMD, the whole 512-entry register
// file), the corpus through the same LDS-DMA ring with half the LDS reads (dense_scan_q64_kernel).
// HR_DEBUG_DENSE_KERNELS bit 16 selects it.
hipError_t launch_scan_q64_g(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                             int nq, int64_t n_super) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(n_super, (int64_t)scan_cus(h)));
    const size_t lds = (size_t)kQregStages * 24 * 1024 + 2 * kSuperRows * sizeof(float);
    static bool ready = false;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)dense_scan_q64_kernel<24, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)dense_scan_q64_kernel<24, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ready = true;
    }
    if (group_rows_for(h, h->n_rows) == 16)
        hipLaunchKernelGGL((dense_scan_q64_kernel<24, 1>), dim3((unsigned)blocks), dim3(256), lds, s, h->tiles.as<chunk_t>(), qfrag,
                           h->scale.as<float>(), mask, gmax, nq, h->n_rows, n_super);
    else
        hipLaunchKernelGGL((dense_scan_q64_kernel<24, 4>), dim3((unsigned)blocks), dim3(256), lds, s, h->tiles.as<chunk_t>(), qfrag,
                           h->scale.as<float>(), mask, gmax, nq, h->n_rows, n_super);
    return hipGetLastError();
}
// 256-query pass as a tiled contraction (dense_scan_gemm_kernel): fp16 shards of any row length from 8 tiles up;
// serves the shapes the register-resident form cannot (D = 1024: BASELINE config 5).
bool gemm_supported(const hr_index* h) {
    return !(g_dense_kernels & 2) && h->dtype == HR_F16 && h->KT >= 8;
}
template <int GQ>
hipError_t launch_scan_gemm_g(const hr_index* h, hipStream_t s, const chunk_t* qfrag, const uint8_t* mask, float* gmax,
                              int nq, int64_t n_super) {
    const int64_t n_tiles = (n_super * kRowBlocksPerSuper + kGemmRowBlocks - 1) / kGemmRowBlocks;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_tiles, scan_cus(h)));
    if (group_rows_for(h, h->n_rows) == 16)
        hipLaunchKernelGGL((dense_scan_gemm_kernel<GQ, 1>), dim3(blocks), dim3(512), 0, s, h->tiles.as<chunk_t>(), qfrag,
                           h->scale.as<float>(), mask, gmax, nq, h->KT, h->n_rows, n_super);
    else
        hipLaunchKernelGGL((dense_scan_gemm_kernel<GQ, 4>), dim3(blocks), dim3(512), 0, s, h->tiles.as<chunk_t>(), qfrag,
                           h->scale.as<float>(), mask, gmax, nq, h->KT, h->n_rows, n_super);
    return hipGetLastError();
}
int max_groups_for_dim(const hr_index* h) {
    // query tile must fit LDS: G * KT KiB <= 144 KiB
    int g = 156 / std::max(h->KT, 1);
    return std::max(1, std::min(4, g));
}

// Two-level candidate selection: per-bucket maxima, then one block per query — for one modality or for both
// modalities of a hybrid search in one pair of launches (select.h: GroupSelPair).
int group_sel_args(hr_index* h, Workspace* ws, int B, int64_t n_groups, int C, GroupSelArgs* a, const TopkArgs* t = nullptr) {
    const int64_t n_buckets = (n_groups + kBucketGroups - 1) / kBucketGroups;
    HIP_TRY(h, ws->bmax.ensure((size_t)B * n_buckets * sizeof(float)));
    a->gmax = ws->gmax.as<float>();
    a->bmax = ws->bmax.as<float>();
    a->n_groups = n_groups;
    a->n_buckets = n_buckets;
    a->C = C;
    a->two_level = n_groups > C && n_buckets > C;
    a->cand = ws->cand.as<int32_t>();
    a->a_cut = ws->acut.as<float>();
    // the data-dependent candidate set: trim against the K-th largest group maximum with the error bound of the list's proof
    a->K_trim = (t && !g_no_trim) ? t->K : 0;
    a->eps_abs = t ? t->eps_abs : 0.f;
    a->eps_rel = t ? t->eps_rel : 0.f;
    a->eps_abs_q = t ? t->eps_abs_q : nullptr;
    return HR_OK;
}
int launch_group_select_pair(hr_index* h, hipStream_t s, int B, const GroupSelPair& p) {
    int64_t blocks = 0;
    for (int m = 0; m < p.n; ++m)
        if (p.m[m].two_level) blocks = std::max(blocks, (p.m[m].n_buckets + 4 * kBucketsPerWave - 1) / (4 * kBucketsPerWave));
    if (blocks > 0) {
        hipLaunchKernelGGL(bucket_max_kernel, dim3((unsigned)blocks, B, p.n), dim3(256), 0, s, p);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(select_groups_kernel, dim3(B, p.n), dim3(1024), 0, s, p);
    HIP_TRY(h, hipGetLastError());
    return HR_OK;
}
int launch_group_select(hr_index* h, Workspace* ws, hipStream_t s, int B, int64_t n_groups, int C, const TopkArgs* t = nullptr) {
    GroupSelPair p{};
    p.n = 1;
    HR_TRY(group_sel_args(h, ws, B, n_groups, C, &p.m[0], t));
    return launch_group_select_pair(h, s, B, p);
}
int launch_topk_pair(hr_index* h, hipStream_t s, int B, const TopkPair& p) {
    hipLaunchKernelGGL(select_topk_kernel, dim3(B, p.n), dim3(1024), 0, s, p);
    HIP_TRY(h, hipGetLastError());
    return HR_OK;
}

// Escalation ladder of the host forms when a list is not provably exact:
// default -> 4x (bounded by the selection kernel's bucket table) -> every group.
constexpr int kMaxSelectGroups = 448;
int next_candidate_count(int C, int64_t n_groups) {
    const int64_t all = round_up(n_groups, 16);
    if (C < kMaxSelectGroups && (int64_t)C * 4 < all) return std::min(C * 4, kMaxSelectGroups);
    return (int)all;
}

int candidate_groups_for_k(int k) {
    int c = k + std::max(16, k / 2);
    return (int)round_up(c, 16);
}

void dense_eps(const hr_index* h, float* eps_abs, int* norm_mode) {
    const double Dp = (double)h->KT * 4 * elems_per_chunk(h->dtype);
    double unit = 2.0 * Dp * std::ldexp(1.0, -24) + 1e-6;        // fp32 accumulation + scale rounding
    unit += (h->dtype == HR_F16) ? std::ldexp(1.0, -11) * 1.01  // query rounded to fp16
                                 : std::ldexp(1.0, -22);        // query normalised in fp32
    if (h->metric == HR_METRIC_COSINE) {
        *eps_abs = (float)unit;
        *norm_mode = 0;
    } else {
        *eps_abs = (float)(unit * (double)h->max_row_norm * 1.0001);
        *norm_mode = 1;
    }
}

// ---- finishing steps shared by the single-modality chains and the hybrid chain --------------------------------
int launch_refine_dense(hr_index* h, Workspace* ws, hipStream_t s, const float* d_q, int B, int C, int GR,
                        const uint8_t* d_mask) {
    const int cosine = h->metric == HR_METRIC_COSINE;
    if (h->dtype == HR_F16)
        hipLaunchKernelGGL((refine_dense_kernel<_Float16>), dim3((C * GR + 63) / 64, B), dim3(64), 0, s,
                           h->tiles.as<chunk_t>(), h->KT, (int)h->dim, d_q, ws->qn2.as<double>(),
                           h->norm2.as<double>(), d_mask, ws->cand.as<int32_t>(), C, GR, h->n_rows, cosine,
                           ws->cscore.as<float>(), ws->crow.as<int32_t>());
    else
        hipLaunchKernelGGL((refine_dense_kernel<float>), dim3((C * GR + 63) / 64, B), dim3(64), 0, s,
                           h->tiles.as<chunk_t>(), h->KT, (int)h->dim, d_q, ws->qn2.as<double>(),
                           h->norm2.as<double>(), d_mask, ws->cand.as<int32_t>(), C, GR, h->n_rows, cosine,
                           ws->cscore.as<float>(), ws->crow.as<int32_t>());
    HIP_TRY(h, hipGetLastError());
    return HR_OK;
}
TopkArgs dense_topk_args(const hr_index* h, Workspace* ws, int C, int GR, int k, int64_t* d_ids, float* d_scores,
                         int32_t* d_flags) {
    float eps_abs;
    int norm_mode;
    dense_eps(h, &eps_abs, &norm_mode);
    TopkArgs a{};
    a.cscore = ws->cscore.as<float>();
    a.crow = ws->crow.as<int32_t>();
    a.n = C * GR;
    a.K = k;
    a.row_offset = h->row_offset;
    a.a_cut = ws->acut.as<float>();
    a.cut_floor = -INFINITY;
    a.eps_abs = eps_abs;
    a.eps_abs_q = nullptr;
    a.eps_rel = 0.0f;
    a.norm_mode = norm_mode;
    a.qn2 = ws->qn2.as<double>();
    a.out_ids = d_ids;
    a.out_scores = d_scores;
    a.flags = d_flags;
    return a;
}
int launch_refine_sparse(hr_index* h, Workspace* ws, hipStream_t s, const int64_t* d_qptr, const int32_t* d_qidx,
                         const float* d_qval, int B, int C, int GR, int stride, const uint8_t* d_mask) {
    // 64 docs per wave: shorter chains finish this kernel sooner (0.60 -> 0.55 ms at 10M docs, 0.168 -> 0.136 ms at 1.25M
    // with 16) but the step does not gain — the kernel runs beside the scans, and what it takes from the HBM sooner they
    // get later (round 2 A/B).
    const int dpw = 64;
    const bool hashed = stride <= kHashMaxTerms;   // the query's terms as a hash table behind the filter (sparse.h)
    const size_t lds = (size_t)kFilterBits / 8 + (hashed ? (size_t)sparse_hash_slots(stride) * 8 : (size_t)stride * 8);
    hipLaunchKernelGGL(hashed ? refine_sparse_kernel<true> : refine_sparse_kernel<false>,
                       dim3((C * GR + 4 * dpw - 1) / (4 * dpw), B), dim3(256), lds, s, h->s_indptr.as<int64_t>(),
                       h->s_idx.as<int32_t>(), h->s_val.as<float>(), d_qptr, d_qidx, d_qval, d_mask,
                       ws->cand.as<int32_t>(), C, GR, h->n_sparse, stride, dpw, ws->cscore.as<float>(),
                       ws->crow.as<int32_t>());
    HIP_TRY(h, hipGetLastError());
    return HR_OK;
}
TopkArgs sparse_topk_args(const hr_index* h, Workspace* ws, int C, int GR, int k, int64_t* d_ids, float* d_scores,
                          int32_t* d_flags) {
    // scan error = fixed-point rounding ((nnz+1)/scale per query, from the prep kernel) + fp32 rounding of w*scale, of
    // the product and of the int->float conversion (relative, 2^-22 with margin).
    TopkArgs a{};
    a.cscore = ws->cscore.as<float>();
    a.crow = ws->crow.as<int32_t>();
    a.n = C * GR;
    a.K = k;
    a.row_offset = h->row_offset;
    a.a_cut = ws->acut.as<float>();
    a.cut_floor = 0.0f;
    a.eps_abs = 0.0f;
    a.eps_abs_q = ws->qeps.as<float>();
    a.eps_rel = (float)(std::ldexp(1.0, -11) * 1.01 + std::ldexp(1.0, -22));  // fp16 posting weights
    a.norm_mode = 0;
    a.qn2 = nullptr;
    a.out_ids = d_ids;
    a.out_scores = d_scores;
    a.flags = d_flags;
    return a;
}

// Enqueue a complete dense search on stream s.  All pointers are device pointers.
enum { PHASE_SCAN = 1, PHASE_FINISH = 2, PHASE_PREP = 4, PHASE_ALL = 7 };

// ---- the finishing chain as one launch (finish.h) --------------------------------------------------------------
// One block per (query, modality): worth it when the batch fills a good part of the chip that way and the candidate
// set fits LDS; otherwise the multi-launch chain, whose refine kernels spread ONE query over many compute units
// (single-query latency path, escalated searches with hundreds of candidate groups).
int g_finish_mode = 0;  // hr_debug_option(HR_DEBUG_FINISH_MODE): 0 = by batch size, 1 = always the chain, 2 = fused whenever it fits
bool finish_fused_ok(int B, int n_mod, int C, int GR, int64_t n_groups) {
    const int64_t n_buckets = (n_groups + kBucketGroups - 1) / kBucketGroups;
    if (g_finish_mode == 1) return false;
    return ((int64_t)B * n_mod >= 64 || g_finish_mode == 2) && C <= kFinishMaxCand && (int64_t)C * GR <= kFinishMaxSlots &&
           n_buckets <= kFinishMaxBuckets;
}
int launch_finish(hr_index* h, hipStream_t s, int B, FinishPair& p) {
    p.key_slots = 0;
    for (int i = 0; i < p.n; ++i) p.key_slots = std::max(p.key_slots, p.m[i].sel.C * p.m[i].group_rows);
    const size_t lds = finish_lds_bytes(p);
    static bool attr_set = false;  // benign race: the attribute is idempotent
    if (!attr_set) {
        HIP_TRY(h, hipFuncSetAttribute((const void*)finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(finish_kernel, dim3(B, p.n), dim3(kFinishThreads), lds, s, p);
    HIP_TRY(h, hipGetLastError());
    return HR_OK;
}

FinishMod finish_mod_dense(hr_index* h, Workspace* ws, const float* d_q, int B, int C, int GR, int64_t n_groups, int k,
                           const uint8_t* d_mask, int64_t* d_ids, float* d_scores, int32_t* d_flags) {
    FinishMod m{};
    m.kind = 0;
    m.group_rows = GR;
    m.sel.gmax = ws->gmax.as<float>();
    m.sel.n_groups = n_groups;
    m.sel.n_buckets = (n_groups + kBucketGroups - 1) / kBucketGroups;
    m.sel.C = C;
    m.sel.two_level = n_groups > C && m.sel.n_buckets > C;
    m.sel.a_cut = ws->acut.as<float>();
    m.topk = dense_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
    m.sel.K_trim = g_no_trim ? 0 : k;
    m.sel.eps_abs = m.topk.eps_abs;
    m.sel.eps_rel = m.topk.eps_rel;
    m.sel.eps_abs_q = m.topk.eps_abs_q;
    m.rowmask = d_mask;
    m.tiles = h->tiles.as<chunk_t>();
    m.KT = h->KT;
    m.dim = (int)h->dim;
    m.cosine = h->metric == HR_METRIC_COSINE;
    m.dtype = h->dtype;
    m.q = d_q;
    m.qn2 = ws->qn2.as<double>();
    m.norm2 = h->norm2.as<double>();
    m.n_rows = h->n_rows;
    return m;
}
FinishMod finish_mod_sparse(hr_index* h, Workspace* ws, const int64_t* d_qptr, const int32_t* d_qidx, const float* d_qval,
                            int B, int C, int GR, int64_t n_groups, int stride, int k, const uint8_t* d_mask,
                            int64_t* d_ids, float* d_scores, int32_t* d_flags) {
    FinishMod m{};
    m.kind = 1;
    m.group_rows = GR;
    m.sel.gmax = ws->gmax.as<float>();
    m.sel.n_groups = n_groups;
    m.sel.n_buckets = (n_groups + kBucketGroups - 1) / kBucketGroups;
    m.sel.C = C;
    m.sel.two_level = n_groups > C && m.sel.n_buckets > C;
    m.sel.a_cut = ws->acut.as<float>();
    m.topk = sparse_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
    m.sel.K_trim = g_no_trim ? 0 : k;
    m.sel.eps_abs = m.topk.eps_abs;
    m.sel.eps_rel = m.topk.eps_rel;
    m.sel.eps_abs_q = m.topk.eps_abs_q;
    m.rowmask = d_mask;
    m.indptr = h->s_indptr.as<int64_t>();
    m.idx = h->s_idx.as<int32_t>();
    m.val = h->s_val.as<float>();
    m.q_indptr = d_qptr;
    m.q_idx = d_qidx;
    m.q_val = d_qval;
    m.q_cap = stride;
    m.n_rows = h->n_sparse;
    return m;
}

// Enqueue a dense search on stream s: PHASE_PREP = query prep, PHASE_SCAN = the shard scan (leaves the group
// maxima in ws), PHASE_FINISH = candidate select + refine + top-k from those maxima.
int dense_search_enqueue(hr_index* h, Workspace* ws, hipStream_t s, const float* d_q, int B, int k,
                         const uint8_t* d_mask, int64_t* d_ids, float* d_scores, int32_t* d_flags, int C,
                         hipEvent_t scan_done = nullptr, int phases = PHASE_ALL) {
    const int GR = group_rows_for(h, h->n_rows);
    const int64_t n_super = (h->n_rows + kSuperRows - 1) / kSuperRows;
    const int64_t n_groups = n_super * (kSuperRows / GR);  // group maxima per query (tail groups hold -inf)
    const int Gsmall = max_groups_for_dim(h);
    // batches beyond what fits LDS whole go through the k-chunked large-batch pass, 128 or 256 queries at a time
    const bool big = B > 16 * Gsmall && h->KT % 4 == 0 && !(g_dense_kernels & 4);
    const bool prefer_gemm = (g_dense_kernels & 8) != 0;
    const bool use_qreg = qreg_supported(h) && !(prefer_gemm && gemm_supported(h));
    const bool big256 = big && B > 128 && (use_qreg || gemm_supported(h));   // 256 queries per pass
    const int Gmax = big256 ? 16 : big ? 8 : Gsmall;
    const int chunk_q = 16 * Gmax;
    const int n_chunks = (B + chunk_q - 1) / chunk_q;
    const size_t chunk_frag = (size_t)Gmax * h->KT * kTileChunks;   // 16-byte chunks of one pass's query fragments
    HIP_TRY(h, ws->qfrag.ensure((size_t)n_chunks * chunk_frag * sizeof(chunk_t)));
    HIP_TRY(h, ws->qn2.ensure((size_t)B * sizeof(double)));
    HIP_TRY(h, ws->gmax.ensure((size_t)B * n_groups * sizeof(float)));
    HIP_TRY(h, ws->cand.ensure((size_t)B * C * sizeof(int32_t)));
    HIP_TRY(h, ws->acut.ensure((size_t)B * sizeof(float)));
    HIP_TRY(h, ws->cscore.ensure((size_t)B * C * GR * sizeof(float)));
    HIP_TRY(h, ws->crow.ensure((size_t)B * C * GR * sizeof(int32_t)));

    auto groups_of = [&](int nq) { return big ? ((big256 && nq > 128) ? 16 : 8) : (nq + 15) / 16; };
    if (phases & PHASE_PREP) {
        // every pass's queries in one launch: pass c owns fragment groups [c * Gmax, ...) of qfrag (slot = query number)
        Span sp(h, s, PH_PREP);
        const int G_total = (n_chunks - 1) * Gmax + groups_of(B - (n_chunks - 1) * chunk_q);
        if (h->dtype == HR_F16)
            hipLaunchKernelGGL((prep_queries_kernel<_Float16>), dim3(16 * G_total), dim3(256), 0, s, d_q, B, (int)h->dim,
                               h->KT, ws->qfrag.as<chunk_t>(), ws->qn2.as<double>());
        else
            hipLaunchKernelGGL((prep_queries_kernel<float>), dim3(16 * G_total), dim3(256), 0, s, d_q, B, (int)h->dim,
                               h->KT, ws->qfrag.as<chunk_t>(), ws->qn2.as<double>());
        HIP_TRY(h, hipGetLastError());
    }
    for (int c0 = 0; (phases & PHASE_SCAN) && c0 < B; c0 += chunk_q) {
        const int nq = std::min(chunk_q, B - c0);
        const bool pass256 = big256 && nq > 128;   // a trailing chunk of <= 128 queries takes the 128-query pass
        const int G = groups_of(nq);
        const chunk_t* qf = ws->qfrag.as<chunk_t>() + (size_t)(c0 / chunk_q) * chunk_frag;
        {
            Span sp(h, s, PH_SCAN);
            float* gm = ws->gmax.as<float>() + (int64_t)c0 * n_groups;
            hipError_t e;
            if (pass256)
                e = (use_qreg && (g_dense_kernels & 16)) ? launch_scan_q64_g(h, s, qf, d_mask, gm, nq, n_super)
                    : use_qreg ? launch_scan_qreg_g(h, s, qf, d_mask, gm, nq, n_super)
                               : launch_scan_gemm_g<16>(h, s, qf, d_mask, gm, nq, n_super);
            else if (big)
                e = (h->dtype == HR_F16) ? launch_scan_bigq_g<_Float16>(h, s, qf, d_mask, gm, nq, n_super)
                                         : launch_scan_bigq_g<float>(h, s, qf, d_mask, gm, nq, n_super);
            else
                e = (h->dtype == HR_F16) ? launch_scan_g<_Float16>(h, s, G, qf, d_mask, gm, nq, n_super)
                                         : launch_scan_g<float>(h, s, G, qf, d_mask, gm, nq, n_super);
            HIP_TRY(h, e);
        }
    }
    if (scan_done) HIP_TRY(h, hipEventRecord(scan_done, s));
    if (!(phases & PHASE_FINISH)) return HR_OK;
    if (finish_fused_ok(B, 1, C, GR, n_groups)) {
        Span sp(h, s, PH_FINISH);
        FinishPair p{};
        p.n = 1;
        p.m[0] = finish_mod_dense(h, ws, d_q, B, C, GR, n_groups, k, d_mask, d_ids, d_scores, d_flags);
        return launch_finish(h, s, B, p);
    }
    {
        Span sp(h, s, PH_GSEL);
        const TopkArgs t = dense_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
        HR_TRY(launch_group_select(h, ws, s, B, n_groups, C, &t));
    }
    {
        Span sp(h, s, PH_REFINE);
        HR_TRY(launch_refine_dense(h, ws, s, d_q, B, C, GR, d_mask));
    }
    {
        Span sp(h, s, PH_TOPK);
        TopkPair p{};
        p.n = 1;
        p.m[0] = dense_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
        HR_TRY(launch_topk_pair(h, s, B, p));
    }
    return HR_OK;
}

// Doc ranges one scan block walks (sparse_scan_kernel pipelines over them): as many as leave the chip
// about six rounds of blocks (two blocks per CU), at most 16; gridDim.y must stay below 65536.
int sparse_ranges_per_block(const hr_index* h, int B) {
    const int forced = g_sparse_rpb;
    const int64_t pairs = (int64_t)B * h->n_ranges;
    int64_t rpb = forced > 0 ? forced : std::min<int64_t>(16, pairs / (6 * 2 * (int64_t)scan_cus(h)));
    rpb = std::max<int64_t>(rpb, (h->n_ranges + 65534) / 65535);
    return (int)std::max<int64_t>(1, rpb);
}

int sparse_search_enqueue(hr_index* h, Workspace* ws, hipStream_t s, const int64_t* d_qptr, const int32_t* d_qidx,
                          const float* d_qval, int B, int max_q_nnz, int k, const uint8_t* d_mask, int64_t* d_ids,
                          float* d_scores, int32_t* d_flags, int C, int phases = PHASE_ALL) {
    const int GR = group_rows_for(h, h->n_sparse);
    const int64_t n_groups = (h->n_sparse + GR - 1) / GR;
    HIP_TRY(h, ws->gmax.ensure((size_t)B * n_groups * sizeof(float)));
    HIP_TRY(h, ws->cand.ensure((size_t)B * C * sizeof(int32_t)));
    HIP_TRY(h, ws->acut.ensure((size_t)B * sizeof(float)));
    HIP_TRY(h, ws->cscore.ensure((size_t)B * C * GR * sizeof(float)));
    HIP_TRY(h, ws->crow.ensure((size_t)B * C * GR * sizeof(int32_t)));
    HIP_TRY(h, ws->qscale.ensure((size_t)B * sizeof(float)));
    HIP_TRY(h, ws->qeps.ensure((size_t)B * sizeof(float)));
    const int64_t V1 = h->sparse_dim + 1;
    const int stride = (int)round_up(std::max(max_q_nnz, 1), 64);  // fixed-stride query layout for the scan
    if (phases & PHASE_PREP) {
        HIP_TRY(h, ws->pq_n.ensure((size_t)B * 4));
        HIP_TRY(h, ws->pq_idx.ensure((size_t)B * stride * 4));
        HIP_TRY(h, ws->pq_w.ensure((size_t)B * stride * 4));
        hipLaunchKernelGGL(sparse_query_prep_kernel, dim3(B), dim3(256), 0, s, d_qptr, d_qidx, d_qval,
                           h->max_sparse_abs, stride, (int)h->sparse_dim, ws->qscale.as<float>(), ws->qeps.as<float>(),
                           ws->pq_n.as<int32_t>(), ws->pq_idx.as<int32_t>(), ws->pq_w.as<float>());
        HIP_TRY(h, hipGetLastError());
    }
    if (phases & PHASE_SCAN) {
        Span sp(h, s, PH_SSCAN);
        const int rpb = sparse_ranges_per_block(h, B);
        const unsigned chunks = (unsigned)((h->n_ranges + rpb - 1) / rpb);
        if (h->n_dense_runs > 0)
            hipLaunchKernelGGL(sparse_scan_kernel<true>, dim3((unsigned)B, chunks), dim3(kScanThreads), 0, s,
                               h->rt_off.as<unsigned int>(), V1, h->range_base.as<int64_t>(), h->post.as<uint32_t>(),
                               ws->pq_n.as<int32_t>(), ws->pq_idx.as<int32_t>(), ws->pq_w.as<float>(), stride,
                               ws->qscale.as<float>(), d_mask, h->n_sparse, n_groups, GR, h->n_ranges, rpb,
                               h->idle_post.as<uint32_t>(), ws->gmax.as<float>());
        else
            hipLaunchKernelGGL(sparse_scan_kernel<false>, dim3((unsigned)B, chunks), dim3(kScanThreads), 0, s,
                               h->rt_off.as<unsigned int>(), V1, h->range_base.as<int64_t>(), h->post.as<uint32_t>(),
                               ws->pq_n.as<int32_t>(), ws->pq_idx.as<int32_t>(), ws->pq_w.as<float>(), stride,
                               ws->qscale.as<float>(), d_mask, h->n_sparse, n_groups, GR, h->n_ranges, rpb,
                               h->idle_post.as<uint32_t>(), ws->gmax.as<float>());
        HIP_TRY(h, hipGetLastError());
    }
    if (!(phases & PHASE_FINISH)) return HR_OK;
    if (finish_fused_ok(B, 1, C, GR, n_groups)) {
        Span sp(h, s, PH_FINISH);
        FinishPair p{};
        p.n = 1;
        p.m[0] = finish_mod_sparse(h, ws, d_qptr, d_qidx, d_qval, B, C, GR, n_groups, stride, k, d_mask, d_ids, d_scores, d_flags);
        return launch_finish(h, s, B, p);
    }
    {
        Span sp(h, s, PH_SGSEL);
        const TopkArgs t = sparse_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
        HR_TRY(launch_group_select(h, ws, s, B, n_groups, C, &t));
    }
    {
        Span sp(h, s, PH_SREFINE);
        HR_TRY(launch_refine_sparse(h, ws, s, d_qptr, d_qidx, d_qval, B, C, GR, stride, d_mask));
    }
    {
        Span sp(h, s, PH_STOPK);
        TopkPair p{};
        p.n = 1;
        p.m[0] = sparse_topk_args(h, ws, C, GR, k, d_ids, d_scores, d_flags);
        HR_TRY(launch_topk_pair(h, s, B, p));
    }
    return HR_OK;
}

// Finishing chain of a hybrid search (both modalities non-empty): the same kernels as the two single chains, with the
// selection steps of both modalities in one launch each — 5 dependent launches (+ 2 refines) instead of 8.
// The profiling spans of the merged launches are booked on the dense phases (group_select, topk).
int hybrid_finish_enqueue(hr_index* h, Workspace* wd, Workspace* wsp, hipStream_t s, const float* d_q,
                          const int64_t* d_qptr, const int32_t* d_qidx, const float* d_qval, int B, int max_q_nnz, int k,
                          const uint8_t* d_mask, int64_t* d_ids, float* d_scores, int32_t* d_flags, int64_t* s_ids,
                          float* s_scores, int32_t* s_flags, int C) {
    const int GRd = group_rows_for(h, h->n_rows), GRs = group_rows_for(h, h->n_sparse);
    const int64_t n_super = (h->n_rows + kSuperRows - 1) / kSuperRows;
    const int64_t ng_d = n_super * (kSuperRows / GRd), ng_s = (h->n_sparse + GRs - 1) / GRs;
    const int stride = (int)round_up(std::max(max_q_nnz, 1), 64);
    if (finish_fused_ok(B, 2, C, GRd, ng_d) && finish_fused_ok(B, 2, C, GRs, ng_s)) {
        for (Workspace* ws : {wd, wsp}) HIP_TRY(h, ws->acut.ensure((size_t)B * sizeof(float)));
        Span sp(h, s, PH_FINISH);
        FinishPair p{};
        p.n = 2;
        p.m[0] = finish_mod_dense(h, wd, d_q, B, C, GRd, ng_d, k, d_mask, d_ids, d_scores, d_flags);
        p.m[1] = finish_mod_sparse(h, wsp, d_qptr, d_qidx, d_qval, B, C, GRs, ng_s, stride, k, d_mask, s_ids, s_scores, s_flags);
        return launch_finish(h, s, B, p);
    }
    for (Workspace* ws : {wd, wsp}) {
        const int GR = ws == wd ? GRd : GRs;
        HIP_TRY(h, ws->cand.ensure((size_t)B * C * sizeof(int32_t)));
        HIP_TRY(h, ws->acut.ensure((size_t)B * sizeof(float)));
        HIP_TRY(h, ws->cscore.ensure((size_t)B * C * GR * sizeof(float)));
        HIP_TRY(h, ws->crow.ensure((size_t)B * C * GR * sizeof(int32_t)));
    }
    {
        Span sp(h, s, PH_GSEL);
        GroupSelPair p{};
        p.n = 2;
        const TopkArgs td = dense_topk_args(h, wd, C, GRd, k, d_ids, d_scores, d_flags);
        const TopkArgs ts = sparse_topk_args(h, wsp, C, GRs, k, s_ids, s_scores, s_flags);
        HR_TRY(group_sel_args(h, wd, B, ng_d, C, &p.m[0], &td));
        HR_TRY(group_sel_args(h, wsp, B, ng_s, C, &p.m[1], &ts));
        HR_TRY(launch_group_select_pair(h, s, B, p));
    }
    {
        Span sp(h, s, PH_REFINE);
        HR_TRY(launch_refine_dense(h, wd, s, d_q, B, C, GRd, d_mask));
    }
    {
        Span sp(h, s, PH_SREFINE);
        HR_TRY(launch_refine_sparse(h, wsp, s, d_qptr, d_qidx, d_qval, B, C, GRs, stride, d_mask));
    }
    {
        Span sp(h, s, PH_TOPK);
        TopkPair p{};
        p.n = 2;
        p.m[0] = dense_topk_args(h, wd, C, GRd, k, d_ids, d_scores, d_flags);
        p.m[1] = sparse_topk_args(h, wsp, C, GRs, k, s_ids, s_scores, s_flags);
        HR_TRY(launch_topk_pair(h, s, B, p));
    }
    return HR_OK;
}

int check_search_args(hr_index* h, int B, int k, bool dense) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (B <= 0) return fail(h, HR_EINVAL, "batch size must be positive (got %d)", B);
    if (k <= 0) return fail(h, HR_EINVAL, "k must be positive (got %d)", k);
    if (k > HR_MAX_TOPK) return fail(h, HR_ELIMIT, "k=%d exceeds HR_MAX_TOPK=%d", k, HR_MAX_TOPK);
    if (!h->finalized) return fail(h, HR_ESTATE, "search before hr_finalize");
    if (dense && h->dim == 0) return fail(h, HR_ESTATE, "handle has no dense collection");
    if (!dense && h->sparse_dim == 0) return fail(h, HR_ESTATE, "handle has no sparse collection");
    return HR_OK;
}

// Results for an empty collection: all padding, proven exact.
int fill_empty(hr_index* h, hipStream_t s, int B, int k, int64_t* d_ids, float* d_scores, int32_t* d_flags) {
    HIP_TRY(h, hipMemsetAsync(d_ids, 0xFF, (size_t)B * k * sizeof(int64_t), s));
    HIP_TRY(h, hipMemsetAsync(d_scores, 0, (size_t)B * k * sizeof(float), s));
    if (d_flags) {
        std::vector<int32_t> ones(B, 1);
        HIP_TRY(h, hipMemcpyAsync(d_flags, ones.data(), (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipStreamSynchronize(s));
    }
    return HR_OK;
}

int grow_dense(hr_index* h, int64_t need_rows) {
    if (need_rows <= h->cap_rows) return HR_OK;
    int64_t new_cap = std::max<int64_t>(round_up(need_rows, kSuperRows),
                                        round_up(h->cap_rows + h->cap_rows / 2, kSuperRows));
    new_cap = std::max<int64_t>(new_cap, 1024);
    DevBuf nt, ns, nn;
    const size_t tb = tile_bytes_for_rows(h, new_cap);
    if (hipMalloc(&nt.p, tb) != hipSuccess || hipMalloc(&ns.p, (size_t)new_cap * 4) != hipSuccess ||
        hipMalloc(&nn.p, (size_t)new_cap * 8) != hipSuccess) {
        (void)hipGetLastError();
        nt.cap = ns.cap = nn.cap = 1;
        nt.release(); ns.release(); nn.release();
        return fail(h, HR_ENOMEM, "cannot allocate dense shard for %lld rows (%zu bytes)", (long long)new_cap, tb);
    }
    nt.cap = tb; ns.cap = (size_t)new_cap * 4; nn.cap = (size_t)new_cap * 8;
    hipStream_t s = h->ingest_stream;
    HIP_TRY(h, hipMemsetAsync(nt.p, 0, tb, s));
    HIP_TRY(h, hipMemsetAsync(ns.p, 0, ns.cap, s));
    HIP_TRY(h, hipMemsetAsync(nn.p, 0, nn.cap, s));
    if (h->cap_rows > 0) {
        HIP_TRY(h, hipMemcpyAsync(nt.p, h->tiles.p, tile_bytes_for_rows(h, h->cap_rows), hipMemcpyDeviceToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(ns.p, h->scale.p, (size_t)h->cap_rows * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(nn.p, h->norm2.p, (size_t)h->cap_rows * 8, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    h->tiles.release(); h->scale.release(); h->norm2.release();
    h->tiles = nt; h->scale = ns; h->norm2 = nn;
    h->cap_rows = new_cap;
    return HR_OK;
}

template <typename SRC>
int add_dense_impl(hr_index* h, const SRC* rows, int64_t n, bool src_on_device, hipStream_t user_stream) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (h->dim == 0) return fail(h, HR_ESTATE, "handle has no dense collection");
    if (n < 0 || (n > 0 && !rows)) return fail(h, HR_EINVAL, "bad rows/n");
    if (n == 0) return HR_OK;
    std::unique_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    HR_TRY(grow_dense(h, h->n_rows + n));
    hipStream_t s = src_on_device && user_stream ? user_stream : h->ingest_stream;
    const int64_t chunk_rows = std::max<int64_t>(1, (64ll << 20) / (h->dim * (int64_t)sizeof(SRC)));
    const int kchunks = h->KT * 4;
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const int64_t m = std::min(chunk_rows, n - r0);
        const SRC* src = rows + r0 * h->dim;
        if (!src_on_device) {
            HIP_TRY(h, h->stage.ensure((size_t)m * h->dim * sizeof(SRC)));
            HIP_TRY(h, hipMemcpyAsync(h->stage.p, src, (size_t)m * h->dim * sizeof(SRC), hipMemcpyHostToDevice, s));
            src = h->stage.as<SRC>();
        }
        const int64_t threads = m * kchunks;
        const unsigned blocks = (unsigned)((threads + 255) / 256);
        if (h->dtype == HR_F16) {
            if constexpr (std::is_same<SRC, float>::value)
                hipLaunchKernelGGL((tile_rows_kernel<_Float16, float>), dim3(blocks), dim3(256), 0, s, src, m,
                                   (int)h->dim, h->KT, h->n_rows + r0, h->tiles.as<chunk_t>());
            else
                hipLaunchKernelGGL((tile_rows_kernel<_Float16, _Float16>), dim3(blocks), dim3(256), 0, s,
                                   reinterpret_cast<const _Float16*>(src), m, (int)h->dim, h->KT, h->n_rows + r0,
                                   h->tiles.as<chunk_t>());
        } else {
            hipLaunchKernelGGL((tile_rows_kernel<float, float>), dim3(blocks), dim3(256), 0, s,
                               reinterpret_cast<const float*>(src), m, (int)h->dim, h->KT, h->n_rows + r0,
                               h->tiles.as<chunk_t>());
        }
        HIP_TRY(h, hipGetLastError());
        if (!src_on_device) HIP_TRY(h, hipStreamSynchronize(s));  // staging buffer is reused
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    h->n_rows += n;
    h->finalized = false;
    return HR_OK;
}

// One CSR batch as hr_add_sparse and hr_load accept it: indptr monotone, indices in [0, sparse_dim) and strictly
// ascending inside a row, values finite and within the fp16 posting range.  *batch_max receives max |value|.
int validate_csr(const hr_index* h, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n,
                 float* batch_max) {
    float mx = 0.f;
    for (int64_t r = 0; r < n; ++r) {
        if (indptr[r + 1] < indptr[r]) return fail(h, HR_EINVAL, "indptr not monotone at row %lld", (long long)r);
        int32_t prev = -1;
        for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            const int32_t t = indices[e];
            if (t < 0 || t >= h->sparse_dim)
                return fail(h, HR_EINVAL, "sparse index %d out of range [0,%lld) in row %lld", t, (long long)h->sparse_dim, (long long)r);
            if (t <= prev) return fail(h, HR_EINVAL, "sparse indices must be strictly ascending (row %lld)", (long long)r);
            if (!std::isfinite(values[e])) return fail(h, HR_EINVAL, "non-finite sparse value in row %lld", (long long)r);
            if (std::fabs(values[e]) > 60000.f)
                return fail(h, HR_ELIMIT, "sparse weight %g in row %lld exceeds the fp16 posting range", (double)values[e], (long long)r);
            mx = std::max(mx, std::fabs(values[e]));
            prev = t;
        }
    }
    *batch_max = mx;
    return HR_OK;
}

// Bring the device-side sparse shard up to date with the rows staged since the last call.
// The staged CSR rows are appended to the device CSR (kept for the canonical refine); the range-major postings are
// rebuilt only from the range that holds the first new doc onwards — earlier ranges' posting blocks and offset rows
// stay where they are — so a flush after a small append costs O(batch + one range), not O(corpus), and the host
// keeps no copy of the corpus (reference indexing.py:377-431: insert + flush per index_chunks call).
int build_sparse(hr_index* h) {
    hipStream_t s = h->ingest_stream;
    const int64_t n = h->n_sparse;
    const int64_t V1 = h->sparse_dim + 1;
    if (h->n_csr == n && h->n_sparse_built == n) return HR_OK;
    // 1. append the staged rows to the device CSR.  Restartable: the staging vectors are only read here (the absolute
    //    row pointers are built in a temporary), and the CSR watermark (n_csr, nnz_csr) is committed together with the
    //    release of the staging vectors — a failure before that leaves everything as it was, a failure after it
    //    (step 2) is retried from the device CSR alone.
    if (h->n_csr < n) {
        const int64_t n0 = h->n_csr, nnz0 = h->nnz_csr;
        const int64_t new_nnz = (int64_t)h->pend_idx.size();
        const int64_t nnz = nnz0 + new_nnz;
        if ((int64_t)h->pend_indptr.size() != n - n0 + 1) return fail(h, HR_ESTATE, "sparse staging out of step with the row count");
        std::vector<int64_t> abs_ptr;
        try {
            abs_ptr.resize(h->pend_indptr.size());
        } catch (const std::exception&) {
            return fail(h, HR_ENOMEM, "out of host memory building sparse row pointers");
        }
        for (size_t i = 0; i < abs_ptr.size(); ++i) abs_ptr[i] = h->pend_indptr[i] + nnz0;  // absolute entry numbers
        HIP_TRY(h, h->s_indptr.grow((size_t)(n + 1) * 8, (size_t)(n0 + 1) * 8, s));
        HIP_TRY(h, h->s_idx.grow((size_t)std::max<int64_t>(nnz, 1) * 4, (size_t)nnz0 * 4, s));
        HIP_TRY(h, h->s_val.grow((size_t)std::max<int64_t>(nnz, 1) * 4, (size_t)nnz0 * 4, s));
        HIP_TRY(h, hipMemcpyAsync(h->s_indptr.as<int64_t>() + n0, abs_ptr.data(), (size_t)(n - n0 + 1) * 8,
                                  hipMemcpyHostToDevice, s));
        if (new_nnz) {
            HIP_TRY(h, hipMemcpyAsync(h->s_idx.as<int32_t>() + nnz0, h->pend_idx.data(), (size_t)new_nnz * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(h, hipMemcpyAsync(h->s_val.as<float>() + nnz0, h->pend_val.data(), (size_t)new_nnz * 4, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(h, hipStreamSynchronize(s));  // the staging vectors are released below
        std::vector<int64_t>{0}.swap(h->pend_indptr);
        std::vector<int32_t>().swap(h->pend_idx);
        std::vector<float>().swap(h->pend_val);
        h->n_csr = n;
        h->nnz_csr = nnz;
    }
    if (h->fault_inject == 1) {  // test hook: what an allocation failure in step 2 leaves behind
        h->fault_inject = 0;
        return fail(h, HR_ENOMEM, "injected failure after the CSR upload (hr_debug_inject_fault)");
    }
    // 2. rebuild the posting blocks of ranges r_d .. n_ranges-1 from the device CSR.  Nothing is committed before the
    //    last kernel has finished: a retry starts again from the same r_d (h_range_base[0 .. r_d] is never rewritten).
    const int64_t n0 = h->n_sparse_built;
    const int64_t r_d = n0 / kRangeDocs;          // range of the first new doc
    const int64_t doc0 = r_d * kRangeDocs;
    const int64_t n_ranges = (n + kRangeDocs - 1) / kRangeDocs;
    const int64_t dirty = n_ranges - r_d;
    HIP_TRY(h, h->rt_off.grow((size_t)n_ranges * V1 * 4, (size_t)r_d * V1 * 4, s));
    HIP_TRY(h, hipMemsetAsync(h->rt_off.as<unsigned int>() + r_d * V1, 0, (size_t)dirty * V1 * 4, s));
    const unsigned doc_blocks = (unsigned)((n - doc0 + 255) / 256);
    hipLaunchKernelGGL(sparse_count_kernel, dim3(doc_blocks), dim3(256), 0, s, h->s_indptr.as<int64_t>(),
                       h->s_idx.as<int32_t>(), doc0, n, V1, h->rt_off.as<unsigned int>());
    HIP_TRY(h, hipGetLastError());
    struct Scratch {  // released on every return path
        DevBuf totals, cursor;
        ~Scratch() { totals.release(); cursor.release(); }
    } tmp;
    HIP_TRY(h, tmp.totals.ensure((size_t)dirty * 8));
    hipLaunchKernelGGL(sparse_scan_offsets_kernel, dim3((unsigned)dirty), dim3(1024), 0, s,
                       h->rt_off.as<unsigned int>(), V1, r_d, tmp.totals.as<unsigned long long>());
    HIP_TRY(h, hipGetLastError());
    std::vector<unsigned long long> ht(dirty);
    HIP_TRY(h, hipMemcpyAsync(ht.data(), tmp.totals.p, (size_t)dirty * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    h->h_range_base.resize(n_ranges + 1);
    h->h_range_dense.resize(n_ranges);
    for (int64_t r = r_d; r < n_ranges; ++r) {
        h->h_range_base[r + 1] = h->h_range_base[r] + (int64_t)(ht[r - r_d] & 0xFFFFFFFFull);
        h->h_range_dense[r] = (unsigned)(ht[r - r_d] >> 32);
    }
    // runs are padded to 4 postings (filler postings); + slack: the scan fetches 16 bytes at a time
    HIP_TRY(h, h->post.grow((size_t)h->h_range_base[n_ranges] * 4 + 64, (size_t)h->h_range_base[r_d] * 4, s));
    HIP_TRY(h, h->range_base.grow((size_t)(n_ranges + 1) * 8, 0, s));
    HIP_TRY(h, hipMemcpyAsync(h->range_base.p, h->h_range_base.data(), (size_t)(n_ranges + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, tmp.cursor.ensure((size_t)dirty * V1 * 4));
    HIP_TRY(h, hipMemcpyAsync(tmp.cursor.p, h->rt_off.as<unsigned int>() + r_d * V1, (size_t)dirty * V1 * 4,
                              hipMemcpyDeviceToDevice, s));
    // dense runs (sparse.h) hold one fp16 weight per doc of the range: every word of the rebuilt blocks starts as "absent /
    // absent"; sparse runs and their fillers are overwritten whole by the two kernels below
    {
        const int64_t w0 = h->h_range_base[r_d], w1 = h->h_range_base[n_ranges];
        if (w1 > w0)
            HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)(h->post.as<uint32_t>() + w0), (int)0x80008000u, (size_t)(w1 - w0), s));
    }
    // (the fill kernel permutes docs inside aligned blocks of 128: its grid covers whole blocks)
    hipLaunchKernelGGL(sparse_fill_kernel, dim3((unsigned)((round_up(n - doc0, 128) + 255) / 256)), dim3(256), 0, s, h->s_indptr.as<int64_t>(),
                       h->s_idx.as<int32_t>(), h->s_val.as<float>(), doc0, n, V1, tmp.cursor.as<unsigned int>(),
                       h->rt_off.as<unsigned int>(), h->range_base.as<int64_t>(), h->post.as<uint32_t>());
    HIP_TRY(h, hipGetLastError());
    const int64_t pairs = dirty * h->sparse_dim;
    hipLaunchKernelGGL(sparse_pad_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s,
                       h->rt_off.as<unsigned int>(), tmp.cursor.as<unsigned int>(), V1, r_d, n_ranges,
                       h->range_base.as<int64_t>(), h->post.as<uint32_t>());
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));
    h->n_ranges = n_ranges;
    h->n_sparse_built = n;
    h->n_dense_runs = 0;
    for (unsigned c : h->h_range_dense) h->n_dense_runs += c;
    return HR_OK;
}

}  // namespace

// =============================================================================
extern "C" {

int hr_version(void) { return 10400; }  // 1.4.0: round 4 (per-query fusion weights, encoder layer kernels, head-dim-64 attention)

const char* hr_last_error(const hr_index* h) {
    if (!h) return g_last_error.c_str();
    std::lock_guard<std::mutex> g(h->err_mu);
    g_last_error = h->err;  // hand back a pointer that outlives the lock
    return g_last_error.c_str();
}

int hr_create(int device, int64_t dim, int dtype, int metric, int64_t sparse_dim, hr_index** out) {
    if (!out) return fail(nullptr, HR_EINVAL, "out is null");
    *out = nullptr;
    if (dim < 0 || sparse_dim < 0 || (dim == 0 && sparse_dim == 0))
        return fail(nullptr, HR_EINVAL, "need dim > 0 and/or sparse_dim > 0");
    if (dim > HR_MAX_DIM) return fail(nullptr, HR_ELIMIT, "dim=%lld exceeds HR_MAX_DIM=%d", (long long)dim, HR_MAX_DIM);
    if (dtype != HR_F32 && dtype != HR_F16) return fail(nullptr, HR_EINVAL, "unknown dtype %d", dtype);
    if (metric != HR_METRIC_IP && metric != HR_METRIC_COSINE) return fail(nullptr, HR_EINVAL, "unknown metric %d", metric);
    if (sparse_dim > (1ll << 24)) return fail(nullptr, HR_ELIMIT, "sparse_dim too large");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, HR_EHIP, "no HIP device available (%s); libhbmrag has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= count) return fail(nullptr, HR_EINVAL, "device %d out of range (0..%d)", device, count - 1);
    hr_index* h = new (std::nothrow) hr_index();
    if (!h) return fail(nullptr, HR_ENOMEM, "out of host memory");
    h->device = device;
    h->dim = dim;
    h->dtype = dtype;
    h->metric = metric;
    h->sparse_dim = sparse_dim;
    if (dim > 0) {
        const int tile_elems = 4 * elems_per_chunk(dtype);
        h->KT = (int)round_up((dim + tile_elems - 1) / tile_elems, 4);  // multiple of the scan's prefetch depth
    }
    h->group_rows_override = g_group_rows;
    DeviceGuard dg(device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->cu_count = prop.multiProcessorCount;
    uint32_t idle[256];  // idle postings of scan lane l: weight 0, accumulator pad word l
    for (unsigned l = 0; l < 256; ++l) idle[l] = filler_posting(l / 4);
    if (hipStreamCreateWithFlags(&h->ingest_stream, hipStreamNonBlocking) != hipSuccess ||
        h->max_norm.ensure(4) != hipSuccess || hipMemset(h->max_norm.p, 0, 4) != hipSuccess ||
        h->idle_post.ensure(sizeof idle) != hipSuccess ||
        hipMemcpy(h->idle_post.p, idle, sizeof idle, hipMemcpyHostToDevice) != hipSuccess) {
        int rc = fail(nullptr, HR_EHIP, "device %d initialisation failed: %s", device, hipGetErrorString(hipGetLastError()));
        delete h;
        return rc;
    }
    *out = h;
    return HR_OK;
}

void hr_destroy(hr_index* h) {
    if (!h) return;
    {
        DeviceGuard dg(h->device);
        (void)hipDeviceSynchronize();
        for (Workspace* w : h->free_ws) { w->release(); delete w; }
        for (auto& kv : h->stream_ws) { kv.second->release(); delete kv.second; }
        for (auto& pair : h->slot_ws)
            for (Workspace* w : pair)
                if (w) { w->release(); delete w; }
        for (auto& sp : h->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
        for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
        for (DevBuf* b : {&h->tiles, &h->scale, &h->norm2, &h->max_norm, &h->stage, &h->s_indptr, &h->s_idx, &h->s_val,
                          &h->rt_off, &h->range_base, &h->post, &h->idle_post})
            b->release();
        if (h->ingest_stream) (void)hipStreamDestroy(h->ingest_stream);
    }
    delete h;
}

int hr_set_row_offset(hr_index* h, int64_t first_row) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (first_row < 0) return fail(h, HR_EINVAL, "row offset must be >= 0");
    h->row_offset = first_row;
    return HR_OK;
}

int hr_reserve(hr_index* h, int64_t n_rows) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (h->dim == 0) return fail(h, HR_ESTATE, "handle has no dense collection");
    if (n_rows < 0 || n_rows > (1ll << 31) - 64) return fail(h, HR_ELIMIT, "row count out of range");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    if (n_rows <= h->cap_rows) return HR_OK;
    // exact-size allocation: temporarily defeat the 1.5x growth policy
    const int64_t want = round_up(n_rows, kSuperRows);
    const int64_t saved = h->cap_rows;
    if (saved == 0) {
        h->cap_rows = 0;
        DevBuf nt, ns, nn;
        const size_t tb = tile_bytes_for_rows(h, std::max<int64_t>(want, kSuperRows));
        const int64_t rows = std::max<int64_t>(want, kSuperRows);
        if (hipMalloc(&nt.p, tb) != hipSuccess || hipMalloc(&ns.p, (size_t)rows * 4) != hipSuccess ||
            hipMalloc(&nn.p, (size_t)rows * 8) != hipSuccess) {
            (void)hipGetLastError();
            nt.cap = ns.cap = nn.cap = 1;
            nt.release(); ns.release(); nn.release();
            return fail(h, HR_ENOMEM, "cannot reserve %lld rows", (long long)n_rows);
        }
        nt.cap = tb; ns.cap = (size_t)rows * 4; nn.cap = (size_t)rows * 8;
        HIP_TRY(h, hipMemsetAsync(nt.p, 0, tb, h->ingest_stream));
        HIP_TRY(h, hipMemsetAsync(ns.p, 0, ns.cap, h->ingest_stream));
        HIP_TRY(h, hipMemsetAsync(nn.p, 0, nn.cap, h->ingest_stream));
        HIP_TRY(h, hipStreamSynchronize(h->ingest_stream));
        h->tiles = nt; h->scale = ns; h->norm2 = nn;
        h->cap_rows = rows;
        return HR_OK;
    }
    return grow_dense(h, want);
}

int hr_add_dense(hr_index* h, const float* rows, int64_t n) {
    return add_dense_impl<float>(h, rows, n, false, nullptr);
}

int hr_add_dense_raw(hr_index* h, const void* rows, int64_t n) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (h->dtype == HR_F16) return add_dense_impl<_Float16>(h, static_cast<const _Float16*>(rows), n, false, nullptr);
    return add_dense_impl<float>(h, static_cast<const float*>(rows), n, false, nullptr);
}

int hr_add_dense_raw_dev(hr_index* h, const void* d_rows, int64_t n, void* stream) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (h->dtype == HR_F16)
        return add_dense_impl<_Float16>(h, static_cast<const _Float16*>(d_rows), n, true, (hipStream_t)stream);
    return add_dense_impl<float>(h, static_cast<const float*>(d_rows), n, true, (hipStream_t)stream);
}

int hr_add_sparse(hr_index* h, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (h->sparse_dim == 0) return fail(h, HR_ESTATE, "handle has no sparse collection");
    if (n < 0 || (n > 0 && !indptr)) return fail(h, HR_EINVAL, "bad indptr/n");
    if (n == 0) return HR_OK;
    const int64_t nnz = indptr[n] - indptr[0];
    if (nnz < 0 || (nnz > 0 && (!indices || !values))) return fail(h, HR_EINVAL, "bad indices/values");
    float batch_max = 0.f;
    HR_TRY(validate_csr(h, indptr, indices, values, n, &batch_max));
    std::unique_lock<std::shared_mutex> lk(h->rw);
    if (h->n_sparse + n > (1ll << 31) - 64) return fail(h, HR_ELIMIT, "too many sparse rows");
    try {  // staged on the host only until the next hr_finalize uploads them
        const int64_t base = h->pend_indptr.back() - indptr[0];
        h->pend_indptr.reserve(h->pend_indptr.size() + n);
        for (int64_t r = 1; r <= n; ++r) h->pend_indptr.push_back(indptr[r] + base);
        h->pend_idx.insert(h->pend_idx.end(), indices + indptr[0], indices + indptr[n]);
        h->pend_val.insert(h->pend_val.end(), values + indptr[0], values + indptr[n]);
    } catch (const std::exception&) {
        return fail(h, HR_ENOMEM, "out of host memory staging sparse rows");
    }
    h->n_sparse += n;
    h->max_sparse_abs = std::max(h->max_sparse_abs, batch_max);
    h->finalized = false;
    return HR_OK;
}

int hr_finalize(hr_index* h) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    hipStream_t s = h->ingest_stream;
    // `*_dev` searches drop their shared lock once their kernels are enqueued: a scan still running on a caller's stream
    // must not see the run tables and posting blocks of the dirty ranges while they are rebuilt in place
    if ((h->dim > 0 && h->n_normed < h->n_rows) || (h->sparse_dim > 0 && h->n_sparse_built != h->n_sparse))
        HIP_TRY(h, hipDeviceSynchronize());
    if (h->dim > 0 && h->n_normed < h->n_rows) {
        const int64_t n = h->n_rows - h->n_normed;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        const int cosine = h->metric == HR_METRIC_COSINE;
        if (h->dtype == HR_F16)
            hipLaunchKernelGGL((row_norms_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, h->tiles.as<chunk_t>(),
                               h->KT, h->n_normed, n, cosine, h->norm2.as<double>(), h->scale.as<float>(),
                               h->max_norm.as<unsigned int>());
        else
            hipLaunchKernelGGL((row_norms_kernel<float>), dim3(blocks), dim3(256), 0, s, h->tiles.as<chunk_t>(), h->KT,
                               h->n_normed, n, cosine, h->norm2.as<double>(), h->scale.as<float>(),
                               h->max_norm.as<unsigned int>());
        HIP_TRY(h, hipGetLastError());
        unsigned int bits = 0;
        HIP_TRY(h, hipMemcpyAsync(&bits, h->max_norm.p, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        std::memcpy(&h->max_row_norm, &bits, 4);
        h->n_normed = h->n_rows;
    }
    if (h->sparse_dim > 0 && (h->n_sparse_built != h->n_sparse || h->n_csr != h->n_sparse)) HR_TRY(build_sparse(h));
    h->finalized = true;
    return HR_OK;
}

namespace {
struct SnapHeader {
    char magic[8];
    int32_t version, dtype, metric, KT;
    int64_t dim, sparse_dim, n_rows, cap_rows, n_sparse, nnz, row_offset;
    float max_row_norm, max_sparse_abs;
    int64_t file_bytes;  // header + every section: a truncated file is refused before anything is uploaded
};
const char kSnapMagic[8] = {'H', 'B', 'M', 'R', 'A', 'G', '0', '2'};

struct File {
    FILE* f = nullptr;
    ~File() { if (f) fclose(f); }
};

// device <-> file in 64 MiB pieces through a host bounce buffer
int stream_out(hr_index* h, FILE* f, const void* dptr, size_t bytes) {
    std::vector<char> buf(std::min<size_t>(bytes, 64u << 20));
    for (size_t off = 0; off < bytes; off += buf.size()) {
        const size_t n = std::min(buf.size(), bytes - off);
        HIP_TRY(h, hipMemcpy(buf.data(), (const char*)dptr + off, n, hipMemcpyDeviceToHost));
        if (fwrite(buf.data(), 1, n, f) != n) return fail(h, HR_EINVAL, "snapshot write failed");
    }
    return HR_OK;
}
int stream_in(hr_index* h, FILE* f, void* dptr, size_t bytes) {
    std::vector<char> buf(std::min<size_t>(bytes, 64u << 20));
    for (size_t off = 0; off < bytes; off += buf.size()) {
        const size_t n = std::min(buf.size(), bytes - off);
        if (fread(buf.data(), 1, n, f) != n) return fail(h, HR_EINVAL, "snapshot truncated");
        HIP_TRY(h, hipMemcpy((char*)dptr + off, buf.data(), n, hipMemcpyHostToDevice));
    }
    return HR_OK;
}

int save_to(hr_index* h, FILE* f) {
    SnapHeader hd{};
    std::memcpy(hd.magic, kSnapMagic, 8);
    hd.version = 2; hd.dtype = h->dtype; hd.metric = h->metric; hd.KT = h->KT;
    hd.dim = h->dim; hd.sparse_dim = h->sparse_dim; hd.n_rows = h->n_rows;
    hd.cap_rows = h->dim ? round_up(std::max<int64_t>(h->n_rows, 1), kSuperRows) : 0;
    hd.n_sparse = h->n_sparse; hd.nnz = h->nnz_csr; hd.row_offset = h->row_offset;
    hd.max_row_norm = h->max_row_norm; hd.max_sparse_abs = h->max_sparse_abs;
    const bool dense = h->dim > 0 && h->n_rows > 0;
    hd.file_bytes = (int64_t)sizeof hd + (dense ? (int64_t)tile_bytes_for_rows(h, hd.cap_rows) + hd.cap_rows * 12 : 0) +
                    (h->n_sparse > 0 ? (h->n_sparse + 1) * 8 + hd.nnz * 8 : 0);
    if (fwrite(&hd, sizeof hd, 1, f) != 1) return fail(h, HR_EINVAL, "snapshot write failed");
    if (dense) {
        HR_TRY(stream_out(h, f, h->tiles.p, tile_bytes_for_rows(h, hd.cap_rows)));
        HR_TRY(stream_out(h, f, h->scale.p, (size_t)hd.cap_rows * 4));
        HR_TRY(stream_out(h, f, h->norm2.p, (size_t)hd.cap_rows * 8));
    }
    if (h->n_sparse > 0) {  // the CSR is read back from the device: the host keeps no copy of it
        HR_TRY(stream_out(h, f, h->s_indptr.p, (size_t)(h->n_sparse + 1) * 8));
        HR_TRY(stream_out(h, f, h->s_idx.p, (size_t)hd.nnz * 4));
        HR_TRY(stream_out(h, f, h->s_val.p, (size_t)hd.nnz * 4));
    }
    return HR_OK;
}
}  // namespace

int hr_save(hr_index* h, const char* path) {
    if (!h || !path) return fail(h, HR_EINVAL, "null argument");
    if (!h->finalized) return fail(h, HR_ESTATE, "hr_save before hr_finalize");
    std::unique_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    // written beside the target and renamed once complete: a full disk or a crash leaves the old snapshot intact
    std::string tmp_path;
    try {
        tmp_path = std::string(path) + ".tmp";
    } catch (const std::exception&) {
        return fail(h, HR_ENOMEM, "out of host memory");
    }
    int rc;
    {
        File file;
        file.f = fopen(tmp_path.c_str(), "wb");
        if (!file.f) return fail(h, HR_EINVAL, "cannot open %s for writing", tmp_path.c_str());
        rc = save_to(h, file.f);
        if (rc == HR_OK && fflush(file.f) != 0) rc = fail(h, HR_EINVAL, "snapshot write failed (flush)");
        FILE* f = file.f;
        file.f = nullptr;
        if (fclose(f) != 0 && rc == HR_OK) rc = fail(h, HR_EINVAL, "snapshot write failed (close)");
    }
    if (rc == HR_OK && std::rename(tmp_path.c_str(), path) != 0) rc = fail(h, HR_EINVAL, "cannot move snapshot into place at %s", path);
    if (rc != HR_OK) (void)std::remove(tmp_path.c_str());
    return rc;
}

static int load_impl(const char* path, int device, hr_index** out) {
    File file;
    file.f = fopen(path, "rb");
    if (!file.f) return fail(nullptr, HR_EINVAL, "cannot open %s", path);
    SnapHeader hd{};
    if (fread(&hd, sizeof hd, 1, file.f) != 1 || std::memcmp(hd.magic, kSnapMagic, 8) != 0 || hd.version != 2)
        return fail(nullptr, HR_EINVAL, "%s is not a libhbmrag snapshot (version 2)", path);
    if (hd.n_rows < 0 || hd.n_sparse < 0 || hd.nnz < 0 || hd.cap_rows < hd.n_rows || hd.n_sparse > (1ll << 31) - 64 ||
        hd.nnz > (1ll << 40) || !(hd.max_sparse_abs >= 0.f) || !(hd.max_row_norm >= 0.f))
        return fail(nullptr, HR_EINVAL, "corrupt snapshot header");
    if (fseek(file.f, 0, SEEK_END) != 0 || (int64_t)ftell(file.f) != hd.file_bytes || fseek(file.f, (long)sizeof hd, SEEK_SET) != 0)
        return fail(nullptr, HR_EINVAL, "snapshot truncated or padded: %s does not hold the %lld bytes its header announces", path,
                    (long long)hd.file_bytes);
    hr_index* h = nullptr;
    HR_TRY(hr_create(device, hd.dim, hd.dtype, hd.metric, hd.sparse_dim, &h));
    struct Guard { hr_index* h; bool keep = false; ~Guard() { if (!keep) hr_destroy(h); } } guard{h};
    if (h->KT != hd.KT) return fail(nullptr, HR_EINVAL, "snapshot tile layout (KT=%d) differs from this build (KT=%d)", hd.KT, h->KT);
    h->row_offset = hd.row_offset;
    DeviceGuard dg(device);
    if (hd.dim > 0 && hd.n_rows > 0) {
        if (hd.cap_rows != round_up(hd.n_rows, kSuperRows)) return fail(nullptr, HR_EINVAL, "corrupt snapshot header (row capacity)");
        HR_TRY(hr_reserve(h, hd.cap_rows));
        int rc = stream_in(h, file.f, h->tiles.p, tile_bytes_for_rows(h, hd.cap_rows));
        if (rc == HR_OK) rc = stream_in(h, file.f, h->scale.p, (size_t)hd.cap_rows * 4);
        if (rc == HR_OK) rc = stream_in(h, file.f, h->norm2.p, (size_t)hd.cap_rows * 8);
        if (rc != HR_OK) return fail(nullptr, rc, "%s", hr_last_error(h));
        h->n_rows = h->n_normed = hd.n_rows;
        h->max_row_norm = hd.max_row_norm;
        unsigned int bits;
        std::memcpy(&bits, &hd.max_row_norm, 4);
        if (hipMemcpy(h->max_norm.p, &bits, 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail(nullptr, HR_EHIP, "snapshot upload failed");
    }
    if (hd.n_sparse > 0) {
        if (hd.sparse_dim <= 0) return fail(nullptr, HR_EINVAL, "corrupt snapshot header (sparse rows without a sparse collection)");
        // the file is not trusted: the CSR goes through the same checks as hr_add_sparse before any kernel indexes with it
        std::vector<int64_t> ptr((size_t)hd.n_sparse + 1);
        std::vector<int32_t> idx((size_t)hd.nnz);
        std::vector<float> val((size_t)hd.nnz);
        const bool ok = fread(ptr.data(), 8, ptr.size(), file.f) == ptr.size() &&
                        fread(idx.data(), 4, idx.size(), file.f) == idx.size() &&
                        fread(val.data(), 4, val.size(), file.f) == val.size();
        if (!ok || ptr.front() != 0 || ptr.back() != hd.nnz)
            return fail(nullptr, HR_EINVAL, "snapshot truncated or corrupt (sparse section)");
        for (int64_t r = 0; r < hd.n_sparse; ++r)
            if (ptr[r + 1] < ptr[r] || ptr[r + 1] > hd.nnz) return fail(nullptr, HR_EINVAL, "snapshot corrupt (sparse row pointers)");
        int rc = hr_add_sparse(h, ptr.data(), idx.data(), val.data(), hd.n_sparse);  // validates; recomputes max |weight|
        if (rc != HR_OK) return fail(nullptr, rc, "snapshot corrupt (sparse section): %s", hr_last_error(h));
    }
    int rc = hr_finalize(h);
    if (rc != HR_OK) return fail(nullptr, rc, "%s", hr_last_error(h));
    guard.keep = true;
    *out = h;
    return HR_OK;
}

int hr_load(const char* path, int device, hr_index** out) {
    if (!path || !out) return fail(nullptr, HR_EINVAL, "null argument");
    *out = nullptr;
    try {
        return load_impl(path, device, out);
    } catch (const std::bad_alloc&) {
        return fail(nullptr, HR_ENOMEM, "out of host memory loading snapshot");
    } catch (const std::exception& e) {  // e.g. std::length_error from an absurd size in a damaged header
        return fail(nullptr, HR_EINVAL, "snapshot refused: %s", e.what());
    }
}

int hr_get_info(const hr_index* h, int64_t* dim, int32_t* dtype, int32_t* metric, int64_t* sparse_dim) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (dim) *dim = h->dim;
    if (dtype) *dtype = h->dtype;
    if (metric) *metric = h->metric;
    if (sparse_dim) *sparse_dim = h->sparse_dim;
    return HR_OK;
}

int64_t hr_num_rows(const hr_index* h) { return h ? h->n_rows : 0; }
int64_t hr_num_sparse_rows(const hr_index* h) { return h ? h->n_sparse : 0; }
int64_t hr_device_bytes(const hr_index* h) {
    if (!h) return 0;
    size_t t = 0;
    for (const DevBuf* b : {&h->tiles, &h->scale, &h->norm2, &h->s_indptr, &h->s_idx, &h->s_val, &h->rt_off,
                            &h->range_base, &h->post})
        t += b->cap;
    return (int64_t)t;
}
int64_t hr_dense_scan_bytes(const hr_index* h) {
    if (!h || h->dim == 0) return 0;
    const int64_t dpad = (int64_t)h->KT * 4 * elems_per_chunk(h->dtype);
    return h->n_rows * dpad * (int64_t)elem_size(h->dtype) + h->n_rows * 4;
}

// ---- device-pointer, asynchronous forms -----------------------------------------
int hr_search_dense_dev(hr_index* h, const float* d_q, int B, int k, const uint8_t* d_rowmask, int64_t* d_ids,
                        float* d_scores, int32_t* d_flags, void* stream) {
    HR_TRY(check_search_args(h, B, k, true));
    if (!d_q || !d_ids || !d_scores) return fail(h, HR_EINVAL, "null buffer");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (h->n_rows == 0) return fill_empty(h, s, B, k, d_ids, d_scores, d_flags);
    StreamWs ws_held;
    Workspace* ws = ws_for_stream(h, stream, ws_held);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    return dense_search_enqueue(h, ws, s, d_q, B, k, d_rowmask, d_ids, d_scores, d_flags, candidate_groups_for_k(k));
}

int hr_search_sparse_dev(hr_index* h, const int64_t* d_q_indptr, const int32_t* d_q_idx, const float* d_q_val, int B,
                         int64_t q_nnz_total, int max_q_nnz, int k, const uint8_t* d_rowmask, int64_t* d_ids,
                         float* d_scores, int32_t* d_flags, void* stream) {
    HR_TRY(check_search_args(h, B, k, false));
    if (!d_q_indptr || !d_ids || !d_scores || (q_nnz_total > 0 && (!d_q_idx || !d_q_val)))
        return fail(h, HR_EINVAL, "null buffer");
    if (max_q_nnz < 0 || max_q_nnz > HR_MAX_QUERY_NNZ) return fail(h, HR_ELIMIT, "query nnz exceeds HR_MAX_QUERY_NNZ");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (h->n_sparse == 0) return fill_empty(h, s, B, k, d_ids, d_scores, d_flags);
    StreamWs ws_held;
    Workspace* ws = ws_for_stream(h, stream, ws_held);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    return sparse_search_enqueue(h, ws, s, d_q_indptr, d_q_idx, d_q_val, B, max_q_nnz, k, d_rowmask, d_ids, d_scores,
                                 d_flags, candidate_groups_for_k(k));
}

int hr_search_hybrid_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                         const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                         const uint8_t* d_rowmask, int64_t* d_ids, float* d_scores, int32_t* d_flags, void* stream) {
    HR_TRY(check_search_args(h, B, k, true));
    HR_TRY(check_search_args(h, B, k, false));
    if (!d_q || !d_q_indptr || !d_ids || !d_scores || (q_nnz_total > 0 && (!d_q_idx || !d_q_val)))
        return fail(h, HR_EINVAL, "null buffer");
    if (max_q_nnz < 0 || max_q_nnz > HR_MAX_QUERY_NNZ) return fail(h, HR_ELIMIT, "query nnz exceeds HR_MAX_QUERY_NNZ");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    hipStream_t s = (hipStream_t)stream;
    int64_t* s_ids = d_ids + (size_t)B * k;
    float* s_scores = d_scores + (size_t)B * k;
    int32_t* s_flags = d_flags ? d_flags + B : nullptr;
    StreamWs ws_held;
    Workspace* ws = ws_for_stream(h, stream, ws_held);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    if (!ws->side) {
        HIP_TRY(h, hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&ws->ev_scan, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&ws->ev_side, hipEventDisableTiming));
        ws->sparse_ws = new (std::nothrow) Workspace();
        if (!ws->sparse_ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    }
    const int C = candidate_groups_for_k(k);
    if (h->n_rows == 0) {
        HR_TRY(fill_empty(h, s, B, k, d_ids, d_scores, d_flags));
        HIP_TRY(h, hipEventRecord(ws->ev_scan, s));
    } else {
        HR_TRY(dense_search_enqueue(h, ws, s, d_q, B, k, d_rowmask, d_ids, d_scores, d_flags, C, ws->ev_scan));
    }
    // sparse chain: starts when the dense scan is done, overlaps the dense tail
    HIP_TRY(h, hipStreamWaitEvent(ws->side, ws->ev_scan, 0));
    if (h->n_sparse == 0) {
        HR_TRY(fill_empty(h, ws->side, B, k, s_ids, s_scores, s_flags));
    } else {
        HR_TRY(sparse_search_enqueue(h, ws->sparse_ws, ws->side, d_q_indptr, d_q_idx, d_q_val, B, max_q_nnz, k,
                                     d_rowmask, s_ids, s_scores, s_flags, C));
    }
    HIP_TRY(h, hipEventRecord(ws->ev_side, ws->side));
    HIP_TRY(h, hipStreamWaitEvent(s, ws->ev_side, 0));
    return HR_OK;
}

static int slot_workspaces(hr_index* h, int slot, Workspace** dense, Workspace** sparse) {
    if (slot < 0 || slot >= HR_MAX_SLOTS) return fail(h, HR_EINVAL, "slot %d out of range [0,%d)", slot, HR_MAX_SLOTS);
    std::lock_guard<std::mutex> g(h->pool_mu);
    for (int m = 0; m < 2; ++m)
        if (!h->slot_ws[slot][m]) {
            h->slot_ws[slot][m] = new (std::nothrow) Workspace();
            if (!h->slot_ws[slot][m]) return fail(h, HR_ENOMEM, "workspace allocation failed");
        }
    *dense = h->slot_ws[slot][0];
    *sparse = h->slot_ws[slot][1];
    return HR_OK;
}

// phases = PHASE_PREP (hr_hybrid_prep_dev) or the scans (+ the prep when the slot was not prepared beforehand)
static int hybrid_scan_phases(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                              const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                              const uint8_t* d_rowmask, int slot, void* stream, bool prep_only) {
    HR_TRY(check_search_args(h, B, k, true));
    HR_TRY(check_search_args(h, B, k, false));
    if (!d_q || !d_q_indptr || (q_nnz_total > 0 && (!d_q_idx || !d_q_val))) return fail(h, HR_EINVAL, "null buffer");
    if (max_q_nnz < 0 || max_q_nnz > HR_MAX_QUERY_NNZ) return fail(h, HR_ELIMIT, "query nnz exceeds HR_MAX_QUERY_NNZ");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    Workspace *wd = nullptr, *wsp = nullptr;
    HR_TRY(slot_workspaces(h, slot, &wd, &wsp));
    if (!wd || !wsp) return fail(h, HR_ENOMEM, "no workspace for slot %d", slot);
    std::lock_guard<std::mutex> slot_held(wd->mu);
    hipStream_t s = (hipStream_t)stream;
    const int C = candidate_groups_for_k(k);
    int phases = PHASE_PREP;
    if (!prep_only) {
        phases = h->slot_prepped[slot] ? PHASE_SCAN : (PHASE_PREP | PHASE_SCAN);
        h->slot_prepped[slot] = false;
    }
    if (h->n_rows > 0)
        HR_TRY(dense_search_enqueue(h, wd, s, d_q, B, k, d_rowmask, nullptr, nullptr, nullptr, C, nullptr, phases));
    if (h->n_sparse > 0)
        HR_TRY(sparse_search_enqueue(h, wsp, s, d_q_indptr, d_q_idx, d_q_val, B, max_q_nnz, k, d_rowmask, nullptr,
                                     nullptr, nullptr, C, phases));
    if (prep_only) h->slot_prepped[slot] = true;
    return HR_OK;
}

int hr_hybrid_prep_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                       const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k, int slot, void* stream) {
    return hybrid_scan_phases(h, d_q, d_q_indptr, d_q_idx, d_q_val, B, q_nnz_total, max_q_nnz, k, nullptr, slot, stream, true);
}

int hr_hybrid_scan_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                       const float* d_q_val, int B, int64_t q_nnz_total, int max_q_nnz, int k,
                       const uint8_t* d_rowmask, int slot, void* stream) {
    return hybrid_scan_phases(h, d_q, d_q_indptr, d_q_idx, d_q_val, B, q_nnz_total, max_q_nnz, k, d_rowmask, slot, stream, false);
}

int hr_hybrid_finish_dev(hr_index* h, const float* d_q, const int64_t* d_q_indptr, const int32_t* d_q_idx,
                         const float* d_q_val, int B, int max_q_nnz, int k, const uint8_t* d_rowmask, int slot,
                         int64_t* d_ids, float* d_scores, int32_t* d_flags, void* stream) {
    HR_TRY(check_search_args(h, B, k, true));
    HR_TRY(check_search_args(h, B, k, false));
    if (!d_q || !d_q_indptr || !d_ids || !d_scores) return fail(h, HR_EINVAL, "null buffer");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    Workspace *wd = nullptr, *wsp = nullptr;
    HR_TRY(slot_workspaces(h, slot, &wd, &wsp));
    if (!wd || !wsp) return fail(h, HR_ENOMEM, "no workspace for slot %d", slot);
    std::lock_guard<std::mutex> slot_held(wd->mu);
    hipStream_t s = (hipStream_t)stream;
    const int C = candidate_groups_for_k(k);
    int64_t* s_ids = d_ids + (size_t)B * k;
    float* s_scores = d_scores + (size_t)B * k;
    int32_t* s_flags = d_flags ? d_flags + B : nullptr;
    if (h->n_rows > 0 && h->n_sparse > 0)
        return hybrid_finish_enqueue(h, wd, wsp, s, d_q, d_q_indptr, d_q_idx, d_q_val, B, max_q_nnz, k, d_rowmask, d_ids,
                                     d_scores, d_flags, s_ids, s_scores, s_flags, C);
    if (h->n_rows > 0)
        HR_TRY(dense_search_enqueue(h, wd, s, d_q, B, k, d_rowmask, d_ids, d_scores, d_flags, C, nullptr, PHASE_FINISH));
    else
        HR_TRY(fill_empty(h, s, B, k, d_ids, d_scores, d_flags));
    if (h->n_sparse > 0)
        HR_TRY(sparse_search_enqueue(h, wsp, s, d_q_indptr, d_q_idx, d_q_val, B, max_q_nnz, k, d_rowmask, s_ids,
                                     s_scores, s_flags, C, PHASE_FINISH));
    else
        HR_TRY(fill_empty(h, s, B, k, s_ids, s_scores, s_flags));
    return HR_OK;
}

int hr_fuse_rrf_dev(const int64_t* d_ids_a, int ka, const int64_t* d_ids_b, int kb, const int64_t* d_ids_c, int kc,
                    int B, double wa, double wb, double wc, int rrf_k, int top_k, int64_t* d_out_ids,
                    double* d_out_scores, int32_t* d_out_methods, int32_t* d_n_out, void* stream) {
    if (B <= 0 || ka < 0 || kb < 0 || kc < 0 || top_k <= 0) return fail(nullptr, HR_EINVAL, "bad fuse sizes");
    if (ka > HR_MAX_TOPK || kb > HR_MAX_TOPK || kc > HR_MAX_TOPK)
        return fail(nullptr, HR_ELIMIT, "list longer than HR_MAX_TOPK=%d", HR_MAX_TOPK);
    if ((ka && !d_ids_a) || (kb && !d_ids_b) || (kc && !d_ids_c) || !d_out_ids || !d_out_scores || !d_out_methods || !d_n_out)
        return fail(nullptr, HR_EINVAL, "null buffer");
    hipLaunchKernelGGL(rrf_fuse_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, d_ids_a, ka, d_ids_b, kb, d_ids_c, kc,
                       wa, wb, wc, rrf_k, top_k, d_out_ids, d_out_scores, d_out_methods, d_n_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "rrf_fuse_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_merge_topk_dev(const float* d_scores, const int64_t* d_ids, int n_lists, int64_t score_stride, int64_t id_stride,
                      int B, int k_in, int k_out, int64_t* d_out_ids, float* d_out_scores, void* stream) {
    if (n_lists <= 0 || B <= 0 || k_in <= 0 || k_out <= 0) return fail(nullptr, HR_EINVAL, "bad merge sizes");
    if (score_stride < (int64_t)B * k_in || id_stride < (int64_t)B * k_in)
        return fail(nullptr, HR_EINVAL, "list stride smaller than one list");
    if (!d_scores || !d_ids || !d_out_ids || !d_out_scores) return fail(nullptr, HR_EINVAL, "null buffer");
    const size_t lds = merge_lds_bytes(n_lists, k_in);
    if (lds <= kMergeLdsMax)
        hipLaunchKernelGGL(merge_topk_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, d_scores, d_ids, n_lists,
                           score_stride, id_stride, k_in, k_out, d_out_ids, d_out_scores);
    else
        hipLaunchKernelGGL(merge_topk_big_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, d_scores, d_ids, n_lists,
                           score_stride, id_stride, k_in, k_out, d_out_ids, d_out_scores);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "merge_topk_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_post_lists_dev(const hr_post_args* a, int B, void* stream) {
    if (!a || B <= 0) return fail(nullptr, HR_EINVAL, "bad post-lists arguments");
    if (a->n_lists < 1 || a->top_k <= 0 || a->k_in[0] <= 0) return fail(nullptr, HR_EINVAL, "bad post-lists sizes");
    size_t lds = 0;
    for (int m = 0; m < 3; ++m) {
        if (a->k_in[m] < 0 || a->k_in[m] > HR_MAX_TOPK || a->k_fuse[m] > HR_MAX_TOPK)
            return fail(nullptr, HR_ELIMIT, "list longer than HR_MAX_TOPK=%d", HR_MAX_TOPK);
        if (!a->k_in[m]) continue;
        if (!a->ids[m] || a->k_fuse[m] <= 0) return fail(nullptr, HR_EINVAL, "null list / bad k_fuse of modality %d", m);
        if (a->n_lists == 1) {
            if (a->k_fuse[m] != a->k_in[m]) return fail(nullptr, HR_EINVAL, "k_fuse must equal k_in without a merge");
        } else {
            if (!a->scores[m] || !a->merged_ids[m] || !a->merged_scores[m])
                return fail(nullptr, HR_EINVAL, "null merge buffer of modality %d", m);
            if (a->score_stride < (int64_t)B * a->k_in[m] || a->id_stride < (int64_t)B * a->k_in[m])
                return fail(nullptr, HR_EINVAL, "list stride smaller than one list");
            lds = std::max(lds, merge_lds_bytes(a->n_lists, a->k_in[m]));
        }
    }
    if (lds > kMergeLdsMax) return fail(nullptr, HR_ELIMIT, "merge of %d lists does not fit LDS; use hr_merge_topk_dev", a->n_lists);
    if (!a->fused_ids || !a->fused_scores || !a->fused_methods || !a->fused_n) return fail(nullptr, HR_EINVAL, "null fusion buffer");
    if (a->rerank && (a->k_out <= 0 || a->top_k > HR_MAX_TOPK || !a->rr_ids || !a->rr_scores || !a->rr_orig))
        return fail(nullptr, HR_EINVAL, "bad rerank buffers");
    if (a->agg_flags && a->n_lists > 1 && (!a->flags || a->n_flag_rows <= 0 || a->flag_stride < a->n_flag_rows))
        return fail(nullptr, HR_EINVAL, "bad flag buffers");
    PostArgs pa;
    pa.a = *a;
    hipLaunchKernelGGL(post_lists_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, pa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "post_lists_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_filter_eval_dev(const hr_filter_term* terms, int n_terms, int64_t n_rows, const uint8_t* d_deleted, uint8_t* d_mask,
                       uint8_t* d_undecided, int32_t* d_counts, void* stream) {
    if (n_terms < 0 || n_terms > kFilterMaxTerms) return fail(nullptr, HR_ELIMIT, "a filter expression may hold up to %d terms", kFilterMaxTerms);
    if (n_rows < 0 || (n_terms > 0 && !terms) || !d_mask || !d_undecided || !d_counts) return fail(nullptr, HR_EINVAL, "bad filter arguments");
    if (((uintptr_t)d_mask | (uintptr_t)d_undecided) & 7) return fail(nullptr, HR_EINVAL, "mask buffers must be 8-byte aligned");
    FilterArgs a{};
    for (int i = 0; i < n_terms; ++i) {
        if (terms[i].kind < HR_COL_I64 || terms[i].kind > HR_COL_STR16 || terms[i].op < HR_OP_EQ || terms[i].op > HR_OP_GE || !terms[i].col)
            return fail(nullptr, HR_EINVAL, "bad filter term %d", i);
        a.t[i] = terms[i];
    }
    a.n_terms = n_terms;
    a.n_rows = n_rows;
    a.deleted = d_deleted;
    a.mask = (unsigned long long*)d_mask;
    a.undecided = (unsigned long long*)d_undecided;
    a.counts = d_counts;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_counts, 0, 8, s);
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    if (n_rows == 0) return HR_OK;
    const int64_t n_words = (n_rows + 63) / 64;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_words + 3) / 4, 4096));
    hipLaunchKernelGGL(filter_eval_kernel, dim3(blocks), dim3(256), 0, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "filter_eval_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_bm25_encode_dev(const uint8_t* d_text, const int64_t* d_off, int n_docs, int sparse_dim, double k1, double b,
                       double avgdl, int cap, int32_t* d_idx, float* d_val, int32_t* d_nnz, int32_t* d_flags, void* stream) {
    if (n_docs < 0 || !d_off || !d_nnz || !d_flags || (n_docs > 0 && (!d_text || !d_idx || !d_val)))
        return fail(nullptr, HR_EINVAL, "bad bm25 arguments");
    if (sparse_dim < 1 || sparse_dim > kBm25MaxDim) return fail(nullptr, HR_ELIMIT, "sparse_dim must be 1..%d for the device encoder", kBm25MaxDim);
    if (cap < 1 || !(avgdl > 0.0) || !(k1 >= 0.0) || !(b >= 0.0 && b <= 1.0)) return fail(nullptr, HR_EINVAL, "bad bm25 parameters");
    if (n_docs == 0) return HR_OK;
    Bm25Args a{};
    a.text = d_text; a.off = d_off; a.n_docs = n_docs; a.sparse_dim = sparse_dim; a.cap = cap;
    a.k1 = k1; a.b = b; a.avgdl = avgdl;
    a.idx = d_idx; a.val = d_val; a.nnz = d_nnz; a.flags = d_flags;
    const size_t lds = (size_t)((sparse_dim + 1) / 2) * 4;
    static bool attr_set = false;  // benign race: the attribute is idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)bm25_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (kBm25MaxDim / 2) * 4);
        if (e != hipSuccess) return fail(nullptr, HR_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(bm25_encode_kernel, dim3((unsigned)n_docs), dim3(kBm25Threads), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "bm25_encode_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_hash_tokenize_dev(const uint8_t* d_text, const int64_t* d_off, int n, int max_len, int vocab, int64_t* d_ids,
                         int32_t* d_lens, int32_t* d_flags, void* stream) {
    if (n < 0 || !d_off || !d_lens || !d_flags || (n > 0 && (!d_text || !d_ids))) return fail(nullptr, HR_EINVAL, "bad tokenizer arguments");
    if (max_len < 3 || vocab <= 1000) return fail(nullptr, HR_EINVAL, "max_len must be >= 3 and the vocabulary larger than 1000");
    if (n == 0) return HR_OK;
    TokArgs a{};
    a.text = d_text; a.off = d_off; a.n = n; a.max_len = max_len; a.vocab = vocab;
    a.ids = d_ids; a.lens = d_lens; a.flags = d_flags;
    hipLaunchKernelGGL(hash_tokenize_kernel, dim3((unsigned)n), dim3(kTokThreads), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "hash_tokenize_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_stream_create(int device, int priority, const uint32_t* cu_mask, int n_words, void** out_stream) {
    if (!out_stream || n_words < 0 || (n_words > 0 && !cu_mask)) return fail(nullptr, HR_EINVAL, "bad stream arguments");
    *out_stream = nullptr;
    DeviceGuard dg(device);
    hipStream_t s = nullptr;
    hipError_t e;
    if (n_words > 0) {
        // (a masked stream takes the default priority: HIP has no entry point that sets both)
        e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask);
    } else {
        int lo = 0, hi = 0;  // lo = numerically greatest = lowest priority
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, std::max(hi, std::min(lo, priority)));
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, HR_EHIP, "stream creation failed: %s", hipGetErrorString(e));
    }
    *out_stream = (void*)s;
    return HR_OK;
}

int hr_stream_destroy(int device, void* stream) {
    if (!stream) return HR_OK;
    DeviceGuard dg(device);
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "hipStreamDestroy: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_set_scan_cus(hr_index* h, int n_cus) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (n_cus < 0) return fail(h, HR_EINVAL, "n_cus must be >= 0");
    h->scan_cus = n_cus;
    return HR_OK;
}

int hr_debug_option(hr_index* h, int key, int value) {
    switch (key) {
        case HR_DEBUG_FINISH_MODE:
            if (value < 0 || value > 2) return fail(h, HR_EINVAL, "finish mode must be 0, 1 or 2");
            g_finish_mode = value;
            return HR_OK;
        case HR_DEBUG_DENSE_KERNELS:
            if (value < 0 || value > 31) return fail(h, HR_EINVAL, "dense kernel mask must be 0..31");
            g_dense_kernels = value;
            return HR_OK;
        case HR_DEBUG_SPARSE_RPB:
            if (value < 0 || value > 64) return fail(h, HR_EINVAL, "ranges per block must be 0..64");
            g_sparse_rpb = value;
            return HR_OK;
        case HR_DEBUG_GROUP_ROWS:
            if (value != 0 && value != 16 && value != 64) return fail(h, HR_EINVAL, "group rows must be 0, 16 or 64");
            g_group_rows = value;
            return HR_OK;
        case HR_DEBUG_NO_TRIM:
            g_no_trim = value != 0;
            return HR_OK;
        case HR_DEBUG_FAIL_NEXT_BUILD:
            if (!h) return fail(nullptr, HR_EINVAL, "null handle");
            h->fault_inject = value ? 1 : 0;
            return HR_OK;
        default:
            return fail(h, HR_EINVAL, "unknown debug option %d", key);
    }
}

int hr_rerank_linear_dev(const int64_t* d_ids, const double* d_scores, const int32_t* d_methods, const int32_t* d_n,
                         const double* d_recency, int B, int k_in, double base_w, double method_bonus,
                         double recency_w, int k_out, int64_t* d_out_ids, double* d_out_scores, double* d_out_orig,
                         void* stream) {
    if (B <= 0 || k_in <= 0 || k_out <= 0) return fail(nullptr, HR_EINVAL, "bad rerank sizes");
    if (k_in > HR_MAX_TOPK) return fail(nullptr, HR_ELIMIT, "k_in exceeds HR_MAX_TOPK=%d", HR_MAX_TOPK);
    if (!d_ids || !d_scores || !d_methods || !d_n || !d_out_ids || !d_out_scores || !d_out_orig)
        return fail(nullptr, HR_EINVAL, "null buffer");
    hipLaunchKernelGGL(rerank_linear_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, d_ids, d_scores, d_methods, d_n,
                       d_recency, k_in, base_w, method_bonus, recency_w, k_out, d_out_ids, d_out_scores, d_out_orig);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "rerank_linear_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_add_layernorm_f16_dev(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, void* d_out,
                             int64_t rows, int hidden, float eps, void* stream) {
    if (rows < 0 || hidden <= 0 || hidden % 8 != 0) return fail(nullptr, HR_EINVAL, "hidden must be a positive multiple of 8");
    if (hidden > 1024) return fail(nullptr, HR_ELIMIT, "hidden exceeds 1024");
    if (!d_x || !d_gamma || !d_beta || !d_out) return fail(nullptr, HR_EINVAL, "null buffer");
    if (((uintptr_t)d_x | (uintptr_t)d_residual | (uintptr_t)d_gamma | (uintptr_t)d_beta | (uintptr_t)d_out) & 15)
        return fail(nullptr, HR_EINVAL, "buffers must be 16-byte aligned");
    if (rows == 0) return HR_OK;
    const int chunks = hidden / 8;
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    auto* x = (const half8_t*)d_x;
    auto* r = (const half8_t*)d_residual;
    auto* g = (const half8_t*)d_gamma;
    auto* b = (const half8_t*)d_beta;
    auto* o = (half8_t*)d_out;
    hipStream_t s = (hipStream_t)stream;
    if (chunks <= 16) hipLaunchKernelGGL((add_layernorm_f16_kernel<1>), grid, block, 0, s, x, r, g, b, o, rows, chunks, eps);
    else if (chunks <= 32) hipLaunchKernelGGL((add_layernorm_f16_kernel<2>), grid, block, 0, s, x, r, g, b, o, rows, chunks, eps);
    else if (chunks <= 48) hipLaunchKernelGGL((add_layernorm_f16_kernel<3>), grid, block, 0, s, x, r, g, b, o, rows, chunks, eps);
    else if (chunks <= 64) hipLaunchKernelGGL((add_layernorm_f16_kernel<4>), grid, block, 0, s, x, r, g, b, o, rows, chunks, eps);
    else hipLaunchKernelGGL((add_layernorm_f16_kernel<8>), grid, block, 0, s, x, r, g, b, o, rows, chunks, eps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "add_layernorm_f16_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_embed_layernorm_f16_dev(const int64_t* d_ids, const int64_t* d_types, const void* d_word, const void* d_pos, const void* d_seg,
                               const void* d_gamma, const void* d_beta, void* d_out, int64_t n_seq, int T, int hidden, float eps,
                               int64_t n_word, int64_t n_seg, void* stream) {
    if (n_word <= 0 || n_seg <= 0) return fail(nullptr, HR_EINVAL, "empty embedding table");
    if (n_seq < 0 || T <= 0 || hidden <= 0 || hidden % 8 != 0) return fail(nullptr, HR_EINVAL, "hidden must be a positive multiple of 8");
    if (hidden > 1024) return fail(nullptr, HR_ELIMIT, "hidden exceeds 1024");
    if (!d_ids || !d_types || !d_word || !d_pos || !d_seg || !d_gamma || !d_beta || !d_out) return fail(nullptr, HR_EINVAL, "null buffer");
    if (((uintptr_t)d_word | (uintptr_t)d_pos | (uintptr_t)d_seg | (uintptr_t)d_gamma | (uintptr_t)d_beta | (uintptr_t)d_out) & 15)
        return fail(nullptr, HR_EINVAL, "buffers must be 16-byte aligned");
    const int64_t rows = n_seq * T;
    if (rows == 0) return HR_OK;
    const int chunks = hidden / 8;
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    auto* w = (const half8_t*)d_word;
    auto* p = (const half8_t*)d_pos;
    auto* sg = (const half8_t*)d_seg;
    auto* g = (const half8_t*)d_gamma;
    auto* b = (const half8_t*)d_beta;
    auto* o = (half8_t*)d_out;
    hipStream_t s = (hipStream_t)stream;
    if (chunks <= 16) hipLaunchKernelGGL((embed_layernorm_f16_kernel<1>), grid, block, 0, s, d_ids, d_types, w, p, sg, g, b, o, rows, T, chunks, eps, n_word, n_seg);
    else if (chunks <= 32) hipLaunchKernelGGL((embed_layernorm_f16_kernel<2>), grid, block, 0, s, d_ids, d_types, w, p, sg, g, b, o, rows, T, chunks, eps, n_word, n_seg);
    else if (chunks <= 48) hipLaunchKernelGGL((embed_layernorm_f16_kernel<3>), grid, block, 0, s, d_ids, d_types, w, p, sg, g, b, o, rows, T, chunks, eps, n_word, n_seg);
    else if (chunks <= 64) hipLaunchKernelGGL((embed_layernorm_f16_kernel<4>), grid, block, 0, s, d_ids, d_types, w, p, sg, g, b, o, rows, T, chunks, eps, n_word, n_seg);
    else hipLaunchKernelGGL((embed_layernorm_f16_kernel<8>), grid, block, 0, s, d_ids, d_types, w, p, sg, g, b, o, rows, T, chunks, eps, n_word, n_seg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "embed_layernorm_f16_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

static int attention_launch(const _Float16* q, int64_t q_seq, int64_t q_tok, const _Float16* k, const _Float16* v, int64_t kv_seq,
                            int64_t kv_tok, const int32_t* d_lengths, _Float16* out, int64_t n_seq, int T, int n_queries, int heads,
                            int head_dim, float scale, hipStream_t stream, bool out_fr = false) {
    if (n_seq < 0 || T <= 0 || heads <= 0 || n_queries <= 0 || n_queries > T) return fail(nullptr, HR_EINVAL, "bad attention sizes");
    if (head_dim != 32 && head_dim != 64) return fail(nullptr, HR_ELIMIT, "the attention kernels serve head dimensions 32 and 64 (got %d)", head_dim);
    const int max_t = kAttnLdsBytes / (4 * head_dim);                  // K + V of a (sequence, head) must fit the block's LDS
    if (T > max_t) return fail(nullptr, HR_ELIMIT, "sequence length %d exceeds %d at head dimension %d", T, max_t, head_dim);
    if (!q || !k || !v || !out) return fail(nullptr, HR_EINVAL, "null buffer");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return fail(nullptr, HR_EINVAL, "buffers must be 16-byte aligned");
    if ((q_seq | q_tok | kv_seq | kv_tok) & 7) return fail(nullptr, HR_EINVAL, "strides must be multiples of 8 halves");
    if (n_seq == 0) return HR_OK;
    if (n_seq * heads > 0x7fffffffll) return fail(nullptr, HR_ELIMIT, "too many (sequence, head) pairs");
    const int n_chunks = (T + 31) / 32;
    const size_t lds = (size_t)n_chunks * 32 * 4 * head_dim;           // K and V^T of a (sequence, head)
    static bool attr_set = false;
    if (!attr_set) {
        const void* kernels[4] = {(const void*)attention_kernel<4, 32>, (const void*)attention_kernel<8, 32>,
                                  (const void*)attention_kernel<4, 64>, (const void*)attention_kernel<8, 64>};
        for (const void* f : kernels) {
            hipError_t e0 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kAttnLdsBytes);
            if (e0 != hipSuccess) return fail(nullptr, HR_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e0));
        }
        attr_set = true;
    }
    const int NW = n_queries <= 128 ? 4 : 8;                           // waves (x 32 queries) per block
    const int HG = head_dim == 32 ? 2 : 1;                             // heads per 128-byte line of the QKV buffer
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lengths = d_lengths;
    a.q_seq = q_seq; a.q_tok = q_tok; a.kv_seq = kv_seq; a.kv_tok = kv_tok;
    a.T = T; a.n_queries = n_queries; a.heads = heads;
    a.n_qblocks = (n_queries + 32 * NW - 1) / (32 * NW);
    a.n_pairs = n_seq * ((heads + HG - 1) / HG);                       // head groups: the blocks of a group share an XCD
    a.scale_log2e = scale * 1.4426950408889634f;
    a.out_fr = out_fr ? 1 : 0;
    const int64_t n_blocks = ((a.n_pairs + 7) / 8) * 8 * HG * a.n_qblocks;
    if (n_blocks > 0x7fffffffll) return fail(nullptr, HR_ELIMIT, "too many attention blocks");
    if (head_dim == 32) {
        if (NW == 4) hipLaunchKernelGGL((attention_kernel<4, 32>), dim3((unsigned)n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((attention_kernel<8, 32>), dim3((unsigned)n_blocks), dim3(512), lds, stream, a);
    } else {
        if (NW == 4) hipLaunchKernelGGL((attention_kernel<4, 64>), dim3((unsigned)n_blocks), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((attention_kernel<8, 64>), dim3((unsigned)n_blocks), dim3(512), lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "attention_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_attention_f16_dev(const void* d_qkv, const int32_t* d_lengths, void* d_out, int64_t n_seq, int T, int heads,
                         int head_dim, float scale, void* stream) {
    if (!d_qkv) return fail(nullptr, HR_EINVAL, "null buffer");
    const int64_t H = (int64_t)heads * head_dim;
    const _Float16* qkv = (const _Float16*)d_qkv;
    return attention_launch(qkv, (int64_t)T * 3 * H, 3 * H, qkv + H, qkv + 2 * H, (int64_t)T * 3 * H, 3 * H, d_lengths, (_Float16*)d_out,
                            n_seq, T, T, heads, head_dim, scale, (hipStream_t)stream);
}

int hr_attention_fr_f16_dev(const void* d_qkv, const int32_t* d_lengths, void* d_out_fr, int64_t n_seq, int T, int heads,
                            int head_dim, float scale, void* stream) {
    if (!d_qkv) return fail(nullptr, HR_EINVAL, "null buffer");
    const int64_t H = (int64_t)heads * head_dim;
    if (H % 32 != 0) return fail(nullptr, HR_EINVAL, "fragment-order output needs a hidden size that is a multiple of 32");
    const _Float16* qkv = (const _Float16*)d_qkv;
    return attention_launch(qkv, (int64_t)T * 3 * H, 3 * H, qkv + H, qkv + 2 * H, (int64_t)T * 3 * H, 3 * H, d_lengths, (_Float16*)d_out_fr,
                            n_seq, T, T, heads, head_dim, scale, (hipStream_t)stream, true);
}

int hr_attention_rows_f16_dev(const void* d_q, int64_t q_seq_stride, int64_t q_token_stride, const void* d_k, const void* d_v,
                              int64_t kv_seq_stride, int64_t kv_token_stride, const int32_t* d_lengths, void* d_out, int64_t n_seq,
                              int T, int n_queries, int heads, int head_dim, float scale, void* stream) {
    return attention_launch((const _Float16*)d_q, q_seq_stride, q_token_stride, (const _Float16*)d_k, (const _Float16*)d_v, kv_seq_stride,
                            kv_token_stride, d_lengths, (_Float16*)d_out, n_seq, T, n_queries, heads, head_dim, scale,
                            (hipStream_t)stream);
}

// ---- encoder layer kernels (csrc/encoder_layer.h) -----------------------------------------------------------------
static int el_allow_lds(const void* kernel, size_t bytes) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? HR_OK : fail(nullptr, HR_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
}

int hr_linear_rows_f16_dev(const void* d_x, int x_fr, const void* d_w_packed, const float* d_bias, void* d_out, int64_t rows, int K, int N,
                           int64_t out_stride, void* stream) {
    if (rows < 0 || K <= 0 || N <= 0 || N % 32 != 0 || out_stride < N || out_stride % 4 != 0) return fail(nullptr, HR_EINVAL, "bad linear sizes");
    if (K != 384) return fail(nullptr, HR_ELIMIT, "the hand-written linear kernel serves K = 384 (got %d)", K);
    if (N > 4096) return fail(nullptr, HR_ELIMIT, "N exceeds 4096");
    if (!d_x || !d_w_packed || !d_bias || !d_out) return fail(nullptr, HR_EINVAL, "null buffer");
    if (((uintptr_t)d_x | (uintptr_t)d_w_packed | (uintptr_t)d_bias | (uintptr_t)d_out) & 15) return fail(nullptr, HR_EINVAL, "buffers must be 16-byte aligned");
    if (rows == 0) return HR_OK;
    constexpr int KS = 12;
    const size_t lds = (size_t)(kElRingStages + 1) * 2 * KS * 1024 + (size_t)N * 4;
    static bool ready = false;
    if (!ready) {
        HR_TRY(el_allow_lds((const void*)linear_rows_kernel<KS, 2>, 160 * 1024));
        HR_TRY(el_allow_lds((const void*)linear_rows_kernel<KS, 4>, 160 * 1024));
        ready = true;
    }
    LinearArgs a{};
    a.x = (const _Float16*)d_x; a.w = (const chunk_t*)d_w_packed; a.bias = d_bias; a.out = (_Float16*)d_out;
    a.M = rows; a.out_stride = out_stride; a.N = N; a.x_fr = x_fr ? 1 : 0;
    // Four token tiles per wave (a weight fragment read from LDS feeds four MFMAs instead of two: 0.367 -> 0.335 ms for
    // 327 680 rows x 1 152 outputs) once blocks of 256 rows still fill the chip; two below that, where a block's latency counts
    const bool wide = rows >= 256 * 256;
    const int rows_per_block = wide ? 256 : 128;
    const int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffffll) return fail(nullptr, HR_ELIMIT, "too many rows");
    if (wide) hipLaunchKernelGGL((linear_rows_kernel<KS, 4>), dim3((unsigned)blocks), dim3(64 * kElWaves), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear_rows_kernel<KS, 2>), dim3((unsigned)blocks), dim3(64 * kElWaves), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "linear_rows_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

int hr_encoder_tail_f16_dev(const void* d_attn_fr, const void* d_x, int x_fr, void* d_out, int out_fr, const void* d_wstream,
                            const float* d_tables, int64_t rows, int hidden, int intermediate, float eps, int gelu_erf, void* stream) {
    if (rows < 0 || hidden <= 0 || intermediate <= 0) return fail(nullptr, HR_EINVAL, "bad encoder-tail sizes");
    if (hidden != 384 || intermediate != 1536)
        return fail(nullptr, HR_ELIMIT, "the fused layer tail serves (hidden, intermediate) = (384, 1536) (got %d, %d)", hidden, intermediate);
    if (!d_attn_fr || !d_x || !d_out || !d_wstream || !d_tables) return fail(nullptr, HR_EINVAL, "null buffer");
    if (((uintptr_t)d_attn_fr | (uintptr_t)d_x | (uintptr_t)d_out | (uintptr_t)d_wstream | (uintptr_t)d_tables) & 15)
        return fail(nullptr, HR_EINVAL, "buffers must be 16-byte aligned");
    if (rows == 0) return HR_OK;
    constexpr int HS = 12, IS = 48, TT = 2;
    const size_t lds = (size_t)kElRingStages * 2 * HS * 1024 + (size_t)(6 * hidden + intermediate) * 4;
    static bool ready = false;
    if (!ready) {
        HR_TRY(el_allow_lds((const void*)encoder_tail_kernel<HS, IS, TT, false>, 160 * 1024));
        HR_TRY(el_allow_lds((const void*)encoder_tail_kernel<HS, IS, TT, true>, 160 * 1024));
        ready = true;
    }
    TailArgs a{};
    a.a = (const _Float16*)d_attn_fr; a.x = (const _Float16*)d_x; a.out = (_Float16*)d_out; a.x_fr = x_fr ? 1 : 0; a.out_fr = out_fr ? 1 : 0;
    a.wstream = (const chunk_t*)d_wstream; a.tables = d_tables; a.M = rows; a.eps = eps;
    const int64_t blocks = (rows + 64 * TT - 1) / (64 * TT);
    if (blocks > 0x7fffffffll) return fail(nullptr, HR_ELIMIT, "too many rows");
    if (gelu_erf)
        hipLaunchKernelGGL((encoder_tail_kernel<HS, IS, TT, true>), dim3((unsigned)blocks), dim3(64 * kElWaves), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((encoder_tail_kernel<HS, IS, TT, false>), dim3((unsigned)blocks), dim3(64 * kElWaves), lds, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, HR_EHIP, "encoder_tail_kernel: %s", hipGetErrorString(e));
    return HR_OK;
}

// ---- host-buffer, synchronous forms -------------------------------------------------
static int search_dense_host(hr_index* h, const float* q, int B, int k, const uint8_t* rowmask, bool mask_on_device,
                             int64_t* out_ids, float* out_scores) {
    HR_TRY(check_search_args(h, B, k, true));
    if (!q || !out_ids || !out_scores) return fail(h, HR_EINVAL, "null buffer");
    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    if (h->n_rows == 0) {
        std::fill(out_ids, out_ids + (size_t)B * k, (int64_t)-1);
        std::fill(out_scores, out_scores + (size_t)B * k, 0.f);
        return HR_OK;
    }
    Workspace* ws = take_ws(h);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    struct Giver { hr_index* h; Workspace* w; ~Giver() { give_ws(h, w); } } giver{h, ws};
    hipStream_t s = ws->stream;
    HIP_TRY(h, ws->d_q.ensure((size_t)B * h->dim * 4));
    HIP_TRY(h, ws->d_ids.ensure((size_t)B * k * 8));
    HIP_TRY(h, ws->d_scores.ensure((size_t)B * k * 4));
    HIP_TRY(h, ws->flags.ensure((size_t)B * 4));
    HIP_TRY(h, hipMemcpyAsync(ws->d_q.p, q, (size_t)B * h->dim * 4, hipMemcpyHostToDevice, s));
    const uint8_t* d_mask = mask_on_device ? rowmask : nullptr;
    if (rowmask && !mask_on_device) {
        const size_t mb = (size_t)(h->n_rows + 7) / 8;
        HIP_TRY(h, ws->d_mask.ensure(mb));
        HIP_TRY(h, hipMemcpyAsync(ws->d_mask.p, rowmask, mb, hipMemcpyHostToDevice, s));
        d_mask = ws->d_mask.as<uint8_t>();
    }
    const int64_t n_groups = (h->n_rows + group_rows_for(h, h->n_rows) - 1) / group_rows_for(h, h->n_rows);
    int C = candidate_groups_for_k(k);
    std::vector<int32_t> flags(B);
    for (;;) {
        HR_TRY(dense_search_enqueue(h, ws, s, ws->d_q.as<float>(), B, k, d_mask, ws->d_ids.as<int64_t>(),
                                    ws->d_scores.as<float>(), ws->flags.as<int32_t>(), C));
        HIP_TRY(h, hipMemcpyAsync(flags.data(), ws->flags.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        const bool all_exact = std::all_of(flags.begin(), flags.end(), [](int32_t f) { return f != 0; });
        if (all_exact || C >= n_groups) break;
        C = next_candidate_count(C, n_groups);  // widen the candidate set and redo
    }
    HIP_TRY(h, hipMemcpyAsync(out_ids, ws->d_ids.p, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(out_scores, ws->d_scores.p, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HR_OK;
}

int hr_search_dense(hr_index* h, const float* q, int B, int k, const uint8_t* rowmask, int64_t* out_ids,
                    float* out_scores) {
    return search_dense_host(h, q, B, k, rowmask, false, out_ids, out_scores);
}
int hr_search_dense_dmask(hr_index* h, const float* q, int B, int k, const uint8_t* d_rowmask, int64_t* out_ids,
                          float* out_scores) {
    return search_dense_host(h, q, B, k, d_rowmask, true, out_ids, out_scores);
}

static int search_sparse_host(hr_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val, int B, int k,
                              float drop_ratio, const uint8_t* rowmask, bool mask_on_device, int64_t* out_ids,
                              float* out_scores) {
    HR_TRY(check_search_args(h, B, k, false));
    if (!q_indptr || !out_ids || !out_scores) return fail(h, HR_EINVAL, "null buffer");
    if (!(drop_ratio >= 0.f && drop_ratio < 1.f)) return fail(h, HR_EINVAL, "drop_ratio must be in [0,1)");
    // Host prep: drop the floor(drop_ratio*nnz) smallest-|value| entries (ties:
    // the later entry goes first), then order by index.
    std::vector<int64_t> ptr(B + 1, 0);
    std::vector<int32_t> idx;
    std::vector<float> val;
    int max_nnz = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t e0 = q_indptr[b], e1 = q_indptr[b + 1];
        if (e1 < e0) return fail(h, HR_EINVAL, "query indptr not monotone");
        const int nnz = (int)(e1 - e0);
        if (nnz > 0 && (!q_idx || !q_val)) return fail(h, HR_EINVAL, "null query arrays");
        std::vector<int> order(nnz);
        for (int i = 0; i < nnz; ++i) order[i] = i;
        const int n_drop = (int)std::floor((double)drop_ratio * nnz);
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
            const float fa = std::fabs(q_val[e0 + a]), fc = std::fabs(q_val[e0 + c]);
            if (fa != fc) return fa < fc;
            return a > c;
        });
        std::vector<int> keep(order.begin() + n_drop, order.end());
        std::sort(keep.begin(), keep.end(), [&](int a, int c) { return q_idx[e0 + a] < q_idx[e0 + c]; });
        int32_t prev = -1;
        for (int i : keep) {
            const int32_t t = q_idx[e0 + i];
            if (t < 0 || t >= h->sparse_dim) return fail(h, HR_EINVAL, "query index %d out of range", t);
            if (t == prev) return fail(h, HR_EINVAL, "duplicate query index %d", t);
            prev = t;
            idx.push_back(t);
            val.push_back(q_val[e0 + i]);
        }
        ptr[b + 1] = (int64_t)idx.size();
        max_nnz = std::max(max_nnz, (int)keep.size());
    }
    if (max_nnz > HR_MAX_QUERY_NNZ) return fail(h, HR_ELIMIT, "query nnz %d exceeds HR_MAX_QUERY_NNZ", max_nnz);

    std::shared_lock<std::shared_mutex> lk(h->rw);
    DeviceGuard dg(h->device);
    if (h->n_sparse == 0) {
        std::fill(out_ids, out_ids + (size_t)B * k, (int64_t)-1);
        std::fill(out_scores, out_scores + (size_t)B * k, 0.f);
        return HR_OK;
    }
    Workspace* ws = take_ws(h);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    struct Giver { hr_index* h; Workspace* w; ~Giver() { give_ws(h, w); } } giver{h, ws};
    hipStream_t s = ws->stream;
    const size_t nnz = idx.size();
    HIP_TRY(h, ws->d_qptr.ensure((size_t)(B + 1) * 8));
    HIP_TRY(h, ws->d_qidx.ensure(std::max<size_t>(nnz, 1) * 4));
    HIP_TRY(h, ws->d_qval.ensure(std::max<size_t>(nnz, 1) * 4));
    HIP_TRY(h, ws->d_ids.ensure((size_t)B * k * 8));
    HIP_TRY(h, ws->d_scores.ensure((size_t)B * k * 4));
    HIP_TRY(h, ws->flags.ensure((size_t)B * 4));
    HIP_TRY(h, hipMemcpyAsync(ws->d_qptr.p, ptr.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, s));
    if (nnz) {
        HIP_TRY(h, hipMemcpyAsync(ws->d_qidx.p, idx.data(), nnz * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(ws->d_qval.p, val.data(), nnz * 4, hipMemcpyHostToDevice, s));
    }
    const uint8_t* d_mask = mask_on_device ? rowmask : nullptr;
    if (rowmask && !mask_on_device) {
        const size_t mb = (size_t)(h->n_sparse + 7) / 8;
        HIP_TRY(h, ws->d_mask.ensure(mb));
        HIP_TRY(h, hipMemcpyAsync(ws->d_mask.p, rowmask, mb, hipMemcpyHostToDevice, s));
        d_mask = ws->d_mask.as<uint8_t>();
    }
    const int64_t n_groups = (h->n_sparse + group_rows_for(h, h->n_sparse) - 1) / group_rows_for(h, h->n_sparse);
    int C = candidate_groups_for_k(k);
    std::vector<int32_t> flags(B);
    for (;;) {
        HR_TRY(sparse_search_enqueue(h, ws, s, ws->d_qptr.as<int64_t>(), ws->d_qidx.as<int32_t>(),
                                     ws->d_qval.as<float>(), B, max_nnz, k, d_mask, ws->d_ids.as<int64_t>(),
                                     ws->d_scores.as<float>(), ws->flags.as<int32_t>(), C));
        HIP_TRY(h, hipMemcpyAsync(flags.data(), ws->flags.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        const bool all_exact = std::all_of(flags.begin(), flags.end(), [](int32_t f) { return f != 0; });
        if (all_exact || C >= n_groups) break;
        C = next_candidate_count(C, n_groups);
    }
    HIP_TRY(h, hipMemcpyAsync(out_ids, ws->d_ids.p, (size_t)B * k * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(out_scores, ws->d_scores.p, (size_t)B * k * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HR_OK;
}

int hr_search_sparse(hr_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val, int B, int k,
                     float drop_ratio, const uint8_t* rowmask, int64_t* out_ids, float* out_scores) {
    return search_sparse_host(h, q_indptr, q_idx, q_val, B, k, drop_ratio, rowmask, false, out_ids, out_scores);
}
int hr_search_sparse_dmask(hr_index* h, const int64_t* q_indptr, const int32_t* q_idx, const float* q_val, int B, int k,
                           float drop_ratio, const uint8_t* d_rowmask, int64_t* out_ids, float* out_scores) {
    return search_sparse_host(h, q_indptr, q_idx, q_val, B, k, drop_ratio, d_rowmask, true, out_ids, out_scores);
}

int hr_fuse_rrf(hr_index* h, const int64_t* ids_a, int na, const int64_t* ids_b, int nb, const int64_t* ids_c, int nc,
                double wa, double wb, double wc, int rrf_k, int64_t* out_ids, double* out_scores, int32_t* out_methods,
                int32_t* n_out) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    if (na < 0 || nb < 0 || nc < 0 || na + nb + nc == 0) {
        if (n_out) *n_out = 0;
        return (na < 0 || nb < 0 || nc < 0) ? fail(h, HR_EINVAL, "negative list length") : HR_OK;
    }
    if (na > HR_MAX_TOPK || nb > HR_MAX_TOPK || nc > HR_MAX_TOPK)
        return fail(h, HR_ELIMIT, "list longer than HR_MAX_TOPK=%d", HR_MAX_TOPK);
    if (!out_ids || !out_scores || !out_methods || !n_out) return fail(h, HR_EINVAL, "null buffer");
    DeviceGuard dg(h->device);
    Workspace* ws = take_ws(h);
    if (!ws) return fail(h, HR_ENOMEM, "workspace allocation failed");
    struct Giver { hr_index* h; Workspace* w; ~Giver() { give_ws(h, w); } } giver{h, ws};
    hipStream_t s = ws->stream;
    const int total = na + nb + nc;
    const int ka = std::max(na, 1);  // kernel wants a readable first list
    std::vector<int64_t> staged((size_t)ka + nb + nc, -1);
    if (na) std::memcpy(staged.data(), ids_a, (size_t)na * 8);
    if (nb) std::memcpy(staged.data() + ka, ids_b, (size_t)nb * 8);
    if (nc) std::memcpy(staged.data() + ka + nb, ids_c, (size_t)nc * 8);
    HIP_TRY(h, ws->f_ids.ensure(staged.size() * 8));
    HIP_TRY(h, ws->f_out_ids.ensure((size_t)total * 8));
    HIP_TRY(h, ws->f_out_scores.ensure((size_t)total * 8));
    HIP_TRY(h, ws->f_out_meth.ensure((size_t)total * 4));
    HIP_TRY(h, ws->f_n.ensure(4));
    HIP_TRY(h, hipMemcpyAsync(ws->f_ids.p, staged.data(), staged.size() * 8, hipMemcpyHostToDevice, s));
    int64_t* d = ws->f_ids.as<int64_t>();
    HR_TRY(hr_fuse_rrf_dev(d, ka, nb ? d + ka : nullptr, nb, nc ? d + ka + nb : nullptr, nc, 1, wa, wb, wc, rrf_k, total,
                           ws->f_out_ids.as<int64_t>(), ws->f_out_scores.as<double>(), ws->f_out_meth.as<int32_t>(),
                           ws->f_n.as<int32_t>(), s));
    HIP_TRY(h, hipMemcpyAsync(out_ids, ws->f_out_ids.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(out_scores, ws->f_out_scores.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(out_methods, ws->f_out_meth.p, (size_t)total * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_out, ws->f_n.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HR_OK;
}

// ---- measurement hooks -----------------------------------------------------------------
int hr_set_profiling(hr_index* h, int enabled) {
    if (!h) return fail(nullptr, HR_EINVAL, "null handle");
    h->profiling = enabled;
    return HR_OK;
}

#ifdef HR_STAMP
HR_API int hr_debug_finish_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hbmrag::hr_finish_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
HR_API int hr_debug_gemm_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hbmrag::hr_gemm_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif
int hr_last_kernel_ms(hr_index* h, float* out_ms, int n) {
    if (!h || !out_ms || n < 2 * PH_COUNT) return fail(h, HR_EINVAL, "need room for %d floats", 2 * PH_COUNT);
    DeviceGuard dg(h->device);
    std::lock_guard<std::mutex> g(h->prof_mu);
    double sum[PH_COUNT] = {0};
    int cnt[PH_COUNT] = {0};
    for (auto& sp : h->spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            sum[sp.phase] += ms;
            cnt[sp.phase] += 1;
        }
        h->event_pool.push_back(sp.a);
        h->event_pool.push_back(sp.b);
    }
    h->spans.clear();
    // out[0..8] = mean ms per launch of each phase, out[9..17] = launches averaged
    for (int p = 0; p < PH_COUNT; ++p) {
        out_ms[p] = cnt[p] ? (float)(sum[p] / cnt[p]) : 0.f;
        out_ms[PH_COUNT + p] = (float)cnt[p];
    }
    return HR_OK;
}

}  // extern "C"
