// Timing probe for csrc/encoder_layer.h (not a test): runs encoder_tail_kernel / linear_rows_kernel on random operands and
// prints ms per launch and the MFMA share; EL_ABLATE bits (compile time) remove one ingredient at a time:
//   1 = no refill DMA after the prologue   2 = no LDS fragment reads   4 = no barrier   8 = no activation
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DEL_ABLATE=n] -o tests/probes/bin/el_probe[_n] tests/probes/el_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../advanced-rag-milvus_amd/csrc/encoder_layer.h"
using namespace hbmrag;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 327680;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
#ifndef EL_LINEAR_TT
#define EL_LINEAR_TT 2
#endif
    constexpr int H = 384, I = 1536, HS = 12, IS = 48, TT = 2, LTT = EL_LINEAR_TT;   // LTT: token tiles per wave of the linear kernel
    const size_t stream_halves = (size_t)(HS + 2 * IS) * (H / 16) * 512;
    std::vector<_Float16> hw(stream_halves), ha((size_t)M * H);
    srand(1);
    for (auto& v : hw) v = (_Float16)((rand() % 2001 - 1000) * 5e-5f);
    for (auto& v : ha) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    std::vector<float> tb(6 * H + I, 0.f);
    for (int i = 0; i < H; ++i) tb[H + i] = tb[4 * H + i] = 1.f;
    _Float16 *dw, *da, *dx, *dout, *dqkv; float* dt;
    CK(hipMalloc(&dw, stream_halves * 2)); CK(hipMalloc(&da, (size_t)M * H * 2)); CK(hipMalloc(&dx, (size_t)M * H * 2));
    CK(hipMalloc(&dout, (size_t)M * H * 2)); CK(hipMalloc(&dqkv, (size_t)M * 3 * H * 2)); CK(hipMalloc(&dt, tb.size() * 4));
    CK(hipMemcpy(dw, hw.data(), stream_halves * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, ha.data(), (size_t)M * H * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, ha.data(), (size_t)M * H * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)encoder_tail_kernel<HS, IS, TT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)linear_rows_kernel<HS, LTT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    TailArgs a{}; a.a = da; a.x = dx; a.out = dout; a.wstream = (const chunk_t*)dw; a.tables = dt; a.M = M; a.eps = 1e-12f; a.x_fr = a.out_fr = argc > 3 ? atoi(argv[3]) : 1;
    LinearArgs l{}; l.x = dx; l.w = (const chunk_t*)dw; l.bias = dt; l.out = dqkv; l.M = M; l.out_stride = 3 * H; l.N = 3 * H; l.x_fr = argc > 3 ? atoi(argv[3]) : 1;
    const size_t lds_t = (size_t)kElRingStages * 2 * HS * 1024 + (size_t)(6 * H + I) * 4;
    const size_t lds_l = (size_t)(kElRingStages + 1) * 2 * HS * 1024 + (size_t)3 * H * 4;
    const unsigned blocks = (unsigned)((M + 64 * TT - 1) / (64 * TT));
    const unsigned blocks_l = (unsigned)((M + 64 * LTT - 1) / (64 * LTT));
#ifdef EL_STAMP
    unsigned long long* dst;
    CK(hipMalloc(&dst, (size_t)blocks * 8 * 8));
    a.stamps = dst;
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        for (int r = 0; r < 3; ++r) {
            if (which == 0) hipLaunchKernelGGL((encoder_tail_kernel<HS, IS, TT, false>), dim3(blocks), dim3(256), lds_t, 0, a);
            else hipLaunchKernelGGL((linear_rows_kernel<HS, LTT>), dim3(blocks_l), dim3(256), lds_l, 0, l);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (which == 0) hipLaunchKernelGGL((encoder_tail_kernel<HS, IS, TT, false>), dim3(blocks), dim3(256), lds_t, 0, a);
            else hipLaunchKernelGGL((linear_rows_kernel<HS, LTT>), dim3(blocks_l), dim3(256), lds_l, 0, l);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        const double flop = which == 0 ? 2.0 * M * ((double)H * H + 2.0 * H * I) : 2.0 * M * H * 3.0 * H;
        printf("%s ablate=%d M=%lld: %.3f ms  %.1f TFLOP/s = %.3f of 2500\n", which == 0 ? "encoder_tail" : "linear_qkv  ",
#ifdef EL_ABLATE
               EL_ABLATE,
#else
               0,
#endif
               (long long)M, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
    }
#ifdef EL_STAMP
    {   // phase shares of the tail kernel's blocks (s_memtime ticks at 100 MHz... on gfx950 the counter runs at the shader clock)
        std::vector<unsigned long long> hs((size_t)blocks * 8);
        hipLaunchKernelGGL((encoder_tail_kernel<HS, IS, TT, false>), dim3(blocks), dim3(256), lds_t, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
        const char* names[6] = {"prologue (DMA start, tables, a / x loads)", "out-projection", "LN1", "FFN loop", "LN2", "stores"};
        double sum[6] = {0}, tot = 0;
        for (unsigned b = 0; b < blocks; ++b) {
            for (int i = 0; i < 6; ++i) sum[i] += (double)(hs[b * 8 + i + 1] - hs[b * 8 + i]);
            tot += (double)(hs[b * 8 + 6] - hs[b * 8]);
        }
        for (int i = 0; i < 6; ++i) printf("  %-44s %9.0f ticks  %5.1f %%\n", names[i], sum[i] / blocks, 100.0 * sum[i] / tot);
        printf("  block total %.0f ticks\n", tot / blocks);
    }
#endif
    return 0;
}
