// Which compute units does a CU-masked HIP stream reach?  Standalone probe (not part of libhbmrag):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_census tests/probes/cu_mask_census.hip && /tmp/cu_mask_census 32
// creates a stream whose mask has the first N bits set (argv[1]; default 32), launches a grid of short spinning blocks
// on it and on an unmasked stream, and prints how many distinct (XCC, SE, CU) places the blocks ran on, per XCC.
// Used once per round to check the statement in include/hbmrag.h that consecutive mask bits alternate over the XCDs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void census(unsigned* out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hwid;
    }
}

static void run(hipStream_t s, const char* name) {
    const int blocks = 8192;
    unsigned* d;
    hipMalloc(&d, blocks * 8);
    hipLaunchKernelGGL(census, dim3(blocks), dim3(256), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b + 1];
        // HW_ID: [11:8] CU id, [14:13] SE id (gfx9 layout); keep both as the place key
        per_xcc[h[2 * b]].insert(((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4));
    }
    size_t total = 0;
    printf("%s:", name);
    for (auto& kv : per_xcc) {
        printf(" xcc%u=%zu", kv.first, kv.second.size());
        total += kv.second.size();
    }
    printf("  total places=%zu\n", total);
    hipFree(d);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 32;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<uint32_t> low((cus + 31) / 32, 0), high((cus + 31) / 32, 0);
    for (int i = 0; i < cus; ++i) (i < n ? low : high)[i >> 5] |= 1u << (i & 31);
    hipStream_t s_all, s_low, s_high;
    hipStreamCreate(&s_all);
    if (hipExtStreamCreateWithCUMask(&s_low, (uint32_t)low.size(), low.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&s_high, (uint32_t)high.size(), high.data()) != hipSuccess) {
        printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    printf("device reports %d CUs; mask = first %d bits\n", cus, n);
    run(s_all, "unmasked ");
    run(s_low, "low bits ");
    run(s_high, "high bits");
    return 0;
}
