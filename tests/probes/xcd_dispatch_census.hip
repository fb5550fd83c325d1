// Does workgroup L of a grid run on XCD L % 8, also when the grid oversubscribes the chip and blocks take unequal time?
// Standalone probe (not part of libhbmrag):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_census tests/probes/xcd_dispatch_census.hip && /tmp/xcd_census
// Launches a 1-D grid and a 2-D grid (128 x 40, x fastest) of 512-thread blocks with 80 KB of LDS (two per CU, like
// sparse_scan_kernel) whose spin time varies per block, and prints, per grid, how many blocks ran on XCD (linear id % 8).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void census(unsigned* out, int spin) {
    __shared__ int pad[20000];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    pad[threadIdx.x] = (int)L;
    long long t0 = clock64();
    const int mine = spin + (int)((L * 2654435761u) >> 20) % spin;   // 1x .. 2x
    while (clock64() - t0 < mine) {}
    if (threadIdx.x == 0) out[L] = (xcc & 0xf) | (pad[1] & 0);
}

static void run(dim3 grid, const char* name) {
    const unsigned blocks = grid.x * grid.y;
    unsigned* d;
    hipMalloc(&d, blocks * 4);
    hipLaunchKernelGGL(census, grid, dim3(512), 0, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks);
    hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    int match[8] = {0};
    for (unsigned rot = 0; rot < 8; ++rot)
        for (unsigned b = 0; b < blocks; ++b) match[rot] += (h[b] == ((b + rot) & 7));
    int best = 0;
    for (int r = 1; r < 8; ++r) if (match[r] > match[best]) best = r;
    printf("%s: %u blocks, XCD == (L + %d) %% 8 for %d of them (%.1f %%); first 24:", name, blocks, best, match[best],
           100.0 * match[best] / blocks);
    for (int b = 0; b < 24; ++b) printf(" %u", h[b]);
    printf("  last 8:");
    for (unsigned b = blocks - 8; b < blocks; ++b) printf(" %u", h[b]);
    printf("\n");
    hipFree(d);
}

int main() {
    run(dim3(5120), "1-D grid      ");
    run(dim3(128, 40), "2-D 128 x 40  ");
    run(dim3(100, 40), "2-D 100 x 40  ");
    return 0;
}
