// Diagnostic: how many 256-thread workgroups with a given LDS allocation are resident at once (start/end stamps, spin 20 us)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256) k(unsigned long long *out, int spin_ticks) {
    extern __shared__ unsigned int lds[];
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = wall_clock64() + lds[5] * 0; }
}
int main() {
    const int grid = 2560;
    unsigned long long *d; hipMalloc(&d, 16 * grid);
    std::vector<unsigned long long> h(2 * grid);
    for (int lds : {1024, 8192, 16384, 18980, 20480, 24576, 32768, 40960, 65536}) {
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, 2000); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), d, 16 * grid, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int i = 0; i < grid; ++i) t0 = std::min(t0, h[2 * i]);
        int at5 = 0; for (int i = 0; i < grid; ++i) if (h[2 * i] - t0 < 500) ++at5;      // started within 5 us (100 MHz clock)
        printf("lds %6d B: started within 5 us: %d of %d  (%.2f per CU)\n", lds, at5, grid, at5 / 256.0);
    }
    return 0;
}
