// Diagnostic: what a 109 MB stream of 512-byte rows costs on MI355X as a write, a copy and a gather-copy (the gather's
// shape: 213 k rows read from random places, written in order), with the store forms at hand.  HIP events, the event
// pair's own overhead printed first.   hipcc --offload-arch=gfx950 -O3 -o wr tools/micro/wr.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int D4 = 32;
template <int MODE> __device__ inline void st16(float4 *p, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = {v.x, v.y, v.z, v.w};
    if (MODE == 0) *p = v;
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
}
// rows per lane group of 32 lanes, 8 rows in flight; src == nullptr: write only; order == nullptr: rows in place
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ src, const int *__restrict__ order, float4 *__restrict__ dst, int n) {
    const int lig = threadIdx.x & 31, grp = (blockIdx.x * 256 + threadIdx.x) >> 5, groups = (gridDim.x * 256) >> 5;
    for (int base = grp * 8; base < n; base += groups * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(base + u, n - 1);
            v[u] = src ? src[(size_t)(order ? order[i] : i) * D4 + lig] : make_float4((float)i, 1.f, 2.f, 3.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (base + u < n) st16<MODE>(&dst[(size_t)(base + u) * D4 + lig], v[u]);
    }
}
int main() {
    const int n = 26 * 8192;
    float4 *a, *b; int *order; char *flush;
    const size_t bytes = (size_t)n * D4 * 16, rows_big = 4u << 20;            // gather source: 4 M rows = 2 GiB
    CK(hipMalloc(&a, rows_big * D4 * 16)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&order, (size_t)n * 4)); CK(hipMalloc(&flush, (size_t)1 << 30));
    CK(hipMemset(a, 0, rows_big * D4 * 16));
    std::vector<int> ord(n); std::mt19937 rng(1);
    for (auto &x : ord) x = (int)(rng() % rows_big);
    CK(hipMemcpy(order, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto F = [&] { hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, st, (const float4 *)flush, (const int *)nullptr, (float4 *)nullptr + 0, 0); };
    int cur = 0;
    auto run = [&](const char *name, auto fn, double mb) {
        std::vector<float> ms;
        for (int it = 0; it < 21; ++it) {
            cur = it % 18;                                            // a fresh 109 MB of the source every time (cold, clean caches)
            CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-44s median %7.2f us  min %7.2f us  (%.0f MB: %.2f TB/s at the median)\n", name, ms[10] * 1e3, ms[0] * 1e3, mb, mb / (ms[10] * 1e3) );
    };
    (void)F;
    const double mb = bytes / 1e6;
    run("nothing (event pair overhead)", [&] {}, 0);
#define RUN3(M, label) \
    run("write only, " label, [&] { hipLaunchKernelGGL(k<M>, dim3(2048), dim3(256), 0, st, (const float4 *)nullptr, (const int *)nullptr, b, n); }, mb); \
    run("copy in place order, " label, [&] { hipLaunchKernelGGL(k<M>, dim3(2048), dim3(256), 0, st, (const float4 *)a + (size_t)cur * n * D4, (const int *)nullptr, b, n); }, 2 * mb); \
    run("gather-copy (random rows of 2 GiB), " label, [&] { hipLaunchKernelGGL(k<M>, dim3(2048), dim3(256), 0, st, (const float4 *)a, (const int *)order, b, n); }, 2 * mb);
    RUN3(0, "plain")
    RUN3(1, "sc1")
    RUN3(2, "nt")
    RUN3(3, "sc0 sc1")
    for (int g : {832, 1024, 4096, 8192})
        run(g == 832 ? "gather-copy sc1, grid 832" : g == 1024 ? "gather-copy sc1, grid 1024" : g == 4096 ? "gather-copy sc1, grid 4096" : "gather-copy sc1, grid 8192",
            [&] { hipLaunchKernelGGL(k<1>, dim3(g), dim3(256), 0, st, (const float4 *)a, (const int *)order, b, n); }, 2 * mb);
    return 0;
}
