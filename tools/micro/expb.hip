// Diagnostic (Experiment B of the round-2 review): is the second read of the step's gradient served by the 256 MiB
// Infinity Cache?  The step reads the 109 MB gradient twice: k_grad_norm streams it in batch order, the gradient pass
// of the update reads the same rows again, in row-list order.  Here: a norm pass (N) and a permuted row pass (P) over one
// buffer of the same size, timed with HIP events,
//   P after N            the order of the step
//   P after N + flush    a 1 GiB read between them (nothing of the buffer is left in any cache)
//   P after P            the same pass twice
//   N after flush / N after N / N after the producer's write of the buffer
// plus streaming-read variants of N (rows per step, grid) for the roofline of a 109 MB stream.
//   hipcc --offload-arch=gfx950 -O3 -o expb tools/micro/expb.hip && ./expb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int D = 128, D4 = D / 4, LPG = 32, G = 256 / LPG;

template <int U>
__global__ void __launch_bounds__(256) k_norm(const float *__restrict__ g, int rows_total, int rows_per_wg, float *__restrict__ gn) {
    const int lig = threadIdx.x % LPG, grp = threadIdx.x / LPG;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, rows_total);
    for (int r = r0 + grp; r < r1; r += U * G) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = min(r + u * G, r1 - 1);
            v[u] = reinterpret_cast<const float4 *>(g + (size_t)rr * D)[lig];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
#pragma unroll
            for (int o = LPG >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, LPG);
            const int rr = r + u * G;
            if (rr < r1 && lig == 0) gn[rr] = sqrtf(a);
        }
    }
}

// rows in the order of a list (the update's gradient pass: rows of one table row are adjacent in the list, anywhere in
// the batch); eight rows in flight per group, one 16-byte sum per eight rows written
__global__ void __launch_bounds__(256) k_perm(const float *__restrict__ g, const int *__restrict__ order, int n, float4 *__restrict__ out) {
    const int lig = threadIdx.x % LPG, grp = threadIdx.x / LPG;
    const int per = 8;
    for (int base = (blockIdx.x * G + grp) * per; base < n; base += gridDim.x * G * per) {
        float4 v[per];
#pragma unroll
        for (int u = 0; u < per; ++u) {
            const int i = min(base + u, n - 1);
            v[u] = reinterpret_cast<const float4 *>(g + (size_t)order[i] * D)[lig];
        }
        float4 s = v[0];
#pragma unroll
        for (int u = 1; u < per; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        out[(size_t)(base / per) * LPG + lig] = s;
    }
}

int main() {
    const int n = 26 * 8192;
    float *g, *gn; float4 *out; int *order; char *flush;
    const size_t flush_bytes = (size_t)1 << 30;
    CK(hipMalloc(&g, (size_t)n * D * 4)); CK(hipMalloc(&gn, (size_t)n * 4)); CK(hipMalloc(&out, (size_t)(n / 8 + 1) * LPG * 16));
    CK(hipMalloc(&order, (size_t)n * 4)); CK(hipMalloc(&flush, flush_bytes));
    std::vector<float> h((size_t)n * D);
    std::mt19937 rng(1);
    for (auto &x : h) x = (float)(rng() & 0xffff) * 1e-4f;
    CK(hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> ord(n);
    std::iota(ord.begin(), ord.end(), 0);
    std::shuffle(ord.begin(), ord.end(), rng);
    CK(hipMemcpy(order, ord.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto N = [&](int variant) {
        switch (variant) {
            case 0: hipLaunchKernelGGL(k_norm<8>, dim3(n / 128), dim3(256), 0, st, g, n, 128, gn); break;     // the library's shape
            case 1: hipLaunchKernelGGL(k_norm<16>, dim3(n / 128), dim3(256), 0, st, g, n, 128, gn); break;
            case 2: hipLaunchKernelGGL(k_norm<8>, dim3(n / 64), dim3(256), 0, st, g, n, 64, gn); break;
            case 3: hipLaunchKernelGGL(k_norm<8>, dim3(n / 256), dim3(256), 0, st, g, n, 256, gn); break;
            case 4: hipLaunchKernelGGL(k_norm<4>, dim3(n / 64), dim3(256), 0, st, g, n, 64, gn); break;
            case 5: hipLaunchKernelGGL(k_norm<4>, dim3(n / 32), dim3(256), 0, st, g, n, 32, gn); break;
            default: hipLaunchKernelGGL(k_norm<16>, dim3(n / 256), dim3(256), 0, st, g, n, 256, gn); break;
        }
    };
    auto P = [&]() { hipLaunchKernelGGL(k_perm, dim3(2048), dim3(256), 0, st, g, order, n, out); };
    // two ways to empty the caches: a 1 GiB fill (leaves dirty lines behind: the pass that follows also pays for their
    // write-back) and a 1 GiB read (clean lines only)
    auto FW = [&]() { CK(hipMemsetAsync(flush, 1, flush_bytes, st)); };
    const int frows = (int)(flush_bytes / (D * 4));
    float *fgn; CK(hipMalloc(&fgn, (size_t)frows * 4));
    auto F = [&]() { hipLaunchKernelGGL(k_norm<8>, dim3(frows / 128), dim3(256), 0, st, (const float *)flush, frows, 128, fgn); };
    auto W = [&]() { CK(hipMemsetAsync(g, 0x3c, (size_t)n * D * 4, st)); };          // the producer writes the gradient
    // `before`: what runs ahead (untimed), `timed`: the pass under the events; median of 21
    auto run = [&](const char *name, auto before, auto timed) {
        std::vector<float> ms;
        for (int it = 0; it < 21; ++it) {
            before();
            CK(hipEventRecord(e0, st));
            timed();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double us = ms[10] * 1e3, tbs = (double)n * D * 4 / us / 1e6;
        printf("%-34s median %7.2f us  min %7.2f us   %.2f TB/s of the 109 MB\n", name, us, ms[0] * 1e3, tbs);
    };
    run("nothing (event pair overhead)", [&] { F(); }, [&] {});
    run("P after N (the step's order)", [&] { F(); N(0); }, P);
    run("P after N + flush", [&] { N(0); F(); }, P);
    run("P after P", [&] { F(); P(); }, P);
    run("N after flush", [&] { F(); }, [&] { N(0); });
    run("N after N", [&] { F(); N(0); }, [&] { N(0); });
    run("N after 1 GiB fill (dirty caches)", [&] { FW(); }, [&] { N(0); });
    run("N after the buffer is written", [&] { F(); W(); }, [&] { N(0); });
    run("P after write + N", [&] { F(); W(); N(0); }, P);
    run("N after P", [&] { F(); P(); }, [&] { N(0); });
    const char *names[] = {"N U=8 rows/wg=128 (library)", "N U=16 rows/wg=128", "N U=8 rows/wg=64", "N U=8 rows/wg=256", "N U=4 rows/wg=64", "N U=4 rows/wg=32", "N U=16 rows/wg=256"};
    for (int v = 0; v < 7; ++v) run(names[v], [&] { F(); }, [&] { N(v); });
    return 0;
}
