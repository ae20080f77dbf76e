// Diagnostic: cost of a chain of dependent float adds in one wave (registers only / with 16-byte LDS reads and writes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *t, int n, int mode) {
    __shared__ __attribute__((aligned(16))) float pc[4096];
    __shared__ __attribute__((aligned(16))) float px[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) pc[i] = 1.0f + i * 1e-3f;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float acc = out[0];
    if (mode == 0) {
        float a = pc[1], b = pc[2], c = pc[3], d = pc[4];
        for (int i = 0; i < n; i += 4) { acc += a; acc += b; acc += c; acc += d; asm volatile("" : "+v"(acc)); }
    } else {
        const float4 *c4 = reinterpret_cast<const float4 *>(pc);
        float4 *w4 = reinterpret_cast<float4 *>(px);
        const bool wr = threadIdx.x == 0;
        float4 q = c4[0];
        for (int i = 0; i < n / 4; ++i) {
            const float4 nq = c4[(i + 1) & 1023];
            const float w0 = acc + q.x, w1 = w0 + q.y, w2 = w1 + q.z, w3 = w2 + q.w;
            acc = w3;
            if (mode == 1) { if (wr) w4[i & 1023] = make_float4(w0, w1, w2, w3); }
            else w4[i & 1023] = make_float4(w0, w1, w2, w3);
            q = nq;
        }
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[1] = acc + px[7]; t[0] = w1 - w0; t[1] = c1 - c0; }
}
int main() {
    float *o; unsigned long long *t, h[2];
    (void)hipMalloc(&o, 64); (void)hipMalloc(&t, 64); (void)hipMemset(o, 0, 64);
    const int n = 4096;
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, t, n, mode);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("mode %d: %d adds: %.2f us wall (%.2f ns/add), %llu shader clocks (%.1f per add)\n", mode, n, h[0] / 100.0, h[0] * 10.0 / n, h[1], (double)h[1] / n);
        }
    return 0;
}
