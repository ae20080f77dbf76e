// Diagnostic: ordered float running sum of n items by a systolic chain across the lanes of one wave
// (v_add_f32_dpp wave_shr:1: lane i adds its item to the running sum of lane i-1), checked against the host's loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int NOP>
__device__ inline float step(float run, float c) {
    if (NOP == 0) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(run) : "v"(c));
    if (NOP == 1) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(run) : "v"(c));
    return run;
}
template <int NOP>
__global__ void __launch_bounds__(64) k(const float *cs, float *ws, unsigned long long *t, int r, float acc0) {
    const int lane = threadIdx.x;
    const unsigned long long w0 = wall_clock64();
    float carry = acc0;
    float cn = lane < r ? cs[lane] : 0.0f;
    for (int base = 0; base < r; base += 64) {
        const float c = cn;
        const int nx = base + 64 + lane;
        cn = nx < r ? cs[nx] : 0.0f;
        float run = carry + c;
#pragma unroll
        for (int i = 0; i < 63; ++i) run = step<NOP>(run, c);
        if (base + lane < r) ws[base + lane] = run;
        carry = __builtin_amdgcn_readlane(run, 63);
    }
    const unsigned long long w1 = wall_clock64();
    if (lane == 0) t[0] = w1 - w0;
}
int main() {
    const int r = 4096;
    std::vector<float> h(r), ref(r), got(r);
    srand(5);
    for (int i = 0; i < r; ++i) h[i] = (float)rand() / RAND_MAX * 3.0f + 1e-3f;
    float acc = 1234.567f;
    for (int i = 0; i < r; ++i) { acc = acc + h[i]; ref[i] = acc; }
    float *dc, *dw; unsigned long long *dt, ht;
    (void)hipMalloc(&dc, 4 * r); (void)hipMalloc(&dw, 4 * r); (void)hipMalloc(&dt, 8);
    (void)hipMemcpy(dc, h.data(), 4 * r, hipMemcpyHostToDevice);
    for (int nop = 0; nop < 2; ++nop)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(dw, 0, 4 * r);
            if (nop == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dc, dw, dt, r, 1234.567f);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dc, dw, dt, r, 1234.567f);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(got.data(), dw, 4 * r, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&ht, dt, 8, hipMemcpyDeviceToHost);
            int bad = 0; for (int i = 0; i < r; ++i) if (got[i] != ref[i]) ++bad;
            printf("nop %d: %d items %.2f us (%.2f ns/item), mismatches %d (last got %.6f want %.6f)\n", nop, r, ht / 100.0, ht * 10.0 / r, bad, got[r - 1], ref[r - 1]);
        }
    return 0;
}
