// Diagnostic: cost of a chain of dependent float adds in one wave (registers only / with 16-byte LDS reads and writes)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J>
__device__ inline void wl_steps(float &acc, float c, int &mine) {
    if constexpr (J < 64) {
        acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), J));
        asm volatile("v_readfirstlane_b32 s20, %1\n\ts_nop 3\n\tv_writelane_b32 %0, s20, %2" : "+v"(mine) : "v"(acc), "n"(J) : "s20");
        wl_steps<J + 1>(acc, c, mine);
    }
}
template <int mode>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *t, int n) {
    __shared__ __attribute__((aligned(16))) float pc[4096];
    __shared__ __attribute__((aligned(16))) float px[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) pc[i] = 1.0f + i * 1e-3f;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float acc = out[0];
    if constexpr (mode == 0) {
        float a = pc[1], b = pc[2], c = pc[3], d = pc[4];
        for (int i = 0; i < n; i += 4) { acc += a; acc += b; acc += c; acc += d; asm volatile("" : "+v"(acc)); }
    } else if constexpr (mode == 1 || mode == 2) {
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
    if constexpr (mode == 3) {
        // 64 items per step: every lane loads its own score, the chain runs on cross-lane reads (readlane -> SGPR operand),
        // lane j keeps the running sum behind item j, one coalesced LDS write per 64 items; the next 64 scores are
        // requested a step ahead
        acc = out[0];
        const int lane = threadIdx.x;
        float c = pc[lane];
        for (int i = 0; i < n; i += 64) {
            const float cn = pc[(i + 64 + lane) & 4095];
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), j));
                mine = lane == j ? acc : mine;
            }
            px[(i + lane) & 4095] = mine;
            c = cn;
        }
    }
    if constexpr (mode == 4) {
        // the same, the running sums written by v_writelane from a scalar copy (off the chain)
        acc = out[0];
        const int lane = threadIdx.x;
        float c = pc[lane];
        for (int i = 0; i < n; i += 64) {
            const float cn = pc[(i + 64 + lane) & 4095];
            int mine = 0;
            wl_steps<0>(acc, c, mine);
            px[(i + lane) & 4095] = __int_as_float(mine);
            c = cn;
        }
    }
    if constexpr (mode == 5 || mode == 6 || mode == 7) {
        // blocks of 16 items, three register sets in rotation (the hot owner's loop).  5: lane 0 writes (under exec),
        // 6: every lane writes (same address, same data), 7: lane l writes the l-th 16 bytes of the block -- one write
        // instruction per block, no exec change
        acc = out[0];
        const int lane = threadIdx.x;
        const bool wr = lane == 0;
        const float4 *c4 = reinterpret_cast<const float4 *>(pc);
        float4 *w4 = reinterpret_cast<float4 *>(px);
        const int nblk = n >> 4;
        auto load = [&](float4 (&q)[4], int blk) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = c4[(4 * blk + k) & 1023];
        };
        auto proc = [&](const float4 (&q)[4], int blk) {
            float4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float w0 = acc + q[k].x, w1 = w0 + q[k].y, w2 = w1 + q[k].z, w3 = w2 + q[k].w;
                acc = w3;
                r[k] = make_float4(w0, w1, w2, w3);
                if constexpr (mode == 5) { if (wr) w4[(4 * blk + k) & 1023] = r[k]; }
                if constexpr (mode == 6) w4[(4 * blk + k) & 1023] = r[k];
            }
            if constexpr (mode == 7) {
                const int l = lane & 3;
                float4 m;
                m.x = l == 0 ? r[0].x : l == 1 ? r[1].x : l == 2 ? r[2].x : r[3].x;
                m.y = l == 0 ? r[0].y : l == 1 ? r[1].y : l == 2 ? r[2].y : r[3].y;
                m.z = l == 0 ? r[0].z : l == 1 ? r[1].z : l == 2 ? r[2].z : r[3].z;
                m.w = l == 0 ? r[0].w : l == 1 ? r[1].w : l == 2 ? r[2].w : r[3].w;
                w4[(4 * blk + l) & 1023] = m;
            }
        };
        float4 qa[4], qb[4], qc[4];
        if (nblk > 0) load(qa, 0);
        if (nblk > 1) load(qb, 1);
        for (int blk = 0; blk < nblk; blk += 3) {
            if (blk + 2 < nblk) load(qc, blk + 2);
            proc(qa, blk);
            if (blk + 3 < nblk) load(qa, blk + 3);
            if (blk + 1 < nblk) proc(qb, blk + 1);
            if (blk + 4 < nblk) load(qb, blk + 4);
            if (blk + 2 < nblk) proc(qc, blk + 2);
        }
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[1] = acc + px[7]; t[0] = w1 - w0; t[1] = c1 - c0; }
}
int main() {
    float *o; unsigned long long *t, h[2];
    (void)hipMalloc(&o, 64); (void)hipMalloc(&t, 64); (void)hipMemset(o, 0, 64);
    const int n = 4096;
    for (int mode = 0; mode < 8; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, o, t, n); break;
                default: hipLaunchKernelGGL(k<7>, dim3(1), dim3(256), 0, 0, o, t, n); break;
            }
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
            printf("mode %d: %d adds: %.2f us wall (%.2f ns/add), %llu shader clocks (%.1f per add)\n", mode, n, h[0] / 100.0, h[0] * 10.0 / n, h[1], (double)h[1] / n);
        }
    return 0;
}
