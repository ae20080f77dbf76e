// Diagnostic: latency of a dependency between two HIP streams on MI355X.  Kernel A (stream 1) stamps its end, kernel B
// (stream 2) stamps its start; B waits for A through (0) nothing: same stream, (1) hipEventRecord + hipStreamWaitEvent,
// (2) A's completion event attached to the launch (hipExtLaunchKernelGGL stop event) + hipStreamWaitEvent,
// (3) hipStreamWriteValue32 behind A + hipStreamWaitValue32 on stream 2 (if the device supports it).
//   hipcc --offload-arch=gfx950 -O3 -o xq tools/micro/xq.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void kA(unsigned long long *t, int spin) {
    // ~spin microseconds of work, then the end stamp
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin * 100) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64();
}
__global__ void kB(unsigned long long *t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned long long *t; CK(hipMalloc(&t, 64));
    uint32_t *flag = nullptr;
    if (can) CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    if (flag) CK(hipMemset(flag, 0, 8));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    uint32_t seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 3 && !flag) continue;
        std::vector<double> us;
        for (int it = 0; it < 30; ++it) {
            ++seq;
            // B's stream is already blocked on the dependency when A finishes (the host is far ahead): enqueue B's side
            // first where the API allows it
            if (mode == 0) {
                hipLaunchKernelGGL(kA, dim3(64), dim3(64), 0, s1, t, 30);
                hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s1, t);
            } else if (mode == 1) {
                hipLaunchKernelGGL(kA, dim3(64), dim3(64), 0, s1, t, 30);
                CK(hipEventRecord(ev, s1));
                CK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s2, t);
            } else if (mode == 2) {
                hipExtLaunchKernelGGL(kA, dim3(64), dim3(64), 0, s1, nullptr, ev, 0, t, 30);
                CK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s2, t);
            } else {
                CK(hipStreamWaitValue32(s2, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(kB, dim3(1), dim3(64), 0, s2, t);
                hipLaunchKernelGGL(kA, dim3(64), dim3(64), 0, s1, t, 30);
                CK(hipStreamWriteValue32(s1, flag, seq, 0));
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            unsigned long long h[2];
            CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
            us.push_back(((double)h[1] - (double)h[0]) / 100.0);
        }
        std::sort(us.begin(), us.end());
        const char *names[] = {"same stream", "event record + wait", "kernel stop event + wait", "write value + wait value"};
        printf("%-28s end of A -> start of B: median %6.2f us  min %6.2f  max %6.2f\n", names[mode], us[15], us[0], us[29]);
    }
    return 0;
}
