// Lab: what one dependent kernel boundary costs inside a HIP graph (and on a plain stream) on MI355X, as a function of
// what the kernels touch.   hipcc --offload-arch=gfx950 -O3 -o tools/graphgap_lab tools/graphgap_lab.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// every thread writes `wq` and reads `rq` 16-byte words, strided over the grid
__global__ __launch_bounds__(256) void touch_kernel(float4* wbuf, const float4* rbuf, int wq, int rq, float4* sink) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
    float4 acc = {0, 0, 0, 0};
    for (int k = 0; k < rq; ++k) {
        const float4 v = rbuf[t + k * n];
        acc.x += v.x;
        acc.y += v.y;
    }
    for (int k = 0; k < wq; ++k) wbuf[t + k * n] = acc;
    if (acc.x == 12345.f) sink[0] = acc;
}

int main() {
    const int NK = 45, grid = 256;
    float4 *wbuf, *rbuf, *sink;
    CK(hipMalloc(&wbuf, (size_t)grid * 256 * 16 * 64));
    CK(hipMalloc(&rbuf, (size_t)grid * 256 * 16 * 64));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(rbuf, 0, (size_t)grid * 256 * 16 * 64));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int cfgs[][2] = {{0, 0}, {1, 0}, {4, 0}, {16, 0}, {0, 4}, {0, 16}, {4, 8}};
    for (auto& c : cfgs) {
        const int wq = c[0], rq = c[1];
        // plain stream
        float best_s = 1e9f, best_g = 1e9f, one = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, wbuf, rbuf, wq, rq, sink);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_s) best_s = ms;
        }
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, wbuf, rbuf, wq, rq, sink);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best_g) best_g = ms;
        }
        // 10 back-to-back replays (as the sampling loop issues them)
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms10;
        CK(hipEventElapsedTime(&ms10, e0, e1));
        // one kernel alone
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(256), 0, st, wbuf, rbuf, wq, rq, sink);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < one) one = ms;
        }
        printf("write %5.1f MB read %5.1f MB per kernel: stream %.2f us/kernel, graph %.2f us/kernel, 10 replays %.2f us/kernel, "
               "single launch %.1f us\n",
               wq * grid * 256 * 16 / 1e6, rq * grid * 256 * 16 / 1e6, best_s * 1e3f / NK, best_g * 1e3f / NK,
               ms10 * 1e3f / (10 * NK), one * 1e3f);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
