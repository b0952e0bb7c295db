// Occupies `k` compute units for `usec` microseconds (one 256-thread workgroup each, spinning on the constant-rate
// real-time counter): a stand-in for the RCCL reduction kernels of an N > 1 run on a one-GPU box.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/lab/libcu_hog.so tools/lab/cu_hog.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void cu_hog_kernel(long ticks, int* sink) {
    // a register footprint like a collective kernel's (128 VGPRs): it cannot slip into the registers a GEMM leaves free
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const long t0 = wall_clock64();
    long n = 0;
    // exit condition every wave reaches: elapsed time, or an iteration bound
    while (wall_clock64() - t0 < ticks && n < (1L << 30)) ++n;
    if (n < 0) *sink = (int)n;
}

extern "C" int cu_hog_launch(int k, int usec, void* stream) {
    if (k <= 0 || usec <= 0 || usec > 100000) return 1;
    hipLaunchKernelGGL(cu_hog_kernel, dim3(k), dim3(256), 0, (hipStream_t)stream, (long)usec * 100, nullptr);  // 100 MHz
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
