// Lab (not product): how many independent VALU / LDS instructions hide under one v_mfma_f32_32x32x16_bf16 when both
// come from the SAME wave (one wave per SIMD)?   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_lab.cpp -o tools/mfma_overlap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// KIND 0: v_fma_f32  1: v_pk_fma_f32  2: v_exp_f32  3: ds_read_b128  4: v_cvt_pk_bf16_f32   ACC 0: VGPR accumulators 1: AGPR
template <int N, int KIND, int ACC, int NM>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ char smem[65536];
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)threadIdx.x; b[i] = (__bf16)1.0f; }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    f32x2 y[8];
    for (int i = 0; i < 8; ++i) y[i] = f32x2{x[i], x[i + 8]};
    unsigned addr = (threadIdx.x & 63) * 16;
    bf16x8 l[4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (ACC == 0)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 1]) : "v"(a), "v"(b));
            FENCE();
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[n % 16]) : "v"(x[(n + 5) % 16]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(y[n % 8]) : "v"(y[(n + 3) % 8]));
                if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[n % 16]));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(l[n % 4]) : "v"(addr));
                if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[n % 16]) : "v"(x[(n + 1) % 16]), "v"(x[(n + 2) % 16]));
            }
            FENCE();
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r] + x[r];
    for (int i = 0; i < 8; ++i) s += y[i][0] + y[i][1];
    for (int i = 0; i < 4; ++i) s += (float)l[i][0];
    if (s == 12345.678f) out[threadIdx.x] = s + smem[threadIdx.x];
}
template <int N, int KIND, int ACC, int NM>
void run(const char* name, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<N, KIND, ACC, NM><<<256, 256>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<N, KIND, ACC, NM><<<256, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s acc=%s mfma/iter=%d ops/mfma=%2d : %7.1f ns per (mfma + ops)\n", name, ACC ? "agpr" : "vgpr", NM, N, ms * 1e6 / iters / (NM ? NM : 1));
}
template <int KIND, int ACC>
void sweep(const char* name, float* d) {
    run<0, KIND, ACC, 8>(name, d); run<2, KIND, ACC, 8>(name, d); run<4, KIND, ACC, 8>(name, d);
    run<6, KIND, ACC, 8>(name, d); run<8, KIND, ACC, 8>(name, d); run<12, KIND, ACC, 8>(name, d);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    sweep<0, 0>("v_fma", d); sweep<0, 1>("v_fma", d);
    sweep<1, 0>("v_pk_fma", d);
    sweep<2, 0>("v_exp", d);
    sweep<3, 0>("ds_read128", d);
    sweep<4, 0>("cvt_pk", d);
    return 0;
}
