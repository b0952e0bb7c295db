// Lab (not product): gap patterns of attn_bwd_pipe.hip in isolation.  One wave per SIMD, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// bit 0: VALU trio (add, fma, exp)   bit 1: two ds_read_b64_tr_b16 -> AGPR   bit 2: ds_read_b128 -> fresh VGPRs
// bit 3: ds_read_b128 -> the A operand of the MFMA just issued (WAR)   bit 4: MFMA B operand from AGPR
// bit 5: cvt_pk every gap  bit 6: no MFMA
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ char smem[65536];
    asm volatile("" ::: "a0", "a255");
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    bf16x8 a[2], b;
    for (int i = 0; i < 8; ++i) { a[0][i] = a[1][i] = (__bf16)(float)threadIdx.x; b[i] = (__bf16)1.0f; }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    unsigned addr = (threadIdx.x & 63) * 16;
    bf16x8 l[4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!(MODE & 64)) {
                if (MODE & 16)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[128:131], %0" : "+v"(acc[m & 1]) : "v"(a[m & 1]));
                else
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a[m & 1]), "v"(b));
            }
            FENCE();
            if (MODE & 2) asm volatile("ds_read_b64_tr_b16 a[192:193], %0\n\tds_read_b64_tr_b16 a[194:195], %0 offset:2048" ::"v"(addr));
            if (MODE & 4) asm volatile("ds_read_b128 %0, %1" : "=v"(l[m & 3]) : "v"(addr));
            if (MODE & 8) asm volatile("ds_read_b128 %0, %1" : "=v"(a[m & 1]) : "v"(addr));
            if (MODE & 1) {
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[(2 * m) % 16]) : "v"(x[(2 * m + 3) % 16]), "v"(x[(2 * m + 5) % 16]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[(2 * m + 1) % 16]) : "v"(x[(2 * m + 7) % 16]), "v"(x[(2 * m) % 16]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(x[(2 * m + 8) % 16]) : "v"(x[(2 * m + 1) % 16]));
            }
            if (MODE & 32) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[(2 * m + 9) % 16]) : "v"(x[(2 * m + 10) % 16]), "v"(x[(2 * m + 11) % 16]));
            FENCE();
        }
        if (MODE & 14) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r] + x[r];
    for (int i = 0; i < 4; ++i) s += (float)l[i][0];
    s += (float)a[0][0] + (float)a[1][0];
    if (s == 12345.678f) out[threadIdx.x] = s + smem[threadIdx.x];
}
template <int MODE>
void run(const char* name, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s : %6.1f ns per gap\n", name, ms * 1e6 / iters / 8);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    run<0>("mfma (all VGPR)", d);
    run<16>("mfma (B in AGPR)", d);
    run<16 | 1>("mfma + add,fma,exp", d);
    run<16 | 1 | 32>("mfma + add,fma,exp,cvt", d);
    run<16 | 2>("mfma + 2 tr64->agpr", d);
    run<16 | 4>("mfma + b128->vgpr", d);
    run<16 | 8>("mfma + b128->A operand (WAR)", d);
    run<16 | 2 | 4>("mfma + 2 tr64 + b128", d);
    run<16 | 1 | 2 | 4>("mfma + add,fma,exp + 2 tr64 + b128", d);
    run<16 | 1 | 2 | 8>("mfma + add,fma,exp + 2 tr64 + b128(WAR)", d);
    run<64 | 1 | 2 | 4>("NO mfma: add,fma,exp + 2 tr64 + b128", d);
    run<64 | 2 | 4>("NO mfma: 2 tr64 + b128", d);
    run<64 | 1>("NO mfma: add,fma,exp", d);
    return 0;
}
