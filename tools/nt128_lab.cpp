// LAB ("the 4-wave <= 80 KiB workgroup so two workgroups share a CU and one's epilogue runs under
// the other's main loop ... measure it"): the NT projection GEMM with a 128 x 128 x 64 tile, FOUR waves and 64 KiB of
// LDS, so that two independent workgroups are resident per CU, against libmgx's 256 x 256 x 64 eight-wave kernel
// (one workgroup per CU).  C[M,N] = A[M,K] . Bt[N,K]^T + bias, bf16.  Same LDS image, LDS-DMA staging, MFMA
// (v_mfma_f32_16x16x32_bf16) and 16-byte epilogue stores as gemm_nt.hip; double-buffered K tiles, two barriers per
// K tile.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/nt128_lab.cpp -o tools/nt128_lab -Lmusicgeneration_amd
// -lmgx -Wl,-rpath,'$ORIGIN/../musicgeneration_amd'.  Run: tools/nt128_lab M N K [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
int mgx_gemm_nt_bf16(const void*, const void*, void*, void*, const float*, const void*, int, int, int, int, int, int, int,
                     float, unsigned long long, void*);
const char* mgx_last_error_string(void);
}
#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void fill_kernel(bf16_t* p, long n, unsigned seed) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        p[i] = (bf16_t)((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f);
    }
}
__global__ void ref_nt_kernel(const bf16_t* A, const bf16_t* Bt, const float* bias, const int* rows, float* out, int N, int K) {
    const int m = rows[blockIdx.y];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)Bt[(long)n * K + k];
    out[(long)blockIdx.y * N + n] = s + bias[n];
}

constexpr int T = 128, BK = 64, HT = 16384;            // tile edge, K tile, bytes of one operand tile (128 rows x 64 k)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    union { bf16x2 v; unsigned u; } x;
    x.v[0] = (bf16_t)a;
    x.v[1] = (bf16_t)b;
    return x.u;
}
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// M, N multiples of 128, K of 64 (lab shapes only)
__global__ __launch_bounds__(256, 2) void nt128_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                        bf16_t* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                        int K) {
    __shared__ __attribute__((aligned(16))) char smem[4 * HT];      // A slots 0, 1 | B slots 0, 1
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;                   // 64 rows x 64 columns per wave
    const int l15 = lane & 15, g4 = lane >> 4;
    // tile order as in gemm_nt.hip: bands of 16 tile rows, column-major inside a band, contiguous runs per XCD
    const int nbn = N / T, nbm = M / T;
    const int tile = xcd_remap(blockIdx.x, nbm * nbn);
    const int band = tile / (16 * nbn);
    const int bh = min(16, nbm - 16 * band);
    const int rem = tile - band * 16 * nbn;
    const int bn = rem / bh, bm = 16 * band + (rem - bn * bh);
    const int m0 = bm * T, n0 = bn * T;

    // staging: wave w brings rows 32 w + 8 i + (lane >> 3), i = 0..3, of each operand tile (one KiB piece each)
    unsigned sa[4], sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * w + 8 * i + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        sa[i] = (unsigned)(((long)(m0 + row) * K + ch * 8) * 2);
        sb[i] = (unsigned)(((long)(n0 + row) * K + ch * 8) * 2);
    }
    const int nk = K / BK;
    char* const stage_base = smem + 4096 * w;
    auto stage = [&](int d, int kt) {
        const long ko = (long)min(kt, nk - 1) * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(A + ko + sa[i]), (lds_void*)(stage_base + d * HT + 1024 * i), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(B + ko + sb[i]), (lds_void*)(stage_base + (2 + d) * HT + 1024 * i), 16, 0, 0);
    };
    const char* ard[2];
    const char* brd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int lo = 1024 * (l15 >> 3) + 128 * (l15 & 7) + 16 * ((4 * ks + g4) ^ ((l15 >> 1) & 7));
        ard[ks] = smem + lo + 8192 * wr;
        brd[ks] = smem + 2 * HT + lo + 8192 * wc;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    stage(1, 1);
    for (int t = 0; t < nk; ++t) {
        const int d = t & 1;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // K tile t landed (tile t + 1 may fly)
        __builtin_amdgcn_s_barrier();
        bf16x8 a[4][2], b[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i][ks] = *reinterpret_cast<const bf16x8*>(ard[ks] + d * HT + 2048 * i);
                b[i][ks] = *reinterpret_cast<const bf16x8*>(brd[ks] + d * HT + 2048 * i);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave holds its fragments: slot d is dead
        stage(d, t + 2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue from the accumulators: lane (m = l15, g4) holds n = 16 j + 4 g4 + r; the 16-lane-row swap of the packed
    // j / j + 1 words gives every lane 8 consecutive columns (gemm_nt.hip)
    const int nsw = (g4 & 1) ? 16 + 4 * (g4 - 1) : 4 * g4;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        f32x4 bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias + n0 + 64 * wc + 32 * jp + 16 * j + 4 * g4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 64 * wr + 16 * i + l15;
            unsigned o[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 v = acc[i][2 * jp + j] + bv[j];
                o[j][0] = pack2(v[0], v[1]);
                o[j][1] = pack2(v[2], v[3]);
            }
            auto s0 = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
            *reinterpret_cast<uint4*>(C + (long)m * N + n0 + 64 * wc + 32 * jp + nsw) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    const int rounds = argc > 4 ? atoi(argv[4]) : 9, reps = 10;
    if (M % 256 || N % 256 || K % 64) { fprintf(stderr, "M, N multiples of 256, K of 64\n"); return 1; }
    bf16_t *A, *Bt, *C0, *C1;
    float *bias, *ref;
    int* rows;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&Bt, (size_t)N * K * 2));
    CK(hipMalloc(&C0, (size_t)M * N * 2));
    CK(hipMalloc(&C1, (size_t)M * N * 2));
    CK(hipMalloc(&bias, (size_t)N * 4));
    const int NR = 32;
    CK(hipMalloc(&ref, (size_t)NR * N * 4));
    CK(hipMalloc(&rows, NR * 4));
    fill_kernel<<<1024, 256>>>(A, (long)M * K, 1u);
    fill_kernel<<<1024, 256>>>(Bt, (long)N * K, 2u);
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = 0.25f * (float)((i * 37) % 17 - 8);
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<int> hr(NR);
    for (int i = 0; i < NR; ++i) hr[i] = (int)(((long)i * 7919 + (i % 3) * 127) % M);
    hr[0] = 0; hr[1] = M - 1; hr[2] = 127; hr[3] = 128;
    CK(hipMemcpy(rows, hr.data(), NR * 4, hipMemcpyHostToDevice));
    ref_nt_kernel<<<dim3((N + 255) / 256, NR), 256>>>(A, Bt, bias, rows, ref, N, K);
    auto run128 = [&]() {
        nt128_kernel<<<dim3((M / T) * (N / T)), 256>>>((const char*)A, (const char*)Bt, C1, bias, M, N, K);
    };
    auto run256 = [&]() {
        if (mgx_gemm_nt_bf16(A, Bt, C0, nullptr, bias, nullptr, M, N, K, K, K, N, -1, 0.f, 0ull, nullptr)) {
            fprintf(stderr, "mgx_gemm_nt_bf16: %s\n", mgx_last_error_string());
            exit(3);
        }
    };
    run128();
    run256();
    CK(hipDeviceSynchronize());
    // check the sampled rows of both
    std::vector<float> href((size_t)NR * N);
    std::vector<bf16_t> row(N);
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
    for (int which = 0; which < 2; ++which) {
        double worst = 0;
        for (int r = 0; r < NR; ++r) {
            CK(hipMemcpy(row.data(), (which ? C1 : C0) + (long)hr[r] * N, N * 2, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; ++n) {
                const double want = href[(size_t)r * N + n], got = (float)row[n];
                worst = std::max(worst, fabs(got - want) / (1.0 + fabs(want)));
            }
        }
        printf("%s: worst relative error on %d sampled rows %.3e\n", which ? "nt128 (4 waves, 2 workgroups per CU)" : "libmgx 256x256 (8 waves)", NR, worst);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> t128, t256;
    for (int r = 0; r < rounds; ++r) {
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) which ? run128() : run256();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            (which ? t128 : t256).push_back(ms * 1e3 / reps);
        }
    }
    const double fl = 2.0 * M * N * K;
    printf("M=%d N=%d K=%d  libmgx 256x256: %.1f us (%.0f TFLOP/s)   nt128: %.1f us (%.0f TFLOP/s)\n", M, N, K, median(t256),
           fl / median(t256) / 1e6, median(t128), fl / median(t128) / 1e6);
    return 0;
}
