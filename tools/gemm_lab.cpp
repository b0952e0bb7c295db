// Stand-alone timing / checking harness for the NT and TN projection GEMMs of libmgx (no torch: a fresh GPU box
// pays no import time).  Build: tools/build_lab.sh.  Run on the GPU box: tools/gemm_lab [nt|tn|mfma] ...
// Timings: interleaved rounds of every variant in one process, median over rounds, random bf16 operands in [-1, 1).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int mgx_gemm_nt_bf16(const void*, const void*, void*, void*, const float*, const void*, int, int, int, int, int, int, int,
                     float, unsigned long long, void*);
int mgx_gemm_nt_dfac_bf16(const void*, const void*, const void*, void*, float*, float*, int, int, int, int, int, int, int, float,
                          void*);
long mgx_gemm_nt_dfac_colpart_floats(int, int);
long mgx_gemm_tn_workspace_floats(int, int, int, int*);
int mgx_gemm_tn_bf16(const void*, const void*, void*, float*, int, int, int, int, int, int, void*);
long mgx_gemm_tn_pair_workspace_floats(int, int, int, int, int);
int mgx_gemm_tn_pair_bf16(const void*, const void*, void*, int, int, int, int, const void*, const void*, void*, int, int, int, int,
                          int, float*, int, void*);
const char* mgx_last_error_string(void);
}

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

typedef __bf16 bf16_t;

__global__ void fill_kernel(bf16_t* p, long n, unsigned seed) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        p[i] = (bf16_t)((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f);
    }
}
// reference rows: C[m, :] for the sampled rows, fp32 accumulation in k order
__global__ void ref_nt_kernel(const bf16_t* A, const bf16_t* Bt, const float* bias, const int* rows, float* out, int N, int K) {
    const int m = rows[blockIdx.y];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)Bt[(long)n * K + k];
    out[(long)blockIdx.y * N + n] = s + (bias ? bias[n] : 0.f);
}
__global__ void ref_tn_kernel(const bf16_t* A, const bf16_t* B, const int* rows, float* out, int M, int N, int K) {
    const int m = rows[blockIdx.y];
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)k * M + m] * (float)B[(long)k * N + n];
    out[(long)blockIdx.y * N + n] = s;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// register-only MFMA loops: the rate the chip holds on random operands with nothing else going on
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_only_kernel(const bf16_t* src, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(src + ((threadIdx.x * 4 + i) * 8 + blockIdx.x * 64) % 65536);
        b[i] = *reinterpret_cast<const bf16x8*>(src + ((threadIdx.x * 4 + i) * 8 + 32768 + blockIdx.x * 64) % 65536);
    }
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 1], acc[i & 3], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

struct Variant {
    std::string name;
    int kind;        // 0 r1, 1 lab variant 0, 2 lab variant 1, 3 production entry (bias), 4 production gelu+dropout
};

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static void run_nt(int M, int N, int K, int rounds, int reps) {
    bf16_t *A, *Bt, *C, *C2;
    float *bias, *ref, *colpart;
    int* rows;
    bf16_t* pre;
    CK(hipMalloc(&pre, (size_t)M * N * 2));
    CK(hipMalloc(&colpart, (size_t)mgx_gemm_nt_dfac_colpart_floats(M, N) * 4));
    fill_kernel<<<1024, 256>>>(pre, (long)M * N, 7u);
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&Bt, (size_t)N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&C2, (size_t)M * N * 2));
    CK(hipMalloc(&bias, (size_t)N * 4));
    const int NR = 48;
    CK(hipMalloc(&ref, (size_t)NR * N * 4));
    CK(hipMalloc(&rows, NR * 4));
    fill_kernel<<<1024, 256>>>(A, (long)M * K, 1u);
    fill_kernel<<<1024, 256>>>(Bt, (long)N * K, 2u);
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = 0.25f * (float)((i * 37) % 17 - 8);
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    std::vector<int> hr(NR);
    for (int i = 0; i < NR; ++i) hr[i] = (int)(((long)i * 7919 + (i % 3) * 255) % M);
    hr[0] = 0; hr[1] = M - 1; hr[2] = std::min(M - 1, 255); hr[3] = std::min(M - 1, 256);
    CK(hipMemcpy(rows, hr.data(), NR * 4, hipMemcpyHostToDevice));
    ref_nt_kernel<<<dim3((N + 255) / 256, NR), 256>>>(A, Bt, bias, rows, ref, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)NR * N);
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));

    std::vector<Variant> vs = {{"nt_bias", 3}, {"nt_act_gelu_drop", 4}, {"nt_act_gelu", 6}, {"nt_act_relu", 7}, {"nt_dfac", 5}, {"nt_dfac_nocolsum", 8}, {"nt_res", 9}};
    auto launch = [&](const Variant& v) -> int {
        switch (v.kind) {
            case 3: return mgx_gemm_nt_bf16(A, Bt, C, nullptr, bias, nullptr, M, N, K, K, K, N, -1, 0.f, 0, nullptr);
            case 5: return mgx_gemm_nt_dfac_bf16(A, Bt, pre, C, colpart, nullptr, M, N, K, K, K, N, 0, 0.1f, nullptr);
            case 8: return mgx_gemm_nt_dfac_bf16(A, Bt, pre, C, nullptr, nullptr, M, N, K, K, K, N, 0, 0.1f, nullptr);
            case 9: return mgx_gemm_nt_bf16(A, Bt, C, nullptr, bias, pre, M, N, K, K, K, N, -1, 0.f, 0, nullptr);
            case 7: return mgx_gemm_nt_bf16(A, Bt, C, C2, bias, nullptr, M, N, K, K, K, N, 1, 0.f, 1234, nullptr);
            case 6: return mgx_gemm_nt_bf16(A, Bt, C, C2, bias, nullptr, M, N, K, K, K, N, 0, 0.f, 1234, nullptr);
            default: return mgx_gemm_nt_bf16(A, Bt, C, C2, bias, nullptr, M, N, K, K, K, N, 0, 0.1f, 1234, nullptr);
        }
    };
    // correctness
    std::vector<bf16_t> hc((size_t)N);
    for (auto& v : vs) {
        if (v.kind == 0 && K % 64) continue;
        CK(hipMemset(C, 0xFF, (size_t)M * N * 2));
        int rc = launch(v);
        if (rc) { printf("  %-22s rc=%d %s\n", v.name.c_str(), rc, mgx_last_error_string()); continue; }
        CK(hipDeviceSynchronize());
        double maxerr = 0, maxref = 0;
        for (int r = 0; r < NR; ++r) {
            CK(hipMemcpy(hc.data(), C + (size_t)hr[r] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; ++n) {
                const double want = href[(size_t)r * N + n], got = (float)hc[n];
                maxerr = std::max(maxerr, std::fabs(want - got));
                maxref = std::max(maxref, std::fabs(want));
            }
        }
        if (v.kind == 3)
            printf("  check %-22s max|err|=%.4g (max|ref|=%.4g) %s\n", v.name.c_str(), maxerr, maxref,
                   maxerr <= 0.01 * maxref ? "ok" : "MISMATCH");
    }
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> t(vs.size());
    for (int r = 0; r < rounds + 1; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < reps; ++k) launch(vs[i]);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) t[i].push_back(ms / reps);
        }
    const double fl = 2.0 * M * N * K;
    for (size_t i = 0; i < vs.size(); ++i) {
        const double med = median(t[i]), mn = *std::min_element(t[i].begin(), t[i].end());
        printf("  NT M=%d N=%d K=%d %-22s median %.1f us = %.0f TF/s   (best %.1f us = %.0f TF/s)\n", M, N, K, vs[i].name.c_str(),
               med * 1e3, fl / med / 1e9, mn * 1e3, fl / mn / 1e9);
    }
    fflush(stdout);
    hipFree(A); hipFree(Bt); hipFree(C); hipFree(C2); hipFree(bias); hipFree(ref); hipFree(rows); hipFree(pre); hipFree(colpart);
}

static void run_tn(int K, int M, int N, int rounds, int reps) {
    bf16_t *A, *B;
    float *C, *ws, *ref;
    int* rows;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&B, (size_t)N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 4));
    int splits = 0;
    const long wsn = mgx_gemm_tn_workspace_floats(M, N, K, &splits);
    CK(hipMalloc(&ws, (size_t)wsn * 4));
    const int NR = 32;
    CK(hipMalloc(&ref, (size_t)NR * N * 4));
    CK(hipMalloc(&rows, NR * 4));
    fill_kernel<<<1024, 256>>>(A, (long)M * K, 3u);
    fill_kernel<<<1024, 256>>>(B, (long)N * K, 4u);
    std::vector<int> hr(NR);
    for (int i = 0; i < NR; ++i) hr[i] = (int)(((long)i * 7919 + (i % 3) * 255) % M);
    hr[0] = 0; hr[1] = M - 1;
    CK(hipMemcpy(rows, hr.data(), NR * 4, hipMemcpyHostToDevice));
    ref_tn_kernel<<<dim3((N + 255) / 256, NR), 256>>>(A, B, rows, ref, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)NR * N), hc(N);
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
    int rc = mgx_gemm_tn_bf16(A, B, C, ws, M, N, K, M, N, 0, nullptr);
    if (rc) { printf("  tn rc=%d %s\n", rc, mgx_last_error_string()); return; }
    CK(hipDeviceSynchronize());
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < NR; ++r) {
        CK(hipMemcpy(hc.data(), C + (size_t)hr[r] * N, (size_t)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            maxerr = std::max(maxerr, (double)std::fabs(href[(size_t)r * N + n] - hc[n]));
            maxref = std::max(maxref, (double)std::fabs(href[(size_t)r * N + n]));
        }
    }
    printf("  check tn max|err|=%.4g (max|ref|=%.4g) %s  splits=%d\n", maxerr, maxref, maxerr <= 1e-3 * maxref ? "ok" : "MISMATCH", splits);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<double> t;
    for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; ++k) mgx_gemm_tn_bf16(A, B, C, ws, M, N, K, M, N, 0, nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) t.push_back(ms / reps);
    }
    const double fl = 2.0 * M * N * K, med = median(t);
    printf("  TN K=%d M=%d N=%d median %.1f us = %.0f TF/s (incl. split-K reduce)\n", K, M, N, med * 1e3, fl / med / 1e9);
    fflush(stdout);
    hipFree(A); hipFree(B); hipFree(C); hipFree(ws); hipFree(ref); hipFree(rows);
}

// two weight gradients: two launches against the paired launch
static void run_tn_pair(int K, int M0, int N0, int M1, int N1, int rounds, int reps) {
    bf16_t *A0, *B0, *A1, *B1;
    float *C0, *C1, *ws0, *ws1, *wsp;
    CK(hipMalloc(&A0, (size_t)M0 * K * 2)); CK(hipMalloc(&B0, (size_t)N0 * K * 2));
    CK(hipMalloc(&A1, (size_t)M1 * K * 2)); CK(hipMalloc(&B1, (size_t)N1 * K * 2));
    CK(hipMalloc(&C0, (size_t)M0 * N0 * 4)); CK(hipMalloc(&C1, (size_t)M1 * N1 * 4));
    CK(hipMalloc(&ws0, (size_t)mgx_gemm_tn_workspace_floats(M0, N0, K, nullptr) * 4));
    CK(hipMalloc(&ws1, (size_t)mgx_gemm_tn_workspace_floats(M1, N1, K, nullptr) * 4));
    CK(hipMalloc(&wsp, (size_t)mgx_gemm_tn_pair_workspace_floats(M0, N0, M1, N1, K) * 4));
    fill_kernel<<<1024, 256>>>(A0, (long)M0 * K, 3u); fill_kernel<<<1024, 256>>>(B0, (long)N0 * K, 4u);
    fill_kernel<<<1024, 256>>>(A1, (long)M1 * K, 5u); fill_kernel<<<1024, 256>>>(B1, (long)N1 * K, 6u);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> t[2];
    for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < reps; ++k) {
                if (v == 0) {
                    mgx_gemm_tn_bf16(A0, B0, C0, ws0, M0, N0, K, M0, N0, 0, nullptr);
                    mgx_gemm_tn_bf16(A1, B1, C1, ws1, M1, N1, K, M1, N1, 0, nullptr);
                } else {
                    mgx_gemm_tn_pair_bf16(A0, B0, C0, M0, N0, M0, N0, A1, B1, C1, M1, N1, M1, N1, K, wsp, 0, nullptr);
                }
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) t[v].push_back(ms / reps);
        }
    const double fl = 2.0 * K * ((double)M0 * N0 + (double)M1 * N1);
    printf("  TN pair K=%d [%dx%d]+[%dx%d]: two launches %.1f us = %.0f TF/s, paired %.1f us = %.0f TF/s\n", K, M0, N0, M1, N1,
           median(t[0]) * 1e3, fl / median(t[0]) / 1e9, median(t[1]) * 1e3, fl / median(t[1]) / 1e9);
    fflush(stdout);
    hipFree(A0); hipFree(B0); hipFree(A1); hipFree(B1); hipFree(C0); hipFree(C1); hipFree(ws0); hipFree(ws1); hipFree(wsp);
}

static void run_mfma() {
    bf16_t* src;
    float* out;
    CK(hipMalloc(&src, 65536 * 2 + 64));
    CK(hipMalloc(&out, 256 * 512 * 4 * 2));
    fill_kernel<<<64, 256>>>(src, 65536, 9u);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int shape : {16, 32}) {
        const int iters = 20000;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            if (shape == 16) mfma_only_kernel<16><<<256, 512>>>(src, out, iters);
            else mfma_only_kernel<32><<<256, 512>>>(src, out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = 256.0 * 8 * iters * (shape == 16 ? 16 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
            printf("  mfma-only %s: %.2f ms = %.0f TF/s\n", shape == 16 ? "16x16x32" : "32x32x16", ms, fl / ms / 1e9);
        }
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "nt";
    const int rounds = 7, reps = 10;
    if (mode == "mfma") { run_mfma(); return 0; }
    if (mode == "nt") {
        if (argc >= 5) { run_nt(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), rounds, reps); return 0; }
        const int shapes[][3] = {{32768, 3072, 1024}, {32768, 1024, 1024}, {32768, 4096, 1024}, {32768, 1024, 4096},
                                 {32768, 1024, 3072}, {32768, 640, 1024}, {32768, 1024, 640}, {4096, 4096, 4096},
                                 {1000, 264, 192}};
        for (auto& s : shapes) run_nt(s[0], s[1], s[2], rounds, reps);
        return 0;
    }
    if (mode == "tn") {
        if (argc >= 5) { run_tn(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), rounds, reps); return 0; }
        const int shapes[][3] = {{32768, 1024, 4096}, {32768, 4096, 1024}, {32768, 1024, 1024}, {32768, 1024, 3072},
                                 {32768, 640, 1024}};
        for (auto& s : shapes) run_tn(s[0], s[1], s[2], rounds, reps);
        return 0;
    }
    if (mode == "tnpair") {
        run_tn_pair(32768, 1024, 4096, 4096, 1024, rounds, reps);
        run_tn_pair(32768, 1024, 1024, 1024, 3072, rounds, reps);
        run_tn_pair(8192, 1024, 4096, 4096, 1024, rounds, reps);
        run_tn_pair(8192, 1024, 1024, 1024, 3072, rounds, reps);
        return 0;
    }
    fprintf(stderr, "usage: gemm_lab nt|tn|tnpair|mfma [dims]\n");
    return 1;
}
