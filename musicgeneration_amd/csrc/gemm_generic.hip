// Generic GEMM of libmgx: every projection the tuned bf16 kernels (gemm_nt.hip, gemm_tn.hip, skinny.hip) do not
// take -- the fp32 "parity mode" of the whole model (north_star: 1e-5 on logits, bit-exact greedy decode) and bf16
// shapes off the 64-element grid (head widths 96, vocabularies 618 / 92 / 130 / 203 unpadded, the sampler's prefill at
// odd widths).  Replaces Conv1d.forward (base_layers.py:54-57: x @ w + b, w [in, out]), nn.Linear
// (alibi_unified.py:90,92,137: x @ W^T + b, W [out, in]) and the GEMMs autograd derives from them (dy @ W, dy^T @ x),
// in either operand orientation, with any sizes and row strides.
//
//   C[M,N] = opA(A)[M,K] . opB(B)[K,N] (+ bias[N]) (+ res[M,N])
//
// Storage fp32 or bf16 (inputs and output independently), products and sums in fp32 on v_mfma_f32_32x32x2_f32: the
// operands are converted to fp32 on their way into LDS, so a bf16 input is exact and an fp32 input is used unrounded.
// Workgroup = 4 waves, tile 128 x 128 x 16; each wave owns 64 x 64 = 2 x 2 MFMA blocks.  LDS images are k-major
// (As[k][m], Bs[k][n], rows padded by 4 floats): the MFMA operand of a lane (one m or n, one k) is a conflict-free
// 32-float row segment.  Every global access is guarded (zero fill / no store), so no shape needs padding.  K is
// walked in order, one launch, no split: results are deterministic.  This is the correctness path: ~25-40 % of the
// fp32 MFMA rate, not a tuned kernel.
#include "mgx_common.h"

namespace {

constexpr int GB = 128, GK = 16, GLD = GB + 4;

template <typename T>
__device__ __forceinline__ float g_load(const T* p, long i) { return to_f32(p[i]); }

struct GenArgs {
    const void *A, *B, *res;
    void* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc, ldr, transA, transB;
};

template <typename TA, typename TB, typename TC, typename TR>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GenArgs g) {
    __shared__ float As[GK][GLD];
    __shared__ float Bs[GK][GLD];
    const TA* A = reinterpret_cast<const TA*>(g.A);
    const TB* B = reinterpret_cast<const TB*>(g.B);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < g.K; k0 += GK) {
        // stage: 2048 elements of each operand, 8 per thread; the fastest thread index follows the operand's
        // contiguous dimension
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            int m, k;
            if (g.transA) { m = idx & (GB - 1); k = idx >> 7; } else { k = idx & (GK - 1); m = idx >> 4; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = g.transA ? g_load(A, (long)gk * g.lda + gm) : g_load(A, (long)gm * g.lda + gk);
            As[k][m] = v;
            int n, kb;
            if (g.transB) { kb = idx & (GK - 1); n = idx >> 4; } else { n = idx & (GB - 1); kb = idx >> 7; }
            const int gn = n0 + n, gkb = k0 + kb;
            float u = 0.f;
            if (gn < g.N && gkb < g.K) u = g.transB ? g_load(B, (long)gn * g.ldb + gkb) : g_load(B, (long)gkb * g.ldb + gn);
            Bs[kb][n] = u;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const int k = 2 * kk + h5;
            const float a0 = As[k][wm + l31], a1 = As[k][wm + 32 + l31];
            const float b0 = Bs[k][wn + l31], b1 = Bs[k][wn + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    TC* C = reinterpret_cast<TC*>(g.C);
    const TR* R = reinterpret_cast<const TR*>(g.res);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + 32 * j + l31;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * h5 + (r & 3);
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (R) v += to_f32(R[(long)row * g.ldr + col]);
                    C[(long)row * g.ldc + col] = from_f32<TC>(v);
                }
            }
    }
}

template <typename TA, typename TB, typename TC>
void launch_generic(const GenArgs& g, int res_dtype, hipStream_t st) {
    const dim3 grid((g.N + GB - 1) / GB, (g.M + GB - 1) / GB);
    if (res_dtype == MGX_BF16)
        hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, TC, bf16_t>), grid, dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_generic_kernel<TA, TB, TC, float>), grid, dim3(256), 0, st, g);
}

}  // namespace

extern "C" int mgx_gemm_generic(const void* A, const void* B, void* C, const float* bias, const void* res, int M, int N,
                                int K, int lda, int ldb, int ldc, int ldr, int transA, int transB, int a_dtype,
                                int b_dtype, int c_dtype, int res_dtype, void* stream) {
    MGX_REQUIRE(A && B && C, MGX_EINVAL, "mgx_gemm_generic: null pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0, MGX_EINVAL, "mgx_gemm_generic: M=%d N=%d K=%d", M, N, K);
    MGX_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N && (!res || ldr >= N), MGX_EINVAL,
                "mgx_gemm_generic: a leading dimension is shorter than its row (lda=%d ldb=%d ldc=%d ldr=%d)", lda, ldb, ldc, ldr);
    for (int dt : {a_dtype, b_dtype, c_dtype, res_dtype})
        MGX_REQUIRE(dt == MGX_F32 || dt == MGX_BF16, MGX_EINVAL, "mgx_gemm_generic: dtype %d", dt);
    MGX_REQUIRE((long)((M + GB - 1) / GB) < 65536, MGX_EUNSUPPORTED, "mgx_gemm_generic: M=%d too large for one launch", M);
    GenArgs g{A, B, res, C, bias, M, N, K, lda, ldb, ldc, ldr, transA != 0, transB != 0};
    hipStream_t st = (hipStream_t)stream;
    const int key = (a_dtype << 2) | (b_dtype << 1) | c_dtype;
    switch (key) {
        case 0: launch_generic<float, float, float>(g, res_dtype, st); break;
        case 1: launch_generic<float, float, bf16_t>(g, res_dtype, st); break;
        case 2: launch_generic<float, bf16_t, float>(g, res_dtype, st); break;
        case 3: launch_generic<float, bf16_t, bf16_t>(g, res_dtype, st); break;
        case 4: launch_generic<bf16_t, float, float>(g, res_dtype, st); break;
        case 5: launch_generic<bf16_t, float, bf16_t>(g, res_dtype, st); break;
        case 6: launch_generic<bf16_t, bf16_t, float>(g, res_dtype, st); break;
        default: launch_generic<bf16_t, bf16_t, bf16_t>(g, res_dtype, st); break;
    }
    MGX_CHECK_LAUNCH("gemm_generic");
    return MGX_OK;
}
