// Decode projections: C[M,N] = A[M,K] . Wt[N,K]^T with M <= 256 token rows (batch x 3 rows per sampling step,
// musicgeneration_amd/decode.py) -- the x@W of Conv1d.forward / nn.Linear (base_layers.py:54-57,
// alibi_unified.py:90,92) at decode time, with the rest of the block's elementwise tail folded in:
// bias, GELU/ReLU (alibi_unified.py:90), or residual + LayerNorm (alibi_unified.py:88-89,92-94).
//
// A library GEMM at M = 192 is a latency chain (~9-15 us for 2-8 MB of weights) followed by separate elementwise
// launches.  Here the work is cut (N/64) x (M/48 or M/64) x SPLITS ways into >= ~128 workgroups:
//   skinny_partial_kernel   grid (N/64, SPLITS, row tiles), 4 waves; 48 or 64 rows x 64 columns x one K slice (up to 1024
//                           deep) per workgroup on v_mfma_f32_16x16x32_bf16.  SPLITS == 1 (the wide projections
//                           qkv and linear1): bias / activation applied in place, bf16 result, no second launch.
//                           SPLITS > 1 (the d x d and ff x d projections): fp32 partial tile -> slab [SPLITS][M][N]
//   skinny_reduce[_ln]_kernel  sum of the slabs + bias, then the epilogue of the site (elementwise: one thread per
//                           8 columns; LayerNorm: one wave per token row)
// Slabs are what a split costs (written, then read back dirty: ~6 us per 12 MB measured), so the slice is as deep
// as LDS allows before K is split.  Plain stream-ordered launches (no in-launch hand-off).
#include "mgx_common.h"

namespace {

constexpr int SK_BN = 64;      // column tile
constexpr int SK_BM = 64;      // largest row tile (A slice in LDS: 64 rows x 1024 k x 2 B = 128 KiB at the deepest slice)
constexpr int SK_TS = 68;      // floats per row of the output tile in LDS
#ifndef SK_MIN_WG
#define SK_MIN_WG 128
#endif

// A-slice image: [k block of 64][row][64 k] = 128-byte rows, 16-byte chunk index XORed with (row >> 1) & 7
// (gemm.hip's NT image: full 128-byte lines per LDS-DMA piece, conflict-free ds_read_b128 for the 16x16x32 operand)
__device__ __forceinline__ int sk_img(int row, int ch) {
    return 1024 * (row >> 3) + 128 * (row & 7) + 16 * (ch ^ ((row >> 1) & 7));
}

__device__ __forceinline__ float sk_gelu(float x) {          // exact-erf GELU via Abramowitz-Stegun 7.1.26 (see elementwise.hip)
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    return 0.5f * x * (1.f + copysignf(fmaf(-poly * t, e, 1.f), z));
}

// grid (N/64, SPLITS, row tiles), 4 waves.  The workgroup's activation slice A[m0 : m0 + 16 NRB, kbeg : kbeg + 32 KS]
// (shared by the four waves) is staged once by LDS-DMA; each wave owns 16 output columns and issues every load of
// its weight rows for the whole slice up front (KS x 16 B per lane), so the workgroup pays ONE memory round trip;
// then KS k steps of {NRB row-block fragments from LDS, MFMA}.  Rows past M are clamped copies whose results are
// dropped.  NRB and KS are compile-time so the accumulators and the weight fragments stay in registers.
// mode < 0: fp32 tile -> slab[split]; mode 0 / 1 / 2 (single split): bias / +GELU / +ReLU -> bf16 out.
template <int NRB, int KS>
__global__ __launch_bounds__(256) void skinny_partial_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              float* __restrict__ slab, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ outp, int M, int N, int K, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int MPAD = NRB * 16, KC = KS * 32, NKB = KC / 64, GROUPS = MPAD / 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int n0 = blockIdx.x * SK_BN + 16 * w;          // this wave's 16 columns
    const int split = blockIdx.y;
    const int m0 = blockIdx.z * MPAD;
    const int kbeg = split * KC;
    // weights: row n0 + l15, k = kbeg + 32*ks + 8*g4; activations: pieces of 8 rows x 128 B, piece p = (k block,
    // 8-row group), round-robin over the waves.  Everything is in flight at once; with a deep slice the requests
    // are issued half by half (k blocks [0, NKB/2) first), so the MFMAs of the first half run while the second
    // half is still landing.
    constexpr bool TWO = KS >= 8 && NKB % 2 == 0 && ((NKB / 2) * GROUPS) % 4 == 0;
    constexpr int HALF_P = (NKB / 2) * GROUPS;            // pieces of the first half
    // LDS-DMA requests per wave in the second half.  Waiting until only that many requests are outstanding covers
    // every first-half piece wherever the compiler puts the (register) weight loads: the DMA requests keep their
    // order among themselves and the second-half pieces are issued last.
    constexpr int C2 = (NKB * GROUPS - HALF_P) / 4;
    bf16x8 bf[KS];
    const bf16_t* bp = W + (long)(n0 + l15) * K + kbeg + 8 * g4;
    auto stage = [&](int p) {
        const int kb = p / GROUPS, gr = p - kb * GROUPS;
        const int row = 8 * gr + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = A + (long)min(m0 + row, M - 1) * K + kbeg + 64 * kb + 8 * ch;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void*)(smem + (kb * MPAD + 8 * gr) * 128), 16, 0, 0);
    };
    f32x4 acc[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfma_steps = [&](int ks0, int ks1) {
#pragma unroll
        for (int ks = ks0; ks < ks1; ++ks) {
            const char* base = smem + ((ks >> 1) * MPAD) * 128 + sk_img(l15, 4 * (ks & 1) + g4);
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(base + 2048 * i);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks], af, acc[i], 0, 0, 0);
            }
        }
    };
    if constexpr (TWO) {
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) bf[ks] = *reinterpret_cast<const bf16x8*>(bp + 32 * ks);
        for (int p = w; p < HALF_P; p += 4) stage(p);
#pragma unroll
        for (int ks = KS / 2; ks < KS; ++ks) bf[ks] = *reinterpret_cast<const bf16x8*>(bp + 32 * ks);
        for (int p = HALF_P + w; p < NKB * GROUPS; p += 4) stage(p);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C2) : "memory");
        __syncthreads();
        mfma_steps(0, KS / 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfma_steps(KS / 2, KS);
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bf[ks] = *reinterpret_cast<const bf16x8*>(bp + 32 * ks);
        for (int p = w; p < NKB * GROUPS; p += 4) stage(p);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfma_steps(0, KS);
    }
    // D[row = n][col = m]: the lane holds C row m0 + 16*i + l15, columns n0 + 4*g4 .. +3.  The tile goes through LDS
    // (rows of 64 + 4 floats: conflict-free both ways) and leaves as whole row segments.
    float* tile = reinterpret_cast<float*>(smem);
    __syncthreads();                                       // every wave is done with the A image
#pragma unroll
    for (int i = 0; i < NRB; ++i)
        *reinterpret_cast<f32x4*>(tile + (16 * i + l15) * SK_TS + 16 * w + 4 * g4) = acc[i];
    __syncthreads();
    const int c4 = 4 * (threadIdx.x & 15);
    const int col = blockIdx.x * SK_BN + c4;
    const int rows = min(MPAD, M - m0);
    if (mode < 0) {
        float* out = slab + (long)split * M * N + (long)m0 * N + col;
        for (int r = threadIdx.x >> 4; r < rows; r += 16)
            *reinterpret_cast<f32x4*>(out + (long)r * N) = *reinterpret_cast<const f32x4*>(tile + r * SK_TS + c4);
        return;
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + col);
    bf16_t* out = outp + (long)m0 * N + col;
    for (int r = threadIdx.x >> 4; r < rows; r += 16) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * SK_TS + c4) + bv;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e];
            if (mode == 1) x = sk_gelu((float)(bf16_t)x);    // the activation sees the bf16 pre-activation, as the
            if (mode == 2) x = fmaxf((float)(bf16_t)x, 0.f); // unfused path (bias_act_fwd on a bf16 GEMM result) does
            o[e] = (bf16_t)x;
        }
        *reinterpret_cast<bf16x4*>(out + (long)r * N) = o;
    }
}

// sum over the split slabs of 8 consecutive columns, four slabs per round so that 8 loads are in flight per lane
// (a plain running sum over a run-time slab count compiled to one dependent load per slab: a latency chain)
__device__ __forceinline__ void sk_slab_sum(const float* __restrict__ p, long MN, int splits, f32x4& s0, f32x4& s1) {
    s0 = f32x4{0.f, 0.f, 0.f, 0.f};
    s1 = s0;
    for (int k0 = 0; k0 < splits; k0 += 4) {
        f32x4 t[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* q = p + (long)min(k0 + j, splits - 1) * MN;
            t[j][0] = *reinterpret_cast<const f32x4*>(q);
            t[j][1] = *reinterpret_cast<const f32x4*>(q + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + j < splits) {
                s0 += t[j][0];
                s1 += t[j][1];
            }
        }
    }
}

// MODE 0: out = sum + bias; 1: gelu; 2: relu -- one thread per 8 columns over the whole [M,N] result
template <int MODE>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ out, int M, int N, int splits) {
    const long MN = (long)M * N;
    const long idx = 8 * ((long)blockIdx.x * 256 + threadIdx.x);
    if (idx >= MN) return;
    const int col = (int)(idx % N);
    f32x4 s0, s1;
    sk_slab_sum(slab + idx, MN, splits, s0, s1);
    s0 += *reinterpret_cast<const f32x4*>(bias + col);
    s1 += *reinterpret_cast<const f32x4*>(bias + col + 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = e < 4 ? s0[e & 3] : s1[e & 3];
        if (MODE == 1) x = sk_gelu((float)(bf16_t)x);       // the activation sees the bf16 pre-activation, as the
        if (MODE == 2) x = fmaxf((float)(bf16_t)x, 0.f);    // unfused path (bias_act_fwd on a bf16 GEMM result) does
        o[e] = (bf16_t)x;
    }
    *reinterpret_cast<bf16x8*>(out + idx) = o;
}

// out = LayerNorm(sum + bias + res) * gamma + beta: one wave (= one workgroup) per token row, N <= 64 * 8 * NV
template <int NV>
__global__ __launch_bounds__(64) void skinny_reduce_ln_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                               const bf16_t* __restrict__ res, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               bf16_t* __restrict__ out, int M, int N, int splits) {
    const int lane = threadIdx.x;
    const int row = blockIdx.x;
    const long MN = (long)M * N;
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < N) {
            f32x4 s0, s1;
            sk_slab_sum(slab + (long)row * N + col, MN, splits, s0, s1);
            s0 += *reinterpret_cast<const f32x4*>(bias + col);
            s1 += *reinterpret_cast<const f32x4*>(bias + col + 4);
            const bf16x8 r = *reinterpret_cast<const bf16x8*>(res + (long)row * N + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // as the unfused path: the projection (+ bias) is rounded to bf16 once (the GEMM's output type), then the
                // residual is added and the statistics are taken in fp32 (ln_fwd_kernel)
                v[c][e] = (float)(bf16_t)(e < 4 ? s0[e & 3] : s1[e & 3]) + (float)r[e];
                sum += v[c][e];
            }
        }
    }
    const float mu = wave_sum(sum) / (float)N;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[c][e] - mu;
                sq = fmaf(t, t, sq);
            }
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)N + eps);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col < N) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + col), g1 = *reinterpret_cast<const f32x4*>(gamma + col + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + col), b1 = *reinterpret_cast<const f32x4*>(beta + col + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = (bf16_t)((v[c][e] - mu) * rs * (e < 4 ? g0[e & 3] : g1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]));
            *reinterpret_cast<bf16x8*>(out + (long)row * N + col) = o;
        }
    }
}

// rows per workgroup tile (16-row blocks) and the number of row tiles
void skinny_rows(int M, int* nrb, int* mtiles) {
    if (M <= SK_BM) {
        *nrb = (M + 15) / 16;
        *mtiles = 1;
        return;
    }
    // 64- or 48-row tiles, whichever pads M less; 48 on a tie (more, lighter workgroups: at M = 192 the load
    // phase per workgroup shrinks from 256 to 224 KB and 192-256 instead of 144-192 workgroups are in flight:
    // -1 us per projection)
    const int t4 = (M + 63) / 64, t3 = (M + 47) / 48;
    *nrb = t3 * 48 <= t4 * 64 ? 3 : 4;
    *mtiles = *nrb == 3 ? t3 : t4;
}

int skinny_splits(int M, int N, int K) {
    // the deepest slice (1024 ... 64, the instantiated depths) that divides K and still gives >= 128 workgroups
    int nrb, mtiles;
    skinny_rows(M, &nrb, &mtiles);
    const int tiles = (N / SK_BN) * mtiles;
    int kc = 64;
    for (int c = 1024; c >= 64; c /= 2)
        if (K % c == 0) { kc = c; break; }
    while (kc > 64 && tiles * (K / kc) < SK_MIN_WG) kc /= 2;
    return K / kc;
}

}  // namespace

extern "C" long mgx_skinny_gemm_workspace_floats(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % SK_BN || K % 64) return 0;
    return (long)skinny_splits(M, N, K) * M * N;
}

extern "C" int mgx_skinny_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* res,
                                    const float* gamma, const float* beta, float eps, void* out, float* workspace,
                                    int M, int N, int K, int mode, void* stream) {
    MGX_REQUIRE(A && Wt && bias && out && workspace, MGX_EINVAL, "mgx_skinny_gemm_bf16: null pointer");
    MGX_REQUIRE(M > 0 && M <= 256, MGX_EUNSUPPORTED, "mgx_skinny_gemm_bf16: M=%d (1..256 token rows)", M);
    MGX_REQUIRE(N > 0 && N % SK_BN == 0 && K > 0 && K % 64 == 0, MGX_EUNSUPPORTED,
                "mgx_skinny_gemm_bf16: N=%d must be a multiple of %d, K=%d of 64", N, SK_BN, K);
    MGX_REQUIRE(mode >= 0 && mode <= 3, MGX_EINVAL, "mgx_skinny_gemm_bf16: bad mode %d", mode);
    MGX_REQUIRE(mode != 3 || (res && gamma && beta && N <= 64 * 8 * 4), MGX_EINVAL,
                "mgx_skinny_gemm_bf16: LayerNorm mode needs res/gamma/beta and N <= 2048 (N=%d)", N);
    const int splits = skinny_splits(M, N, K);
    const int kc = K / splits;
    hipStream_t st = (hipStream_t)stream;
    int NRB, mtiles;
    skinny_rows(M, &NRB, &mtiles);
    const bool direct = splits == 1 && mode != 3;          // epilogue in the GEMM launch, no slab
    const int pmode = direct ? mode : -1;
    const size_t lds = std::max((size_t)NRB * 16 * kc * 2, (size_t)NRB * 16 * SK_TS * 4);   // A image, then the output tile
    const dim3 pgrid(N / SK_BN, splits, mtiles);
#define MGX_SK_PART(R, S)                                                                                          \
    hipLaunchKernelGGL((skinny_partial_kernel<R, S>), pgrid, dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)Wt, \
                       workspace, bias, (bf16_t*)out, M, N, K, pmode)
#define MGX_SK_ROWS(S)                                                                                             \
    do {                                                                                                           \
        if (NRB == 1) MGX_SK_PART(1, S);                                                                           \
        else if (NRB == 2) MGX_SK_PART(2, S);                                                                      \
        else if (NRB == 3) MGX_SK_PART(3, S);                                                                      \
        else MGX_SK_PART(4, S);                                                                                    \
    } while (0)
    if (kc == 64) MGX_SK_ROWS(2);
    else if (kc == 128) MGX_SK_ROWS(4);
    else if (kc == 256) MGX_SK_ROWS(8);
    else if (kc == 512) MGX_SK_ROWS(16);
    else if (kc == 1024) MGX_SK_ROWS(32);
    else MGX_REQUIRE(false, MGX_EUNSUPPORTED, "mgx_skinny_gemm_bf16: K=%d has no slice depth of 64 ... 1024", K);
#undef MGX_SK_ROWS
#undef MGX_SK_PART
    MGX_CHECK_LAUNCH("skinny_partial");
    if (direct) return MGX_OK;
    const long vecs = (long)M * N / 8;
    const dim3 egrid((unsigned)((vecs + 255) / 256));
#define MGX_SK_RED(MODE)                                                                                           \
    hipLaunchKernelGGL((skinny_reduce_kernel<MODE>), egrid, dim3(256), 0, st, workspace, bias, (bf16_t*)out, M, N, splits)
#define MGX_SK_LN(NV)                                                                                              \
    hipLaunchKernelGGL((skinny_reduce_ln_kernel<NV>), dim3(M), dim3(64), 0, st, workspace, bias, (const bf16_t*)res, \
                       gamma, beta, eps, (bf16_t*)out, M, N, splits)
    if (mode == 0) MGX_SK_RED(0);
    else if (mode == 1) MGX_SK_RED(1);
    else if (mode == 2) MGX_SK_RED(2);
    else if (N <= 512) MGX_SK_LN(1);
    else if (N <= 1024) MGX_SK_LN(2);
    else MGX_SK_LN(4);
#undef MGX_SK_RED
#undef MGX_SK_LN
    MGX_CHECK_LAUNCH("skinny_reduce");
    return MGX_OK;
}
