// HBM-bound fused row kernels: (dropout +) residual + LayerNorm forward/backward,
// bias + GELU/ReLU (+ dropout) forward/backward, fused AdamW.
// Replaces alibi_unified.py:88-94 (drop/add/norm, activation/drop) and
// base_layers.py:36-41 (final Norm); AdamW = build_everything_unified.py:42.
//
// Layout: activations [T, d] row-major in fp32 or bf16; one wave per row, 16-byte
// vector accesses (8 bf16 or 4 fp32 per lane per step), row held in registers
// between the statistics pass and the normalise pass so each element is read once.
#include "mgx_common.h"
#include <mutex>

namespace {

constexpr int PARTIAL_ROWS = 512;  // row-blocks of the two-stage column reductions

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4 type;
};
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    typedef bf16x8 type;
};

template <typename T, int N>
__device__ __forceinline__ void load_vec(const T* p, float (&o)[N]) {
    typename Vec<T>::type v = *reinterpret_cast<const typename Vec<T>::type*>(p);
#pragma unroll
    for (int k = 0; k < N; ++k) o[k] = to_f32(v[k]);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const float (&o)[N]) {
    typename Vec<T>::type v;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = from_f32<T>(o[k]);
    *reinterpret_cast<typename Vec<T>::type*>(p) = v;
}

// ---------------------------------------------------------------------------
// LayerNorm forward.  One wave per row; NV vectors of Vec<T>::N per lane.
// ---------------------------------------------------------------------------
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ a, const T* __restrict__ res,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      T* __restrict__ y, T* __restrict__ z, float* __restrict__ mean,
                                                      float* __restrict__ rstd, long T_rows, int d, float eps,
                                                      float p_drop, unsigned long long seed) {
    constexpr int N = Vec<T>::N;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= T_rows) return;
    const unsigned thr16 = mgx_drop_thr16(p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
    float v[NV][N];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * N;
        if (col < d) {
            load_vec<T, N>(a + row * d + col, v[c]);
            if (p_drop > 0.f) {
#pragma unroll
                for (int k = 0; k < N; k += 2) {
                    bool k0, k1;
                    mgx_keep_pair(seed, (unsigned long long)(row * d + col + k), thr16, k0, k1);
                    v[c][k] = k0 ? v[c][k] * keep_scale : 0.f;
                    v[c][k + 1] = k1 ? v[c][k + 1] * keep_scale : 0.f;
                }
            }
            if (res) {
                float r[N];
                load_vec<T, N>(res + row * d + col, r);
#pragma unroll
                for (int k = 0; k < N; ++k) v[c][k] += r[k];
            }
            if (z) {
                store_vec<T, N>(z + row * d + col, v[c]);
                // statistics are taken on the value the backward will re-read
#pragma unroll
                for (int k = 0; k < N; ++k) v[c][k] = to_f32(from_f32<T>(v[c][k]));
            }
#pragma unroll
            for (int k = 0; k < N; ++k) sum += v[c][k];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[c][k] = 0.f;
        }
    }
    const float mu = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * N;
        if (col < d) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float t = v[c][k] - mu;
                sq = fmaf(t, t, sq);
            }
        }
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)d + eps);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * N;
        if (col < d) {
            float o[N];
#pragma unroll
            for (int k = 0; k < N; ++k) o[k] = (v[c][k] - mu) * rs * gamma[col + k] + beta[col + k];
            store_vec<T, N>(y + row * d + col, o);
        }
    }
    if (lane == 0 && mean) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// ---------------------------------------------------------------------------
// LayerNorm backward.  One wave per row for dz; per-block partial sums of
// dgamma/dbeta in LDS -> partial[blk][2][d]; a second kernel reduces the blocks.
// ---------------------------------------------------------------------------
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                      const T* __restrict__ z, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      T* __restrict__ dz, T* __restrict__ da,
                                                      float* __restrict__ partial, long T_rows, int d, float p_drop,
                                                      unsigned long long seed, int want_dsum) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][d], one quantity at a time
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned thr16 = mgx_drop_thr16(p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
    float g_acc[NV][N], b_acc[NV][N], a_acc[NV][N];   // dgamma, dbeta, column sum of da (the bias gradient of the branch)
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int k = 0; k < N; ++k) g_acc[c][k] = b_acc[c][k] = a_acc[c][k] = 0.f;

    // The next row's vectors (raw, 16 bytes each) are requested before the current row's arithmetic and its two
    // wave reductions: one row at a time left the wave idle for a memory round trip per row (4.2 TB/s).
    typedef typename Vec<T>::type RawT;
    float gm[NV][N];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 64 + lane) * N;
#pragma unroll
        for (int k = 0; k < N; ++k) gm[c][k] = col < d ? gamma[col + k] : 0.f;
    }
    RawT r_dy[NV], r_dy2[NV], r_z[NV];
    float r_mu = 0.f, r_rs = 0.f;
    auto request = [&](long row) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = (c * 64 + lane) * N;
            if (col < d) {
                r_dy[c] = *reinterpret_cast<const RawT*>(dy + row * d + col);
                if (dy2) r_dy2[c] = *reinterpret_cast<const RawT*>(dy2 + row * d + col);
                r_z[c] = *reinterpret_cast<const RawT*>(z + row * d + col);
            }
        }
        r_mu = mean[row];
        r_rs = rstd[row];
    };
    const long stride = (long)gridDim.x * 4;
    long row = (long)blockIdx.x * 4 + w;
    if (row < T_rows) request(row);
    while (row < T_rows) {
        const float mu = r_mu, rs = r_rs;
        float g[NV][N], xh[NV][N];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = (c * 64 + lane) * N;
            if (col < d) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    g[c][k] = to_f32(r_dy[c][k]);
                    if (dy2) g[c][k] += to_f32(r_dy2[c][k]);
                    xh[c][k] = (to_f32(r_z[c][k]) - mu) * rs;
                }
            }
        }
        const long next = row + stride;
        if (next < T_rows) request(next);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = (c * 64 + lane) * N;
            if (col < d) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    b_acc[c][k] += g[c][k];
                    g_acc[c][k] = fmaf(g[c][k], xh[c][k], g_acc[c][k]);
                    g[c][k] *= gm[c][k];                // dxhat
                    s1 += g[c][k];
                    s2 = fmaf(g[c][k], xh[c][k], s2);
                }
            }
        }
        s1 = wave_sum(s1) / (float)d;
        s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = (c * 64 + lane) * N;
            if (col < d) {
                float o[N];
#pragma unroll
                for (int k = 0; k < N; ++k) o[k] = (g[c][k] - s1 - xh[c][k] * s2) * rs;
                store_vec<T, N>(dz + row * d + col, o);
                if (da && (da != dz || p_drop > 0.f)) {
                    if (p_drop > 0.f) {
#pragma unroll
                        for (int k = 0; k < N; k += 2) {
                            bool k0, k1;
                            mgx_keep_pair(seed, (unsigned long long)(row * d + col + k), thr16, k0, k1);
                            o[k] = k0 ? o[k] * keep_scale : 0.f;
                            o[k + 1] = k1 ? o[k + 1] * keep_scale : 0.f;
                        }
                    }
                    store_vec<T, N>(da + row * d + col, o);
                }
                if (want_dsum) {
#pragma unroll
                    for (int k = 0; k < N; ++k) a_acc[c][k] += o[k];
                }
            }
        }
        row = next;
    }
    // block reduction of the 4 waves' column partials, one quantity at a time through [4][d] floats
    const int nq = want_dsum ? 3 : 2;
    float* out = partial + (size_t)blockIdx.x * nq * d;
    float* mine = smem + (size_t)w * d;
    for (int q = 0; q < nq; ++q) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int col = (c * 64 + lane) * N;
            if (col < d) {
#pragma unroll
                for (int k = 0; k < N; ++k) mine[col + k] = q == 0 ? g_acc[c][k] : (q == 1 ? b_acc[c][k] : a_acc[c][k]);
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < d; e += 256) out[q * d + e] = smem[e] + smem[d + e] + smem[2 * d + e] + smem[3 * d + e];
        __syncthreads();
    }
}

// out_q[c] = sum_r partial[r][q][c] for q < nseg (1..3)
// Block = 64 columns x 16 row slices (fixed summation order -> deterministic).
__global__ __launch_bounds__(1024) void column_reduce_kernel(const float* __restrict__ partial,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float* __restrict__ out2, int nblk, int d, int nseg) {
    __shared__ float sh[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int width = nseg * d;
    float s0 = 0.f, s1 = 0.f;
    if (c < width) {
        int r = slice;
        for (; r + 16 < nblk; r += 32) {
            s0 += partial[(size_t)r * width + c];
            s1 += partial[(size_t)(r + 16) * width + c];
        }
        if (r < nblk) s0 += partial[(size_t)r * width + c];
    }
    sh[slice][lane] = s0 + s1;
    __syncthreads();
    if (slice == 0 && c < width) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sh[k][lane];
        if (c < d) out0[c] = s;
        else if (c < 2 * d) out1[c - d] = s;
        else out2[c - 2 * d] = s;
    }
}

// ---------------------------------------------------------------------------
// bias + activation (+ dropout)
// ---------------------------------------------------------------------------
// erf for the bf16 path: Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution)
// on v_rcp/v_exp; the fp32 path keeps the libm erff so fp32 parity stays at rounding level.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-poly * t, e, 1.f);
    return copysignf(r, x);
}
template <bool FAST>
__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == MGX_ACT_RELU) return fmaxf(x, 0.f);
    const float z = x * 0.70710678118654752440f;
    return 0.5f * x * (1.f + (FAST ? erf_fast(z) : erff(z)));
}
// d/dx gelu(x) = Phi(x) + x phi(x); erf(x/sqrt2) and phi(x) share the one exp(-x^2/2)
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);     // exp(-x^2/2)
    const float erfv = copysignf(fmaf(-poly * t, e, 1.f), x);
    return fmaf(x * 0.3989422804014327f, e, fmaf(0.5f, erfv, 0.5f));
}
template <bool FAST>
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == MGX_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    if (FAST) return gelu_grad_fast(x);
    const float z = x * 0.70710678118654752440f;
    const float cdf = 0.5f * (1.f + (FAST ? erf_fast(z) : erff(z)));
    const float pdf = 0.3989422804014327f * (FAST ? __builtin_amdgcn_exp2f(-0.5f * x * x * 1.4426950408889634f)
                                                  : expf(-0.5f * x * x));
    return cdf + x * pdf;
}

// Activation and dropout are compile-time: with a run-time `act` hipcc branched per element
// (one small basic block per value) and the kernels ran VALU/branch-bound at 2.7-3.9 TB/s.
// Block b handles rows b, b+grid, ...; thread t handles the 16-byte column vectors t, t+256, ...;
// U rows are loaded before the first is used.
template <typename T, int ACT, bool DROP>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const T* __restrict__ u, const float* __restrict__ bias,
                                                            T* __restrict__ pre, T* __restrict__ h, long T_rows, int n,
                                                            float p_drop, unsigned long long seed) {
    constexpr int N = Vec<T>::N, U = 4;
    constexpr bool FAST = sizeof(T) == 2;
    const unsigned thr16 = mgx_drop_thr16(p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
    for (int col = threadIdx.x * N; col < n; col += 256 * N) {
        float bv[N];
#pragma unroll
        for (int k = 0; k < N; ++k) bv[k] = bias ? bias[col + k] : 0.f;     // bias == NULL: u already is the pre-activation
        for (long row0 = blockIdx.x; row0 < T_rows; row0 += (long)U * gridDim.x) {
            float v[U][N];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const long row = min(row0 + (long)q * gridDim.x, T_rows - 1);
                load_vec<T, N>(u + row * n + col, v[q]);
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const long row = row0 + (long)q * gridDim.x;
                if (row < T_rows) {
                    const long e = row * n + col;
                    float o[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) v[q][k] += bv[k];
                    if (pre) {
                        store_vec<T, N>(pre + e, v[q]);
#pragma unroll
                        for (int k = 0; k < N; ++k) v[q][k] = to_f32(from_f32<T>(v[q][k]));
                    }
#pragma unroll
                    for (int k = 0; k < N; k += 2) {
                        o[k] = act_fwd<FAST>(v[q][k], ACT);
                        o[k + 1] = act_fwd<FAST>(v[q][k + 1], ACT);
                        if (DROP) {
                            bool k0, k1;
                            mgx_keep_pair(seed, (unsigned long long)(e + k), thr16, k0, k1);
                            o[k] = k0 ? o[k] * keep_scale : 0.f;
                            o[k + 1] = k1 ? o[k + 1] * keep_scale : 0.f;
                        }
                    }
                    store_vec<T, N>(h + e, o);
                }
            }
        }
    }
}

// dpre = dh * mask * act'(pre); partial[blk][n] column sums for dbias.
template <typename T, int ACT, bool DROP>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ pre,
                                                            T* __restrict__ dpre, float* __restrict__ partial,
                                                            long T_rows, int n, float p_drop, unsigned long long seed) {
    constexpr int N = Vec<T>::N, U = 4;
    constexpr bool FAST = sizeof(T) == 2;
    const unsigned thr16 = mgx_drop_thr16(p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
    for (int col = threadIdx.x * N; col < n; col += 256 * N) {
        float acc[N];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] = 0.f;
        // raw vectors of the next U rows are requested before the arithmetic of the current ones
        typedef typename Vec<T>::type RawT;
        RawT rg[U], rx[U];
        auto request = [&](long row0) {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const long row = min(row0 + (long)q * gridDim.x, T_rows - 1);
                rg[q] = *reinterpret_cast<const RawT*>(dh + row * n + col);
                rx[q] = *reinterpret_cast<const RawT*>(pre + row * n + col);
            }
        };
        const long step = (long)U * gridDim.x;
        long row0 = blockIdx.x;
        if (row0 < T_rows) request(row0);
        while (row0 < T_rows) {
            float g[U][N], x[U][N];
#pragma unroll
            for (int q = 0; q < U; ++q)
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    g[q][k] = to_f32(rg[q][k]);
                    x[q][k] = to_f32(rx[q][k]);
                }
            if (row0 + step < T_rows) request(row0 + step);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const long row = row0 + (long)q * gridDim.x;
                if (row < T_rows) {
                    const long e = row * n + col;
#pragma unroll
                    for (int k = 0; k < N; k += 2) {
                        if (DROP) {
                            bool k0, k1;
                            mgx_keep_pair(seed, (unsigned long long)(e + k), thr16, k0, k1);
                            g[q][k] = k0 ? g[q][k] * keep_scale : 0.f;
                            g[q][k + 1] = k1 ? g[q][k + 1] * keep_scale : 0.f;
                        }
                        g[q][k] *= act_grad<FAST>(x[q][k], ACT);
                        g[q][k + 1] *= act_grad<FAST>(x[q][k + 1], ACT);
                    }
                    store_vec<T, N>(dpre + e, g[q]);
#pragma unroll
                    for (int k = 0; k < N; ++k) acc[k] += to_f32(from_f32<T>(g[q][k]));
                }
            }
            row0 += step;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) partial[(size_t)blockIdx.x * n + col + k] = acc[k];
    }
}

template <typename T>
int bias_act_fwd_launch(const void* u, const float* bias, void* pre, void* h, long T_rows, int n, int act,
                        float p_drop, unsigned long long seed, hipStream_t st) {
    const int nblk = (int)(T_rows < 8192 ? T_rows : 8192);
#define MGX_BA_FWD(ACT, DROP)                                                                                     \
    hipLaunchKernelGGL((bias_act_fwd_kernel<T, ACT, DROP>), dim3(nblk), dim3(256), 0, st, (const T*)u, bias, (T*)pre, \
                       (T*)h, T_rows, n, p_drop, seed)
    const bool drop = p_drop > 0.f;
    if (act == MGX_ACT_GELU) {
        if (drop) MGX_BA_FWD(MGX_ACT_GELU, true); else MGX_BA_FWD(MGX_ACT_GELU, false);
    } else {
        if (drop) MGX_BA_FWD(MGX_ACT_RELU, true); else MGX_BA_FWD(MGX_ACT_RELU, false);
    }
#undef MGX_BA_FWD
    MGX_CHECK_LAUNCH("bias_act_fwd");
    return MGX_OK;
}

template <typename T>
int bias_act_bwd_launch(const void* dh, const void* pre, void* dpre, float* partial, long T_rows, int n, int nblk,
                        int act, float p_drop, unsigned long long seed, hipStream_t st) {
#define MGX_BA_BWD(ACT, DROP)                                                                                      \
    hipLaunchKernelGGL((bias_act_bwd_kernel<T, ACT, DROP>), dim3(nblk), dim3(256), 0, st, (const T*)dh, (const T*)pre, \
                       (T*)dpre, partial, T_rows, n, p_drop, seed)
    const bool drop = p_drop > 0.f;
    if (act == MGX_ACT_GELU) {
        if (drop) MGX_BA_BWD(MGX_ACT_GELU, true); else MGX_BA_BWD(MGX_ACT_GELU, false);
    } else {
        if (drop) MGX_BA_BWD(MGX_ACT_RELU, true); else MGX_BA_BWD(MGX_ACT_RELU, false);
    }
#undef MGX_BA_BWD
    MGX_CHECK_LAUNCH("bias_act_bwd");
    return MGX_OK;
}

// ---------------------------------------------------------------------------
// AdamW (decoupled weight decay, torch.optim.AdamW semantics)
// ---------------------------------------------------------------------------
template <typename G>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const G* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v,
                                                     bf16_t* __restrict__ shadow, long n, float lr, float b1, float b2,
                                                     float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gr = to_f32(g[i]) * gscale;
        float w = p[i];
        w -= lr * wd * w;
        const float mm = b1 * m[i] + (1.f - b1) * gr;
        const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        w -= (lr / bc1) * (mm / denom);
        p[i] = w;
        if (shadow) shadow[i] = (bf16_t)w;
    }
}

// Multi-tensor form: one launch updates up to ADAM_MAX tensors (their pointers travel in the kernel
// arguments); block b works on a 4096-element chunk of the tensor whose block range contains b.
constexpr int ADAM_MAX = 32, ADAM_CHUNK = 4096;
struct AdamMulti {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    long n[ADAM_MAX];
    int bstart[ADAM_MAX + 1];
    int count;
};
__device__ __forceinline__ float adamw_one(float w, float gr, float& m, float& v, float lr, float b1, float b2,
                                            float eps, float wd, float bc1, float bc2_sqrt) {
    w -= lr * wd * w;
    m = b1 * m + (1.f - b1) * gr;
    v = b2 * v + (1.f - b2) * gr * gr;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    return w - (lr / bc1) * (m / denom);
}
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamMulti a, float lr, float b1, float b2, float eps,
                                                           float wd, float bc1, float bc2_sqrt, float gscale) {
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.bstart[t + 1]) ++t;          // block-uniform
    const long n = a.n[t];
    const long base = (long)((int)blockIdx.x - a.bstart[t]) * ADAM_CHUNK;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    // 16-byte accesses only when every tensor allows them (gradients can be views at odd offsets, e.g. the slices
    // of a fused [d, 92+130+203] head gradient)
    const bool vec = (n & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
#pragma unroll
    for (int q = 0; q < ADAM_CHUNK / 1024; ++q) {
        const long i = base + q * 1024 + threadIdx.x * 4;
        if (vec && i + 4 <= n) {
            f32x4 w4 = *reinterpret_cast<const f32x4*>(p + i);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 m4 = *reinterpret_cast<const f32x4*>(m + i);
            f32x4 v4 = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mm = m4[k], vv = v4[k];
                w4[k] = adamw_one(w4[k], g4[k] * gscale, mm, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
                m4[k] = mm;
                v4[k] = vv;
            }
            *reinterpret_cast<f32x4*>(p + i) = w4;
            *reinterpret_cast<f32x4*>(m + i) = m4;
            *reinterpret_cast<f32x4*>(v + i) = v4;
        } else {
            for (long e = i; e < min(i + 4, n); ++e) {
                float mm = m[e], vv = v[e];
                p[e] = adamw_one(p[e], g[e] * gscale, mm, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
                m[e] = mm;
                v[e] = vv;
            }
        }
    }
}

template <typename T>
int ln_fwd_launch(const void* a, const void* res, const float* gamma, const float* beta, void* y, void* z,
                  float* mean, float* rstd, long T_rows, int d, float eps, float p, unsigned long long seed,
                  hipStream_t st) {
    constexpr int N = Vec<T>::N;
    const int nv = (d + 64 * N - 1) / (64 * N);
    dim3 grid((unsigned)((T_rows + 3) / 4)), block(256);
#define LN_FWD_CASE(NVV)                                                                                         \
    hipLaunchKernelGGL((ln_fwd_kernel<T, NVV>), grid, block, 0, st, (const T*)a, (const T*)res, gamma, beta,  \
                       (T*)y, (T*)z, mean, rstd, T_rows, d, eps, p, seed)
    if (nv <= 1) LN_FWD_CASE(1);
    else if (nv <= 2) LN_FWD_CASE(2);
    else if (nv <= 4) LN_FWD_CASE(4);
    else LN_FWD_CASE(8);
#undef LN_FWD_CASE
    MGX_CHECK_LAUNCH("ln_fwd");
    return MGX_OK;
}

template <typename T>
int ln_bwd_launch(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                  const float* gamma, void* dz, void* da, float* dgamma, float* dbeta, float* dsum, float* partial,
                  long T_rows, int d, float p, unsigned long long seed, hipStream_t st) {
    constexpr int N = Vec<T>::N;
    const int nv = (d + 64 * N - 1) / (64 * N);
    const int nblk = (int)((T_rows + 3) / 4 < PARTIAL_ROWS ? (T_rows + 3) / 4 : PARTIAL_ROWS);
    const size_t smem = (size_t)4 * d * sizeof(float);
    const int nq = dsum ? 3 : 2;
    dim3 grid(nblk), block(256);
#define LN_BWD_CASE(NVV)                                                                                          \
    hipLaunchKernelGGL((ln_bwd_kernel<T, NVV>), grid, block, smem, st, (const T*)dy, (const T*)dy2, (const T*)z, \
                       mean, rstd, gamma, (T*)dz, (T*)da, partial, T_rows, d, p, seed, dsum ? 1 : 0)
    if (nv <= 1) LN_BWD_CASE(1);
    else if (nv <= 2) LN_BWD_CASE(2);
    else if (nv <= 4) LN_BWD_CASE(4);
    else LN_BWD_CASE(8);
#undef LN_BWD_CASE
    MGX_CHECK_LAUNCH("ln_bwd");
    hipLaunchKernelGGL(column_reduce_kernel, dim3((nq * d + 63) / 64), dim3(1024), 0, st, partial, dgamma, dbeta, dsum,
                       nblk, d, nq);
    MGX_CHECK_LAUNCH("ln_bwd_reduce");
    return MGX_OK;
}


// ---- bf16 weight shadows ---------------------------------------------------------------------
// fp32 master [R,C] -> bf16 copy [R,C] and/or bf16 transposed copy [C,R] in ONE pass over the master (the
// projections want every block weight in both orientations, base_layers.BlockWeights).  64 x 64 tiles through
// LDS; every global access is a whole 128-byte (bf16) or 256-byte (fp32) row segment.  Multi-tensor: one launch
// covers up to CAST_MAX matrices (tile index -> matrix by a prefix table passed by value).
constexpr int CAST_MAX = 16, CAST_LD = 68;
struct CastMulti {
    const float* src[CAST_MAX];
    bf16_t* dst[CAST_MAX];
    bf16_t* dst_t[CAST_MAX];
    int rows[CAST_MAX], cols[CAST_MAX];
    int bstart[CAST_MAX + 1];
    int count;
};

__global__ __launch_bounds__(256) void cast_shadow_kernel(CastMulti a) {
    __shared__ __attribute__((aligned(8))) bf16_t tile[64 * CAST_LD];
    int i = 0;
    while (i + 1 < a.count && (int)blockIdx.x >= a.bstart[i + 1]) ++i;
    const int local = blockIdx.x - a.bstart[i];
    const int R = a.rows[i], C = a.cols[i];
    const int tiles_c = C / 64;
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
    const float* __restrict__ src = a.src[i];
    bf16_t* __restrict__ dst = a.dst[i];
    bf16_t* __restrict__ dst_t = a.dst_t[i];
    const int tr = threadIdx.x >> 4, tc = (threadIdx.x & 15) * 4;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (long)(r0 + tr + 16 * j) * C + c0 + tc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bf16x4 o = {(bf16_t)v[j][0], (bf16_t)v[j][1], (bf16_t)v[j][2], (bf16_t)v[j][3]};
        if (dst) *reinterpret_cast<bf16x4*>(dst + (long)(r0 + tr + 16 * j) * C + c0 + tc) = o;
        *reinterpret_cast<bf16x4*>(tile + (tr + 16 * j) * CAST_LD + tc) = o;
    }
    if (!dst_t) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tr + 16 * j;                       // source column = row of the transposed copy
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = tile[(tc + e) * CAST_LD + c];
        *reinterpret_cast<bf16x4*>(dst_t + (long)(c0 + c) * R + r0 + tc) = o;
    }
}


// ---- column sums of a bf16 matrix (bias gradient of c_attn: dqkv.sum(0), alibi_unified.py:59) ----
// stage 1: grid (n / 512, COLSUM_CHUNKS): a workgroup adds up its share of the rows for 512 columns (16-byte loads,
// 8 columns per lane, fp32 accumulation, 8 loads in flight per lane); stage 2: one thread per column adds the
// COLSUM_CHUNKS partials in a fixed order.  Deterministic, no atomics.
constexpr int COLSUM_CHUNKS = 128;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                              long T_rows, int n) {
    __shared__ float red[3][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const long per = (T_rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const long r0 = (long)blockIdx.y * per, r1 = min(r0 + per, T_rows);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (col < n) {
        for (long r = r0 + w; r < r1; r += 32) {
            bf16x8 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const bf16x8*>(x + min(r + 4 * q, r1 - 1) * n + col);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (r + 4 * q < r1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += (float)v[q][k];
                }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) red[w - 1][lane * 8 + k] = acc[k];
    }
    __syncthreads();
    if (w == 0 && col < n) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            partial[(long)blockIdx.y * n + col + k] = ((acc[k] + red[0][lane * 8 + k]) + red[1][lane * 8 + k]) + red[2][lane * 8 + k];
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float s = 0.f;
    for (int p = 0; p < COLSUM_CHUNKS; ++p) s += partial[(long)p * n + c];
    out[c] = s;
}

}  // namespace

extern "C" int mgx_ln_bwd_partial_rows(void) { return PARTIAL_ROWS; }

extern "C" int mgx_ln_residual_fwd(const void* a, const void* res, const float* gamma, const float* beta, void* y,
                                   void* z, float* mean, float* rstd, long T, int d, float eps, float p_drop,
                                   unsigned long long seed, int dtype, void* stream) {
    MGX_REQUIRE(a && gamma && beta && y, MGX_EINVAL, "mgx_ln_residual_fwd: null pointer");
    MGX_REQUIRE(T > 0 && d > 0, MGX_EINVAL, "mgx_ln_residual_fwd: T=%ld d=%d", T, d);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_ln_residual_fwd: bad dtype %d", dtype);
    const int N = dtype == MGX_F32 ? 4 : 8;
    MGX_REQUIRE(d % N == 0, MGX_EUNSUPPORTED, "mgx_ln_residual_fwd: d=%d must be a multiple of %d", d, N);
    MGX_REQUIRE(d <= 64 * N * 8, MGX_EUNSUPPORTED, "mgx_ln_residual_fwd: d=%d too large (max %d)", d, 64 * N * 8);
    MGX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, MGX_EINVAL, "mgx_ln_residual_fwd: p_drop=%f", p_drop);
    MGX_REQUIRE((z == nullptr) || (mean && rstd), MGX_EINVAL, "mgx_ln_residual_fwd: z without mean/rstd");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        return ln_fwd_launch<float>(a, res, gamma, beta, y, z, mean, rstd, T, d, eps, p_drop, seed, st);
    return ln_fwd_launch<bf16_t>(a, res, gamma, beta, y, z, mean, rstd, T, d, eps, p_drop, seed, st);
}

extern "C" int mgx_ln_residual_bwd(const void* dy, const void* dy2, const void* z, const float* mean,
                                   const float* rstd, const float* gamma, void* dz, void* da, float* dgamma,
                                   float* dbeta, float* dsum, float* partial, long T, int d, float p_drop,
                                   unsigned long long seed, int dtype, void* stream) {
    MGX_REQUIRE(dy && z && mean && rstd && gamma && dz && dgamma && dbeta && partial, MGX_EINVAL,
                "mgx_ln_residual_bwd: null pointer");
    MGX_REQUIRE(T > 0 && d > 0, MGX_EINVAL, "mgx_ln_residual_bwd: T=%ld d=%d", T, d);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_ln_residual_bwd: bad dtype %d", dtype);
    const int N = dtype == MGX_F32 ? 4 : 8;
    MGX_REQUIRE(d % N == 0 && d <= 64 * N * 8, MGX_EUNSUPPORTED, "mgx_ln_residual_bwd: unsupported d=%d", d);
    MGX_REQUIRE((size_t)4 * d * sizeof(float) <= 64 * 1024, MGX_EUNSUPPORTED, "mgx_ln_residual_bwd: d=%d too large", d);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        return ln_bwd_launch<float>(dy, dy2, z, mean, rstd, gamma, dz, da, dgamma, dbeta, dsum, partial, T, d, p_drop, seed, st);
    return ln_bwd_launch<bf16_t>(dy, dy2, z, mean, rstd, gamma, dz, da, dgamma, dbeta, dsum, partial, T, d, p_drop, seed, st);
}

extern "C" int mgx_bias_act_fwd(const void* u, const float* bias, void* pre, void* h, long T, int n, int act,
                                float p_drop, unsigned long long seed, int dtype, void* stream) {
    MGX_REQUIRE(u && h, MGX_EINVAL, "mgx_bias_act_fwd: null pointer");
    MGX_REQUIRE(T > 0 && n > 0, MGX_EINVAL, "mgx_bias_act_fwd: T=%ld n=%d", T, n);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_bias_act_fwd: bad dtype %d", dtype);
    MGX_REQUIRE(act == MGX_ACT_GELU || act == MGX_ACT_RELU, MGX_EINVAL, "mgx_bias_act_fwd: bad act %d", act);
    const int N = dtype == MGX_F32 ? 4 : 8;
    MGX_REQUIRE(n % N == 0, MGX_EUNSUPPORTED, "mgx_bias_act_fwd: n=%d must be a multiple of %d", n, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32) return bias_act_fwd_launch<float>(u, bias, pre, h, T, n, act, p_drop, seed, st);
    return bias_act_fwd_launch<bf16_t>(u, bias, pre, h, T, n, act, p_drop, seed, st);
}

extern "C" int mgx_bias_act_bwd(const void* dh, const void* pre, void* dpre, float* dbias, float* partial, long T,
                                int n, int act, float p_drop, unsigned long long seed, int dtype, void* stream) {
    MGX_REQUIRE(dh && pre && dpre && dbias && partial, MGX_EINVAL, "mgx_bias_act_bwd: null pointer");
    MGX_REQUIRE(T > 0 && n > 0, MGX_EINVAL, "mgx_bias_act_bwd: T=%ld n=%d", T, n);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_bias_act_bwd: bad dtype %d", dtype);
    const int N = dtype == MGX_F32 ? 4 : 8;
    MGX_REQUIRE(n % N == 0, MGX_EUNSUPPORTED, "mgx_bias_act_bwd: n=%d must be a multiple of %d", n, N);
    const int nblk = (int)(T < PARTIAL_ROWS ? T : PARTIAL_ROWS);
    hipStream_t st = (hipStream_t)stream;
    MGX_REQUIRE(act == MGX_ACT_GELU || act == MGX_ACT_RELU, MGX_EINVAL, "mgx_bias_act_bwd: bad act %d", act);
    const int rc = dtype == MGX_F32
                       ? bias_act_bwd_launch<float>(dh, pre, dpre, partial, T, n, nblk, act, p_drop, seed, st)
                       : bias_act_bwd_launch<bf16_t>(dh, pre, dpre, partial, T, n, nblk, act, p_drop, seed, st);
    if (rc != MGX_OK) return rc;
    hipLaunchKernelGGL(column_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, st, partial, dbias, dbias, dbias, nblk, n, 1);
    MGX_CHECK_LAUNCH("bias_act_bwd_reduce");
    return MGX_OK;
}

extern "C" int mgx_adamw(float* param, const void* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, long n,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float grad_scale, int grad_dtype, void* stream) {
    MGX_REQUIRE(param && grad && exp_avg && exp_avg_sq, MGX_EINVAL, "mgx_adamw: null pointer");
    MGX_REQUIRE(n > 0 && step >= 1, MGX_EINVAL, "mgx_adamw: n=%ld step=%d", n, step);
    MGX_REQUIRE(grad_dtype == MGX_F32 || grad_dtype == MGX_BF16, MGX_EINVAL, "mgx_adamw: bad grad dtype");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (grad_dtype == MGX_F32)
        hipLaunchKernelGGL((adamw_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, param, (const float*)grad,
                           exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                           grad_scale);
    else
        hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, param, (const bf16_t*)grad,
                           exp_avg, exp_avg_sq, (bf16_t*)shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                           grad_scale);
    MGX_CHECK_LAUNCH("adamw");
    return MGX_OK;
}

extern "C" int mgx_adamw_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const long* sizes, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int step, float grad_scale, void* stream) {
    MGX_REQUIRE(count > 0 && params && grads && exp_avg && exp_avg_sq && sizes, MGX_EINVAL,
                "mgx_adamw_multi: null pointer or count=%d", count);
    MGX_REQUIRE(step >= 1, MGX_EINVAL, "mgx_adamw_multi: step=%d", step);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < count; t0 += ADAM_MAX) {
        AdamMulti a;
        a.count = count - t0 < ADAM_MAX ? count - t0 : ADAM_MAX;
        long blocks = 0;
        for (int i = 0; i < a.count; ++i) {
            const int t = t0 + i;
            MGX_REQUIRE(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t] && sizes[t] > 0, MGX_EINVAL,
                        "mgx_adamw_multi: tensor %d: null pointer or empty", t);
            MGX_REQUIRE(((uintptr_t)params[t] | (uintptr_t)grads[t] | (uintptr_t)exp_avg[t] | (uintptr_t)exp_avg_sq[t]) % 4 == 0,
                        MGX_EUNSUPPORTED, "mgx_adamw_multi: tensor %d is not 4-byte aligned", t);
            a.p[i] = params[t];
            a.g[i] = grads[t];
            a.m[i] = exp_avg[t];
            a.v[i] = exp_avg_sq[t];
            a.n[i] = sizes[t];
            a.bstart[i] = (int)blocks;
            blocks += (sizes[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            MGX_REQUIRE(blocks < (1L << 30), MGX_EUNSUPPORTED, "mgx_adamw_multi: too many elements in one launch");
        }
        a.bstart[a.count] = (int)blocks;
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, lr, beta1, beta2, eps,
                           weight_decay, bc1, bc2s, grad_scale);
        MGX_CHECK_LAUNCH("adamw_multi");
    }
    return MGX_OK;
}

extern "C" int mgx_cast_shadow_bf16(int count, const float* const* src, void* const* dst, void* const* dst_t,
                                    const int* rows, const int* cols, void* stream) {
    MGX_REQUIRE(count > 0 && src && dst && dst_t && rows && cols, MGX_EINVAL, "mgx_cast_shadow_bf16: null pointer or count=%d", count);
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < count; t0 += CAST_MAX) {
        CastMulti a;
        a.count = count - t0 < CAST_MAX ? count - t0 : CAST_MAX;
        long blocks = 0;
        for (int i = 0; i < a.count; ++i) {
            const int t = t0 + i;
            MGX_REQUIRE(src[t] && (dst[t] || dst_t[t]), MGX_EINVAL, "mgx_cast_shadow_bf16: matrix %d: null source or no destination", t);
            MGX_REQUIRE(rows[t] > 0 && cols[t] > 0 && rows[t] % 64 == 0 && cols[t] % 64 == 0, MGX_EUNSUPPORTED,
                        "mgx_cast_shadow_bf16: matrix %d is %d x %d (both must be multiples of 64)", t, rows[t], cols[t]);
            MGX_REQUIRE((uintptr_t)src[t] % 16 == 0 && (uintptr_t)dst[t] % 8 == 0 && (uintptr_t)dst_t[t] % 8 == 0, MGX_EUNSUPPORTED,
                        "mgx_cast_shadow_bf16: matrix %d is not aligned (16 B source, 8 B destinations)", t);
            a.src[i] = src[t];
            a.dst[i] = (bf16_t*)dst[t];
            a.dst_t[i] = (bf16_t*)dst_t[t];
            a.rows[i] = rows[t];
            a.cols[i] = cols[t];
            a.bstart[i] = (int)blocks;
            blocks += (long)(rows[t] / 64) * (cols[t] / 64);
            MGX_REQUIRE(blocks < (1L << 30), MGX_EUNSUPPORTED, "mgx_cast_shadow_bf16: too many tiles in one launch");
        }
        a.bstart[a.count] = (int)blocks;
        hipLaunchKernelGGL(cast_shadow_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
        MGX_CHECK_LAUNCH("cast_shadow");
    }
    return MGX_OK;
}

extern "C" long mgx_colsum_workspace_floats(int n) { return n > 0 ? (long)COLSUM_CHUNKS * n : 0; }

// ---- mgx_colsum_bf16_static: the same reduction without a caller-provided workspace ----------------------------
// For callers that cannot allocate (HIP-graph capture, serving loops, harnesses that forbid hipMalloc): the stage-1
// partial sums live in a file-scope device array, 4 MiB per GPU for the life of the process.  In host code the
// array's name is only a stand-in (not a device address): its address on the CURRENT device is looked up once per
// device with hipGetSymbolAddress and that pointer is what the kernels receive.
constexpr int COLSUM_STATIC_MAX_N = 8192;
static __device__ __align__(256) float scratch[COLSUM_CHUNKS * COLSUM_STATIC_MAX_N];
constexpr int MGX_MAX_DEVICES = 64;
static float* g_scratch_ptr[MGX_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

static int colsum_static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    MGX_REQUIRE(e == hipSuccess && dev >= 0 && dev < MGX_MAX_DEVICES, MGX_ELAUNCH,
                "mgx_colsum_bf16_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));       // this device's copy of the array
        MGX_REQUIRE(e == hipSuccess && p, MGX_ELAUNCH, "mgx_colsum_bf16_static: hipGetSymbolAddress: %s",
                    hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return MGX_OK;
}

extern "C" int mgx_colsum_static_max_n(void) { return COLSUM_STATIC_MAX_N; }

extern "C" int mgx_colsum_bf16_static(const void* x, float* out, long T, int n, void* stream) {
    MGX_REQUIRE(x && out, MGX_EINVAL, "mgx_colsum_bf16_static: null pointer");
    MGX_REQUIRE(T > 0 && n > 0 && n % 8 == 0, MGX_EUNSUPPORTED, "mgx_colsum_bf16_static: T=%ld n=%d (n must be a multiple of 8)", T, n);
    MGX_REQUIRE(n <= COLSUM_STATIC_MAX_N, MGX_EUNSUPPORTED,
                "mgx_colsum_bf16_static: n=%d exceeds the static scratch (at most %d columns; use mgx_colsum_bf16 with a "
                "workspace for wider matrices)", n, COLSUM_STATIC_MAX_N);
    float* ws = nullptr;
    if (int rc = colsum_static_scratch(&ws)) return rc;
    // every partial the second stage reads is written by the first (all COLSUM_CHUNKS rows of its n columns): nothing
    // has to start at zero, whatever an earlier (larger) call left behind
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((n + 511) / 512, COLSUM_CHUNKS), dim3(256), 0, st, (const bf16_t*)x,
                       ws, T, n);
    MGX_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, out, n);
    MGX_CHECK_LAUNCH("colsum_final");
    return MGX_OK;
}

extern "C" int mgx_colsum_bf16(const void* x, float* out, float* workspace, long T, int n, void* stream) {
    MGX_REQUIRE(x && out && workspace, MGX_EINVAL, "mgx_colsum_bf16: null pointer");
    MGX_REQUIRE(T > 0 && n > 0 && n % 8 == 0, MGX_EUNSUPPORTED, "mgx_colsum_bf16: T=%ld n=%d (n must be a multiple of 8)", T, n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((n + 511) / 512, COLSUM_CHUNKS), dim3(256), 0, st, (const bf16_t*)x,
                       workspace, T, n);
    MGX_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, workspace, out, n);
    MGX_CHECK_LAUNCH("colsum_final");
    return MGX_OK;
}
