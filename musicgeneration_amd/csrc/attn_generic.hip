// Generic causal attention kernels (any head dim <= 256, fp32 or bf16 storage,
// fp32 math): ALiBi bias (model_baseline/alibi_unified.py:40-54) and skew
// relative bias (old_models/relative_transformer.py:18-44).
//
// These are the exact-fp32 parity path and the fallback for head dims the MFMA
// kernels do not cover.  One workgroup = 4 waves = 4 consecutive rows sharing
// 64-key (or 64-query) tiles staged in LDS; one lane per key, wavefront
// reductions for the softmax.  HBM traffic is the same as the flash kernels
// (nothing of size S x S is ever written).
#include "mgx_common.h"

namespace {

constexpr int ROWS = 4;        // rows (waves) per workgroup
constexpr int TILE = 64;       // keys per tile = lanes per wave
constexpr int MAXC = 4;        // dh <= 64*MAXC

enum { MODE_ALIBI = 0, MODE_REL = 1 };

struct AttnShape {
    int B, S, H, dh, max_seq;
    float scale;
};

template <typename T>
__device__ __forceinline__ void stage_rows(float* dst, int ld, const T* src, long row_stride, int r0, int S,
                                           int dh, int tid, int nthreads) {
    // dst[r][d] = src[(r0 + r) * row_stride + d], zero for rows >= S
    for (int e = tid; e < TILE * dh; e += nthreads) {
        int r = e / dh, d = e - r * dh;
        int g = r0 + r;
        dst[r * ld + d] = (g < S) ? to_f32(src[(long)g * row_stride + d]) : 0.f;
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_fwd_generic(const T* __restrict__ qkv, const float* __restrict__ slopes,
                                                         const T* __restrict__ E, T* __restrict__ out,
                                                         float* __restrict__ lse, AttnShape sh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = sh.S, H = sh.H, dh = sh.dh, ld = dh + 1;
    float* Ks = smem;                      // [TILE][ld]
    float* Vs = Ks + TILE * ld;            // [TILE][ld]
    float* Qs = Vs + TILE * ld;            // [ROWS][dh]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ROWS;
    const int i = i0 + w;
    const bool active = i < S;
    const long rs = 3L * H * dh;
    const T* base = qkv + (long)b * S * rs + (long)h * dh;
    const T* qp = base;
    const T* kp = base + (long)H * dh;
    const T* vp = base + 2L * H * dh;
    for (int d = lane; d < dh; d += 64) Qs[w * dh + d] = active ? to_f32(qp[(long)i * rs + d]) : 0.f;
    const float slope = (MODE == MODE_ALIBI) ? slopes[h] : 0.f;
    const T* Eh = (MODE == MODE_REL) ? E + (long)h * sh.max_seq * dh : nullptr;

    float m = -INFINITY, l = 0.f;
    float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
    const int i_last = min(i0 + ROWS - 1, S - 1);
    for (int j0 = 0; j0 <= i_last; j0 += TILE) {
        __syncthreads();
        stage_rows(Ks, ld, kp, rs, j0, S, dh, tid, 256);
        stage_rows(Vs, ld, vp, rs, j0, S, dh, tid, 256);
        __syncthreads();
        if (active && j0 <= i) {
            const int j = j0 + lane;
            float s = 0.f;
            const float* kr = Ks + lane * ld;
            const float* qr = Qs + w * dh;
            for (int d = 0; d < dh; ++d) s = fmaf(qr[d], kr[d], s);
            if (MODE == MODE_REL) {
                if (j <= i) {
                    const T* er = Eh + (long)(sh.max_seq - 1 - (i - j)) * dh;
                    float r = 0.f;
                    for (int d = 0; d < dh; ++d) r = fmaf(qr[d], to_f32(er[d]), r);
                    s += r;
                }
                s *= sh.scale;
            } else {
                s = s * sh.scale + slope * (float)(j - i);
            }
            if (j > i) s = -INFINITY;
            const float m_new = fmaxf(m, wave_max(s));
            const float p = expf(s - m_new);
            const float alpha = expf(m - m_new);
            l = l * alpha + wave_sum(p);
            m = m_new;
            const int jn = min(TILE, i - j0 + 1);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= dh) break;           // wave-uniform
                const int d = lane + 64 * c;
                const bool ok = d < dh;
                float a = acc[c] * alpha;
                for (int jj = 0; jj < jn; ++jj) {
                    const float pj = __shfl(p, jj, 64);   // every lane takes part in the exchange
                    if (ok) a = fmaf(pj, Vs[jj * ld + d], a);
                }
                acc[c] = a;
            }
        }
    }
    if (active) {
        const float inv = 1.f / l;
        T* op = out + ((long)b * S + i) * H * dh + (long)h * dh;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int d = lane + 64 * c;
            if (d < dh) op[d] = from_f32<T>(acc[c] * inv);
        }
        if (lane == 0) lse[((long)b * H + h) * S + i] = m + logf(l);
    }
}

// delta[b,h,i] = sum_d dout[b,i,h,d] * out[b,i,h,d]
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ dout, const T* __restrict__ out, float* __restrict__ delta,
                                  int B, int S, int H, int dh) {
    const long n = (long)B * S * H;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int h = t % H;
    const long bs = t / H;
    const int i = bs % S;
    const int b = bs / S;
    const T* a = dout + t * dh;
    const T* o = out + t * dh;
    float s = 0.f;
    for (int d = 0; d < dh; ++d) s = fmaf(to_f32(a[d]), to_f32(o[d]), s);
    delta[((long)b * H + h) * S + i] = s;
}

// dQ: one wave per query row; tiles of 64 keys (K and V staged in LDS).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_dq_generic(const T* __restrict__ dout, const T* __restrict__ qkv,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta,
                                                            const float* __restrict__ slopes, const T* __restrict__ E,
                                                            T* __restrict__ dqkv, AttnShape sh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = sh.S, H = sh.H, dh = sh.dh, ld = dh + 1;
    float* Ks = smem;
    float* Vs = Ks + TILE * ld;
    float* Qs = Vs + TILE * ld;            // [ROWS][dh]
    float* Gs = Qs + ROWS * dh;            // [ROWS][dh]  dout rows
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ROWS;
    const int i = i0 + w;
    const bool active = i < S;
    const long rs = 3L * H * dh;
    const T* base = qkv + (long)b * S * rs + (long)h * dh;
    const T* kp = base + (long)H * dh;
    const T* vp = base + 2L * H * dh;
    const T* gp = dout + (long)b * S * H * dh + (long)h * dh;
    for (int d = lane; d < dh; d += 64) {
        Qs[w * dh + d] = active ? to_f32(base[(long)i * rs + d]) : 0.f;
        Gs[w * dh + d] = active ? to_f32(gp[(long)i * H * dh + d]) : 0.f;
    }
    const float slope = (MODE == MODE_ALIBI) ? slopes[h] : 0.f;
    const T* Eh = (MODE == MODE_REL) ? E + (long)h * sh.max_seq * dh : nullptr;
    const float lse_i = active ? lse[((long)b * H + h) * S + i] : 0.f;
    const float delta_i = active ? delta[((long)b * H + h) * S + i] : 0.f;

    float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
    const int i_last = min(i0 + ROWS - 1, S - 1);
    for (int j0 = 0; j0 <= i_last; j0 += TILE) {
        __syncthreads();
        stage_rows(Ks, ld, kp, rs, j0, S, dh, tid, 256);
        stage_rows(Vs, ld, vp, rs, j0, S, dh, tid, 256);
        __syncthreads();
        if (active && j0 <= i) {
            const int j = j0 + lane;
            const float* kr = Ks + lane * ld;
            const float* vr = Vs + lane * ld;
            const float* qr = Qs + w * dh;
            const float* gr = Gs + w * dh;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < dh; ++d) {
                s = fmaf(qr[d], kr[d], s);
                dp = fmaf(gr[d], vr[d], dp);
            }
            if (MODE == MODE_REL) {
                if (j <= i) {
                    const T* er = Eh + (long)(sh.max_seq - 1 - (i - j)) * dh;
                    float r = 0.f;
                    for (int d = 0; d < dh; ++d) r = fmaf(qr[d], to_f32(er[d]), r);
                    s += r;
                }
                s *= sh.scale;
            } else {
                s = s * sh.scale + slope * (float)(j - i);
            }
            const float p = (j <= i) ? expf(s - lse_i) : 0.f;
            const float ds = p * (dp - delta_i) * sh.scale;
            const int jn = min(TILE, i - j0 + 1);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= dh) break;
                const int d = lane + 64 * c;
                const bool ok = d < dh;
                float a = acc[c];
                for (int jj = 0; jj < jn; ++jj) {
                    const float dsj = __shfl(ds, jj, 64);
                    if (ok) {
                        float kv = Ks[jj * ld + d];
                        if (MODE == MODE_REL) kv += to_f32(Eh[(long)(sh.max_seq - 1 - (i - j0 - jj)) * dh + d]);
                        a = fmaf(dsj, kv, a);
                    }
                }
                acc[c] = a;
            }
        }
    }
    if (active) {
        T* dq = dqkv + ((long)b * S + i) * rs + (long)h * dh;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int d = lane + 64 * c;
            if (d < dh) dq[d] = from_f32<T>(acc[c]);
        }
    }
}

// dK, dV: one wave per key row; tiles of 64 queries (Q and dout staged in LDS).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_dkv_generic(const T* __restrict__ dout, const T* __restrict__ qkv,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ delta,
                                                             const float* __restrict__ slopes, const T* __restrict__ E,
                                                             T* __restrict__ dqkv, AttnShape sh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = sh.S, H = sh.H, dh = sh.dh, ld = dh + 1;
    float* Qt = smem;                      // [TILE][ld]
    float* Gt = Qt + TILE * ld;            // [TILE][ld]
    float* Kr = Gt + TILE * ld;            // [ROWS][dh]
    float* Vr = Kr + ROWS * dh;            // [ROWS][dh]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * ROWS;
    const int j = j0 + w;
    const bool active = j < S;
    const long rs = 3L * H * dh;
    const T* base = qkv + (long)b * S * rs + (long)h * dh;
    const T* kp = base + (long)H * dh;
    const T* vp = base + 2L * H * dh;
    const T* gp = dout + (long)b * S * H * dh + (long)h * dh;
    for (int d = lane; d < dh; d += 64) {
        Kr[w * dh + d] = active ? to_f32(kp[(long)j * rs + d]) : 0.f;
        Vr[w * dh + d] = active ? to_f32(vp[(long)j * rs + d]) : 0.f;
    }
    const float slope = (MODE == MODE_ALIBI) ? slopes[h] : 0.f;
    const T* Eh = (MODE == MODE_REL) ? E + (long)h * sh.max_seq * dh : nullptr;
    const float* lse_bh = lse + ((long)b * H + h) * S;
    const float* delta_bh = delta + ((long)b * H + h) * S;

    float dk[MAXC] = {0.f, 0.f, 0.f, 0.f}, dv[MAXC] = {0.f, 0.f, 0.f, 0.f};
    const int t0 = (j0 / TILE) * TILE;
    for (int q0 = t0; q0 < S; q0 += TILE) {
        __syncthreads();
        stage_rows(Qt, ld, base, rs, q0, S, dh, tid, 256);
        stage_rows(Gt, ld, gp, (long)H * dh, q0, S, dh, tid, 256);
        __syncthreads();
        if (active && q0 + TILE - 1 >= j) {
            const int i = q0 + lane;
            const bool valid = (i < S) && (i >= j);
            const float* qr = Qt + lane * ld;
            const float* gr = Gt + lane * ld;
            const float* kr = Kr + w * dh;
            const float* vr = Vr + w * dh;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < dh; ++d) {
                s = fmaf(qr[d], kr[d], s);
                dp = fmaf(gr[d], vr[d], dp);
            }
            if (MODE == MODE_REL) {
                if (valid) {
                    const T* er = Eh + (long)(sh.max_seq - 1 - (i - j)) * dh;
                    float r = 0.f;
                    for (int d = 0; d < dh; ++d) r = fmaf(qr[d], to_f32(er[d]), r);
                    s += r;
                }
                s *= sh.scale;
            } else {
                s = s * sh.scale + slope * (float)(j - i);
            }
            const float p = valid ? expf(s - lse_bh[i]) : 0.f;
            const float ds = valid ? p * (dp - delta_bh[i]) * sh.scale : 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= dh) break;
                const int d = lane + 64 * c;
                const bool ok = d < dh;
                float ak = dk[c], av = dv[c];
                for (int ii = 0; ii < TILE; ++ii) {
                    const float pi = __shfl(p, ii, 64);
                    const float dsi = __shfl(ds, ii, 64);
                    if (ok) {
                        av = fmaf(pi, Gt[ii * ld + d], av);
                        ak = fmaf(dsi, Qt[ii * ld + d], ak);
                    }
                }
                dk[c] = ak;
                dv[c] = av;
            }
        }
    }
    if (active) {
        T* o = dqkv + ((long)b * S + j) * rs + (long)h * dh;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int d = lane + 64 * c;
            if (d < dh) {
                o[(long)H * dh + d] = from_f32<T>(dk[c]);
                o[2L * H * dh + d] = from_f32<T>(dv[c]);
            }
        }
    }
}

// dE[h, r, :] = sum_b sum_{i-j = max_seq-1-r} ds[b,h,i,j] * q[b,h,i,:]   (one wave per (r,h))
template <typename T>
__global__ __launch_bounds__(64) void rel_bwd_dE_generic(const T* __restrict__ dout, const T* __restrict__ qkv,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta, const T* __restrict__ E,
                                                          float* __restrict__ dE, AttnShape sh) {
    const int S = sh.S, H = sh.H, dh = sh.dh, M = sh.max_seq;
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    float* o = dE + ((long)h * M + r) * dh;
    const int t = M - 1 - r;               // i - j
    if (t >= S) {
        for (int d = lane; d < dh; d += 64) o[d] = 0.f;
        return;
    }
    const long rs = 3L * H * dh;
    const T* er = E + ((long)h * M + r) * dh;
    float acc[MAXC] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < sh.B; ++b) {
        const T* base = qkv + (long)b * S * rs + (long)h * dh;
        const T* gp = dout + (long)b * S * H * dh + (long)h * dh;
        const float* lse_bh = lse + ((long)b * H + h) * S;
        const float* delta_bh = delta + ((long)b * H + h) * S;
        for (int q0 = t; q0 < S; q0 += 64) {
            const int i = q0 + lane;
            float ds = 0.f;
            if (i < S) {
                const int j = i - t;
                const T* qr = base + (long)i * rs;
                const T* kr = base + (long)j * rs + (long)H * dh;
                const T* vr = base + (long)j * rs + 2L * H * dh;
                const T* gr = gp + (long)i * H * dh;
                float s = 0.f, dp = 0.f;
                for (int d = 0; d < dh; ++d) {
                    const float q = to_f32(qr[d]);
                    s = fmaf(q, to_f32(kr[d]) + to_f32(er[d]), s);
                    dp = fmaf(to_f32(gr[d]), to_f32(vr[d]), dp);
                }
                const float p = expf(s * sh.scale - lse_bh[i]);
                ds = p * (dp - delta_bh[i]) * sh.scale;
            }
            const int n = min(64, S - q0);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (64 * c >= dh) break;
                const int d = lane + 64 * c;
                const bool ok = d < dh;
                float a = acc[c];
                for (int ii = 0; ii < n; ++ii) {
                    const float dsi = __shfl(ds, ii, 64);
                    if (ok) a = fmaf(dsi, to_f32(base[(long)(q0 + ii) * rs + d]), a);
                }
                acc[c] = a;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int d = lane + 64 * c;
        if (d < dh) o[d] = acc[c];
    }
}

size_t fwd_smem(int dh) { return (size_t)(2 * TILE * (dh + 1) + ROWS * dh) * sizeof(float); }
size_t bwd_smem(int dh) { return (size_t)(2 * TILE * (dh + 1) + 2 * ROWS * dh) * sizeof(float); }

template <typename F>
void allow_smem(F* fn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T, int MODE>
int launch_fwd(const void* qkv, const float* slopes, const void* E, void* out, float* lse, AttnShape sh,
               hipStream_t st) {
    dim3 grid((sh.S + ROWS - 1) / ROWS, sh.H, sh.B);
    allow_smem(attn_fwd_generic<T, MODE>, fwd_smem(sh.dh));
    hipLaunchKernelGGL((attn_fwd_generic<T, MODE>), grid, dim3(256), fwd_smem(sh.dh), st, (const T*)qkv, slopes,
                       (const T*)E, (T*)out, lse, sh);
    MGX_CHECK_LAUNCH("attn_fwd_generic");
    return MGX_OK;
}

template <typename T, int MODE>
int launch_bwd(const void* dout, const void* qkv, const void* out, const float* lse, const float* slopes,
               const void* E, void* dqkv, float* dE, float* delta, AttnShape sh, hipStream_t st) {
    const long n = (long)sh.B * sh.S * sh.H;
    hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const T*)dout,
                       (const T*)out, delta, sh.B, sh.S, sh.H, sh.dh);
    MGX_CHECK_LAUNCH("attn_delta");
    dim3 grid((sh.S + ROWS - 1) / ROWS, sh.H, sh.B);
    allow_smem(attn_bwd_dq_generic<T, MODE>, bwd_smem(sh.dh));
    allow_smem(attn_bwd_dkv_generic<T, MODE>, bwd_smem(sh.dh));
    hipLaunchKernelGGL((attn_bwd_dq_generic<T, MODE>), grid, dim3(256), bwd_smem(sh.dh), st, (const T*)dout,
                       (const T*)qkv, lse, delta, slopes, (const T*)E, (T*)dqkv, sh);
    MGX_CHECK_LAUNCH("attn_bwd_dq_generic");
    hipLaunchKernelGGL((attn_bwd_dkv_generic<T, MODE>), grid, dim3(256), bwd_smem(sh.dh), st, (const T*)dout,
                       (const T*)qkv, lse, delta, slopes, (const T*)E, (T*)dqkv, sh);
    MGX_CHECK_LAUNCH("attn_bwd_dkv_generic");
    if (MODE == MODE_REL) {
        hipLaunchKernelGGL((rel_bwd_dE_generic<T>), dim3(sh.max_seq, sh.H), dim3(64), 0, st, (const T*)dout,
                           (const T*)qkv, lse, delta, (const T*)E, dE, sh);
        MGX_CHECK_LAUNCH("rel_bwd_dE_generic");
    }
    return MGX_OK;
}

}  // namespace

int mgx_check_attn_shape(const char* fn, int B, int S, int H, int dh, int dtype) {
    MGX_REQUIRE(B > 0 && S > 0 && H > 0 && dh > 0, MGX_EINVAL, "%s: non-positive shape B=%d S=%d H=%d dh=%d", fn, B, S,
                H, dh);
    MGX_REQUIRE(dh <= 64 * MAXC, MGX_EUNSUPPORTED, "%s: head dim %d > %d", fn, dh, 64 * MAXC);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "%s: bad dtype %d", fn, dtype);
    MGX_REQUIRE(B <= 65535 && H <= 65535, MGX_EUNSUPPORTED, "%s: B or H exceeds the grid limit", fn);
    return MGX_OK;
}

int mgx_attn_fwd_generic(int mode, const void* qkv, const float* slopes, const void* E, void* out, float* lse, int B,
                         int S, int H, int dh, int max_seq, int dtype, hipStream_t st) {
    AttnShape sh{B, S, H, dh, max_seq, 1.0f / sqrtf((float)dh)};
    if (dtype == MGX_F32)
        return mode == MODE_ALIBI ? launch_fwd<float, MODE_ALIBI>(qkv, slopes, E, out, lse, sh, st)
                                  : launch_fwd<float, MODE_REL>(qkv, slopes, E, out, lse, sh, st);
    return mode == MODE_ALIBI ? launch_fwd<bf16_t, MODE_ALIBI>(qkv, slopes, E, out, lse, sh, st)
                              : launch_fwd<bf16_t, MODE_REL>(qkv, slopes, E, out, lse, sh, st);
}

int mgx_attn_bwd_generic(int mode, const void* dout, const void* qkv, const void* out, const float* lse,
                         const float* slopes, const void* E, void* dqkv, float* dE, float* delta, int B, int S, int H,
                         int dh, int max_seq, int dtype, hipStream_t st) {
    AttnShape sh{B, S, H, dh, max_seq, 1.0f / sqrtf((float)dh)};
    if (dtype == MGX_F32)
        return mode == MODE_ALIBI
                   ? launch_bwd<float, MODE_ALIBI>(dout, qkv, out, lse, slopes, E, dqkv, dE, delta, sh, st)
                   : launch_bwd<float, MODE_REL>(dout, qkv, out, lse, slopes, E, dqkv, dE, delta, sh, st);
    return mode == MODE_ALIBI
               ? launch_bwd<bf16_t, MODE_ALIBI>(dout, qkv, out, lse, slopes, E, dqkv, dE, delta, sh, st)
               : launch_bwd<bf16_t, MODE_REL>(dout, qkv, out, lse, slopes, E, dqkv, dE, delta, sh, st);
}
