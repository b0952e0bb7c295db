// KV-cached decode attention for the sampling loop (replaces the full re-forward
// per token of model_baseline/inference_unified.py:12-21).
//
// HBM-bound: each (batch, head) streams its K and V cache rows once per step.
// One workgroup (4 waves) per (b, h); a key row of dh elements is spread over
// LPR = dh*sizeof(T)/16 lanes with one 16-byte load each, so one wave-instruction
// fetches 64/LPR whole rows (1 KiB, fully coalesced).  Up to 4 query rows (the
// "i+1 quirk" needs 2-3 provisional rows per step) share every K/V
// load.  Softmax statistics are kept per lane group and merged by wavefront
// shuffles, then across the 4 waves through LDS.
// Cache layout: [B, H, Lmax, dh], contiguous rows per (b, h) -> pure streaming.
#include "mgx_common.h"

namespace {

constexpr int MAXR = 4;

template <typename T> struct DVec;
template <> struct DVec<float> {
    static constexpr int N = 4;
    typedef f32x4 type;
};
template <> struct DVec<bf16_t> {
    static constexpr int N = 8;
    typedef bf16x8 type;
};

template <typename T>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                           const T* __restrict__ vc, const float* __restrict__ slopes,
                                                           T* __restrict__ out, int R, int H, int dh, int Lmax,
                                                           int pos0_host, const int* __restrict__ pos_dev,
                                                           float scale) {
    constexpr int N = DVec<T>::N;
    typedef typename DVec<T>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][MAXR][dh + 2]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x;
    int pos0 = pos_dev ? *pos_dev : pos0_host;
    if (pos0 + R > Lmax) pos0 = Lmax - R;   // device-side positions cannot be checked on the host
    const int lpr = dh / N;              // lanes per key row (power of two, <= 64)
    const int rpl = 64 / lpr;            // key rows per wave-load
    const int sub = lane / lpr;          // which of the rpl rows this lane reads
    const int ch = lane - sub * lpr;     // 16-byte chunk inside the row
    const float slope = slopes ? slopes[h] : 0.f;

    float qv[MAXR][N];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r < R) {
            vec_t t = *reinterpret_cast<const vec_t*>(q + (((long)b * R + r) * H + h) * dh + ch * N);
#pragma unroll
            for (int k = 0; k < N; ++k) qv[r][k] = to_f32(t[k]) * scale;
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) qv[r][k] = 0.f;
        }
    }
    float m[MAXR], l[MAXR], acc[MAXR][N];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) acc[r][k] = 0.f;
    }
    const T* kb = kc + ((long)b * H + h) * Lmax * dh;
    const T* vb = vc + ((long)b * H + h) * Lmax * dh;
    const int nkeys = pos0 + R;          // keys 0 .. pos0+R-1
    for (int j0 = w * rpl; j0 < nkeys; j0 += 4 * rpl) {
        const int j = j0 + sub;
        const bool inb = j < nkeys;
        vec_t kvec, vvec;
        if (inb) {
            kvec = *reinterpret_cast<const vec_t*>(kb + (long)j * dh + ch * N);
            vvec = *reinterpret_cast<const vec_t*>(vb + (long)j * dh + ch * N);
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                kvec[k] = from_f32<T>(0.f);
                vvec[k] = from_f32<T>(0.f);
            }
        }
        float kf[N], vf[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            kf[k] = to_f32(kvec[k]);
            vf[k] = to_f32(vvec[k]);
        }
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < R) {                                   // uniform
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < N; ++k) s = fmaf(qv[r][k], kf[k], s);
                for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o, 64);
                const int i = pos0 + r;
                s += slope * (float)(j - i);
                if (!inb || j > i) s = -INFINITY;
                const float mn = fmaxf(m[r], s);
                // mn may still be -inf when every key seen so far is masked
                const float alpha = (mn == -INFINITY) ? 1.f : expf(m[r] - mn);
                const float p = (mn == -INFINITY) ? 0.f : expf(s - mn);
                l[r] = l[r] * alpha + p;
                m[r] = mn;
#pragma unroll
                for (int k = 0; k < N; ++k) acc[r][k] = fmaf(p, vf[k], acc[r][k] * alpha);
            }
        }
    }
    // merge the rpl lane groups of this wave
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r < R) {
            for (int o = lpr; o < 64; o <<= 1) {
                const float m2 = __shfl_xor(m[r], o, 64);
                const float l2 = __shfl_xor(l[r], o, 64);
                const float mn = fmaxf(m[r], m2);
                const float a1 = (mn == -INFINITY) ? 1.f : expf(m[r] - mn);
                const float a2 = (mn == -INFINITY) ? 1.f : expf(m2 - mn);
                l[r] = l[r] * a1 + l2 * a2;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const float other = __shfl_xor(acc[r][k], o, 64);
                    acc[r][k] = acc[r][k] * a1 + other * a2;
                }
                m[r] = mn;
            }
        }
    }
    // merge the 4 waves through LDS
    const int ldw = dh + 2;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r < R && sub == 0) {
            float* dst = smem + ((size_t)w * MAXR + r) * ldw;
#pragma unroll
            for (int k = 0; k < N; ++k) dst[ch * N + k] = acc[r][k];
            if (ch == 0) {
                dst[dh] = m[r];
                dst[dh + 1] = l[r];
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < R * dh; e += 256) {
        const int r = e / dh, d = e - r * dh;
        float mm = -INFINITY;
        for (int ww = 0; ww < 4; ++ww) mm = fmaxf(mm, smem[((size_t)ww * MAXR + r) * ldw + dh]);
        float num = 0.f, den = 0.f;
        for (int ww = 0; ww < 4; ++ww) {
            const float* src = smem + ((size_t)ww * MAXR + r) * ldw;
            const float a = (src[dh] == -INFINITY) ? 0.f : expf(src[dh] - mm);
            num = fmaf(src[d], a, num);
            den = fmaf(src[dh + 1], a, den);
        }
        out[(((long)b * R + r) * H + h) * dh + d] = from_f32<T>(num / den);
    }
}


// ---- bf16 streaming kernel for dh 64 / 128 -------------------------------------------------
// Same partition as above, tuned for HBM rate: U wave-loads of K and V in flight per wave and
// the next chunk prefetched under the arithmetic of the current one; q.k on v_dot2_f32_bf16
// (K is never widened), the in-row reduction on DPP adds (no LDS traffic), one online-softmax
// rescale per chunk instead of per key, p.v on packed fp32 FMAs.
__device__ __forceinline__ float dpp_add(float v, const int ctrl_tag) {
    // ctrl_tag: 0 = xor 1, 1 = xor 2 (quad permutes), 2 = row_half_mirror, 3 = row_mirror
    int y;
    const int x = __float_as_int(v);
    switch (ctrl_tag) {
        case 0: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;
        case 1: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;
        case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;
        default: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;
    }
    return v + __int_as_float(y);
}

template <int DH, int NW, int R>
__global__ __launch_bounds__(NW * 64, 4) void decode_attn_bf16_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
    const float* __restrict__ slopes, bf16_t* __restrict__ out, int H, int Lmax, int pos0_host,
    const int* __restrict__ pos_dev, float scale_log2e, const bf16_t* __restrict__ qkv_new) {
    // qkv_new != NULL (fused step): q, and the K/V rows of positions pos0.., come straight from the
    // projection output [B,R,3,H,DH]; this workgroup also files those K/V rows into the caches.
    constexpr int LPR = DH / 8;          // lanes per key row: 16 (dh 128) or 8 (dh 64)
    constexpr int RPL = 64 / LPR;        // key rows per wave-load
    constexpr int U = 4;                 // wave-loads per chunk
    constexpr float NEG = -1.0e30f;
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [NW][R][DH + 2]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x;
    const int sub = lane / LPR, ch = lane % LPR;
    int pos0 = pos_dev ? *pos_dev : pos0_host;
    if (pos0 + R > Lmax) pos0 = Lmax - R;
    const float slope2 = (slopes ? slopes[h] : 0.f) * LOG2E;

    bf16x2 qp[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bf16_t* qsrc = qkv_new ? qkv_new + (((long)b * R + r) * 3 * H + h) * DH : q + (((long)b * R + r) * H + h) * DH;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(qsrc + ch * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) qp[r][k] = bf16x2{t[2 * k], t[2 * k + 1]};
    }
    float m[R], l[R];
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = NEG;
        l[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][k] = f32x2{0.f, 0.f};
    }
    const bf16_t* kb = kc + ((long)b * H + h) * Lmax * DH + ch * 8;
    const bf16_t* vb = vc + ((long)b * H + h) * Lmax * DH + ch * 8;
    const int nkeys = pos0 + R;
    const int last = nkeys - 1;
    constexpr int STEP = NW * U * RPL;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

    // Fused step: keys 0..pos0-1 come from the cache, the R new keys straight from qkv_new (wave 0 takes
    // them as one extra chunk after its share of the cache) and are filed into the cache on the way.
    const bf16_t* nk = qkv_new ? qkv_new + (((long)b * R * 3 + 1) * H + h) * DH + ch * 8 : nullptr;   // K of new row 0
    const long nstep = 3L * H * DH;                      // to the next new row; V sits H*DH after K
    if (nk && w == 0) {
        for (int e = lane; e < R * LPR; e += 64) {
            const int r = e / LPR, c = e % LPR;
            const bf16_t* src = qkv_new + (((long)b * R + r) * 3 + 1) * H * DH + (long)h * DH + c * 8;
            const long dst = (((long)b * H + h) * Lmax + pos0 + r) * DH + c * 8;
            *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(kc) + dst) = *reinterpret_cast<const u32x4*>(src);
            *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(vc) + dst) = *reinterpret_cast<const u32x4*>(src + (long)H * DH);
        }
    }
    const int nkc = nk ? pos0 : nkeys;                   // keys served by the cache
    const int lastc = max(nkc - 1, 0);
    u32x4 kcur[U], vcur[U];
    auto load_cache = [&](int jb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(jb + u * RPL + sub, lastc);
            // non-temporal: every cache row is read once per step, and 3.3 GB of them per step would otherwise push the
            // 151 MB of weights out of the Infinity Cache (5.6 -> 6.4 TB/s at context 4096, step 0.896 -> 0.840 ms)
            kcur[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)j * DH));
            vcur[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)j * DH));
        }
    };
    auto load_new = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(u * RPL + sub, R - 1);
            kcur[u] = *reinterpret_cast<const u32x4*>(nk + r * nstep);
            vcur[u] = *reinterpret_cast<const u32x4*>(nk + r * nstep + (long)H * DH);
        }
    };
    int j0 = w * U * RPL;
    int lim = nkc - 1;                                   // keys j <= lim are real in the current pass
    bool tail = false, more = true;                      // wave-uniform
    if (j0 < nkc) load_cache(j0);
    else if (nk && w == 0) { load_new(); j0 = pos0; lim = last; tail = true; }
    else more = false;
    while (more) {
        float s[R][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * RPL + sub;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < R) {                 // uniform
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned kk = kcur[u][k];   // (bit_cast of a vector-element lvalue reads element 0)
                        a = __builtin_amdgcn_fdot2_f32_bf16(qp[r][k], __builtin_bit_cast(bf16x2, kk), a, false);
                    }
                    a = dpp_add(a, 0);
                    a = dpp_add(a, 1);
                    a = dpp_add(a, 2);
                    if (LPR == 16) a = dpp_add(a, 3);
                    const int i = pos0 + r;
                    a = fmaf(a, scale_log2e, slope2 * (float)(j - i));
                    s[r][u] = (j > i || j > lim) ? NEG : a;   // causal, and rows clamped past the end
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < R) {
                float mx = s[r][0];
#pragma unroll
                for (int u = 1; u < U; ++u) mx = fmaxf(mx, s[r][u]);
                const float mn = fmaxf(m[r], mx);
                const float alpha = __builtin_amdgcn_exp2f(m[r] - mn);
                m[r] = mn;
                float p[U], ps = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    p[u] = (s[r][u] <= NEG) ? 0.f : __builtin_amdgcn_exp2f(s[r][u] - mn);
                    ps += p[u];
                }
                l[r] = fmaf(l[r], alpha, ps);
                const f32x2 al = {alpha, alpha};
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[r][k] *= al;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f32x2 pp = {p[u], p[u]};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned x = vcur[u][k];
                        const f32x2 vv = {__uint_as_float(x << 16), __uint_as_float(x & 0xFFFF0000u)};
                        acc[r][k] = __builtin_elementwise_fma(pp, vv, acc[r][k]);
                    }
                }
            }
        }
        if (tail) break;
        j0 += STEP;
        if (j0 < nkc) load_cache(j0);
        else if (nk && w == 0) { load_new(); j0 = pos0; lim = last; tail = true; }
        else more = false;
    }
    // merge the RPL lane groups of this wave (base-2 domain)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < R) {
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                const float m2 = __shfl_xor(m[r], o, 64);
                const float l2 = __shfl_xor(l[r], o, 64);
                const float mn = fmaxf(m[r], m2);
                const float a1 = __builtin_amdgcn_exp2f(m[r] - mn);
                const float a2 = __builtin_amdgcn_exp2f(m2 - mn);
                l[r] = l[r] * a1 + l2 * a2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float o0 = __shfl_xor(acc[r][k][0], o, 64);
                    const float o1 = __shfl_xor(acc[r][k][1], o, 64);
                    acc[r][k][0] = acc[r][k][0] * a1 + o0 * a2;
                    acc[r][k][1] = acc[r][k][1] * a1 + o1 * a2;
                }
                m[r] = mn;
            }
        }
    }
    constexpr int ldw = DH + 2;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r < R && sub == 0) {
            float* dst = smem + ((size_t)w * R + r) * ldw;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dst[ch * 8 + 2 * k] = acc[r][k][0];
                dst[ch * 8 + 2 * k + 1] = acc[r][k][1];
            }
            if (ch == 0) {
                dst[DH] = m[r];
                dst[DH + 1] = l[r];
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < R * DH; e += NW * 64) {
        const int r = e / DH, d = e - r * DH;
        float mm = NEG;
        for (int ww = 0; ww < NW; ++ww) mm = fmaxf(mm, smem[((size_t)ww * R + r) * ldw + DH]);
        float num = 0.f, den = 0.f;
        for (int ww = 0; ww < NW; ++ww) {
            const float* src = smem + ((size_t)ww * R + r) * ldw;
            const float a = __builtin_amdgcn_exp2f(src[DH] - mm);
            num = fmaf(src[d], a, num);
            den = fmaf(src[DH + 1], a, den);
        }
        out[(((long)b * R + r) * H + h) * DH + d] = (bf16_t)(num / den);
    }
}

template <typename T>
__global__ void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc, int B, int R,
                                 int H, int dh, int Lmax, int pos0_host, const int* __restrict__ pos_dev) {
    int pos0 = pos_dev ? *pos_dev : pos0_host;
    if (pos0 + R > Lmax) pos0 = Lmax - R;
    const long n = (long)B * R * H * dh;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int d = t % dh;
    long u = t / dh;
    const int h = u % H;
    u /= H;
    const int r = u % R;
    const int b = u / R;
    const long src = (((long)b * R + r) * 3) * H * dh + (long)h * dh + d;
    const long dst = (((long)b * H + h) * Lmax + pos0 + r) * dh + d;
    kc[dst] = qkv[src + (long)H * dh];
    vc[dst] = qkv[src + 2L * H * dh];
}

}  // namespace

namespace {

int decode_attn_impl(const void* q, const void* kcache, const void* vcache, const float* slopes, void* out, int B, int R,
                     int H, int dh, int Lmax, int pos0, const int* pos_dev, int dtype, void* stream,
                     const void* qkv_new = nullptr) {
    MGX_REQUIRE((q || qkv_new) && kcache && vcache && out, MGX_EINVAL, "mgx_decode_attn: null pointer");
    MGX_REQUIRE(!qkv_new || (dtype == MGX_BF16 && (dh == 64 || dh == 128)), MGX_EUNSUPPORTED,
                "mgx_decode_step_attn: fused step needs bf16 and head dim 64 or 128 (got dtype %d, dh %d)", dtype, dh);
    MGX_REQUIRE(B > 0 && H > 0 && dh > 0 && R >= 1 && R <= MAXR, MGX_EINVAL, "mgx_decode_attn: B=%d R=%d H=%d dh=%d", B,
                R, H, dh);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_decode_attn: bad dtype %d", dtype);
    MGX_REQUIRE(R <= Lmax, MGX_EINVAL, "mgx_decode_attn: R=%d beyond Lmax=%d", R, Lmax);
    MGX_REQUIRE(pos_dev || (pos0 >= 0 && pos0 + R <= Lmax), MGX_EINVAL, "mgx_decode_attn: pos0=%d R=%d beyond Lmax=%d",
                pos0, R, Lmax);
    const int N = dtype == MGX_F32 ? 4 : 8;
    const int lpr = dh / N;
    MGX_REQUIRE(dh % N == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, MGX_EUNSUPPORTED,
                "mgx_decode_attn: head dim %d must be a power of two in [%d, %d]", dh, N, 64 * N);
    MGX_REQUIRE(B <= 65535, MGX_EUNSUPPORTED, "mgx_decode_attn: B too large");
    const float scale = 1.0f / sqrtf((float)dh);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(H, B);
    if (dtype == MGX_BF16 && (dh == 64 || dh == 128)) {
        // 512 workgroups of 8 waves fill the chip (2 per CU); smaller launches get 16 waves each
        const bool wide = (long)B * H < 512;
        const int nw = wide ? 16 : 8;
        const size_t smem = (size_t)nw * R * (dh + 2) * sizeof(float);
        const float sl2 = scale * 1.4426950408889634f;
#define MGX_DECODE_LAUNCH(DH_, NW_, R_)                                                                             \
    hipLaunchKernelGGL((decode_attn_bf16_kernel<DH_, NW_, R_>), grid, dim3(NW_ * 64), smem, st, (const bf16_t*)q,    \
                       (const bf16_t*)kcache, (const bf16_t*)vcache, slopes, (bf16_t*)out, H, Lmax, pos0, pos_dev,   \
                       sl2, (const bf16_t*)qkv_new)
#define MGX_DECODE_R(DH_, NW_)                      \
    switch (R) {                                    \
        case 1: MGX_DECODE_LAUNCH(DH_, NW_, 1); break; \
        case 2: MGX_DECODE_LAUNCH(DH_, NW_, 2); break; \
        case 3: MGX_DECODE_LAUNCH(DH_, NW_, 3); break; \
        default: MGX_DECODE_LAUNCH(DH_, NW_, 4); break; \
    }
        if (dh == 128) {
            if (wide) MGX_DECODE_R(128, 16) else MGX_DECODE_R(128, 8)
        } else {
            if (wide) MGX_DECODE_R(64, 16) else MGX_DECODE_R(64, 8)
        }
#undef MGX_DECODE_R
#undef MGX_DECODE_LAUNCH
        MGX_CHECK_LAUNCH("decode_attn_bf16");
        return MGX_OK;
    }
    const size_t smem = (size_t)4 * MAXR * (dh + 2) * sizeof(float);
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((decode_attn_kernel<float>), grid, dim3(256), smem, st, (const float*)q,
                           (const float*)kcache, (const float*)vcache, slopes, (float*)out, R, H, dh, Lmax, pos0,
                           pos_dev, scale);
    else
        hipLaunchKernelGGL((decode_attn_kernel<bf16_t>), grid, dim3(256), smem, st, (const bf16_t*)q,
                           (const bf16_t*)kcache, (const bf16_t*)vcache, slopes, (bf16_t*)out, R, H, dh, Lmax, pos0,
                           pos_dev, scale);
    MGX_CHECK_LAUNCH("decode_attn");
    return MGX_OK;
}

int kv_append_impl(const void* qkv, void* kcache, void* vcache, int B, int R, int H, int dh, int Lmax, int pos0,
                   const int* pos_dev, int dtype, void* stream) {
    MGX_REQUIRE(qkv && kcache && vcache, MGX_EINVAL, "mgx_kv_append: null pointer");
    MGX_REQUIRE(B > 0 && R > 0 && H > 0 && dh > 0 && R <= Lmax, MGX_EINVAL, "mgx_kv_append: bad shape");
    MGX_REQUIRE(pos_dev || (pos0 >= 0 && pos0 + R <= Lmax), MGX_EINVAL, "mgx_kv_append: pos0=%d R=%d beyond Lmax=%d",
                pos0, R, Lmax);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_kv_append: bad dtype %d", dtype);
    const long n = (long)B * R * H * dh;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((kv_append_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const float*)qkv, (float*)kcache, (float*)vcache, B, R, H, dh, Lmax, pos0, pos_dev);
    else
        hipLaunchKernelGGL((kv_append_kernel<bf16_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache, B, R, H, dh, Lmax, pos0, pos_dev);
    MGX_CHECK_LAUNCH("kv_append");
    return MGX_OK;
}

}  // namespace

extern "C" int mgx_decode_attn(const void* q, const void* kcache, const void* vcache, const float* slopes, void* out,
                               int B, int R, int H, int dh, int Lmax, int pos0, int dtype, void* stream) {
    return decode_attn_impl(q, kcache, vcache, slopes, out, B, R, H, dh, Lmax, pos0, nullptr, dtype, stream);
}

extern "C" int mgx_decode_attn_at(const void* q, const void* kcache, const void* vcache, const float* slopes, void* out,
                                  int B, int R, int H, int dh, int Lmax, const int* pos0_dev, int dtype, void* stream) {
    MGX_REQUIRE(pos0_dev, MGX_EINVAL, "mgx_decode_attn_at: null position pointer");
    return decode_attn_impl(q, kcache, vcache, slopes, out, B, R, H, dh, Lmax, 0, pos0_dev, dtype, stream);
}

extern "C" int mgx_decode_step_attn(const void* qkv, void* kcache, void* vcache, const float* slopes, void* out, int B,
                                    int R, int H, int dh, int Lmax, int pos0, const int* pos0_dev, int dtype,
                                    void* stream) {
    MGX_REQUIRE(qkv, MGX_EINVAL, "mgx_decode_step_attn: null pointer");
    return decode_attn_impl(nullptr, kcache, vcache, slopes, out, B, R, H, dh, Lmax, pos0, pos0_dev, dtype, stream, qkv);
}

extern "C" int mgx_kv_append(const void* qkv, void* kcache, void* vcache, int B, int R, int H, int dh, int Lmax,
                             int pos0, int dtype, void* stream) {
    return kv_append_impl(qkv, kcache, vcache, B, R, H, dh, Lmax, pos0, nullptr, dtype, stream);
}

extern "C" int mgx_kv_append_at(const void* qkv, void* kcache, void* vcache, int B, int R, int H, int dh, int Lmax,
                                const int* pos0_dev, int dtype, void* stream) {
    MGX_REQUIRE(pos0_dev, MGX_EINVAL, "mgx_kv_append_at: null position pointer");
    return kv_append_impl(qkv, kcache, vcache, B, R, H, dh, Lmax, 0, pos0_dev, dtype, stream);
}
