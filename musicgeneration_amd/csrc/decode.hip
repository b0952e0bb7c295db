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
                                                           int pos0, float scale) {
    constexpr int N = DVec<T>::N;
    typedef typename DVec<T>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][MAXR][dh + 2]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, h = blockIdx.x;
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

template <typename T>
__global__ void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc, int B, int R,
                                 int H, int dh, int Lmax, int pos0) {
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

extern "C" int mgx_decode_attn(const void* q, const void* kcache, const void* vcache, const float* slopes, void* out,
                               int B, int R, int H, int dh, int Lmax, int pos0, int dtype, void* stream) {
    MGX_REQUIRE(q && kcache && vcache && out, MGX_EINVAL, "mgx_decode_attn: null pointer");
    MGX_REQUIRE(B > 0 && H > 0 && dh > 0 && R >= 1 && R <= MAXR, MGX_EINVAL, "mgx_decode_attn: B=%d R=%d H=%d dh=%d", B,
                R, H, dh);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_decode_attn: bad dtype %d", dtype);
    MGX_REQUIRE(pos0 >= 0 && pos0 + R <= Lmax, MGX_EINVAL, "mgx_decode_attn: pos0=%d R=%d beyond Lmax=%d", pos0, R, Lmax);
    const int N = dtype == MGX_F32 ? 4 : 8;
    const int lpr = dh / N;
    MGX_REQUIRE(dh % N == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, MGX_EUNSUPPORTED,
                "mgx_decode_attn: head dim %d must be a power of two in [%d, %d]", dh, N, 64 * N);
    MGX_REQUIRE(B <= 65535, MGX_EUNSUPPORTED, "mgx_decode_attn: B too large");
    const float scale = 1.0f / sqrtf((float)dh);
    const size_t smem = (size_t)4 * MAXR * (dh + 2) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(H, B);
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((decode_attn_kernel<float>), grid, dim3(256), smem, st, (const float*)q,
                           (const float*)kcache, (const float*)vcache, slopes, (float*)out, R, H, dh, Lmax, pos0, scale);
    else
        hipLaunchKernelGGL((decode_attn_kernel<bf16_t>), grid, dim3(256), smem, st, (const bf16_t*)q,
                           (const bf16_t*)kcache, (const bf16_t*)vcache, slopes, (bf16_t*)out, R, H, dh, Lmax, pos0,
                           scale);
    MGX_CHECK_LAUNCH("decode_attn");
    return MGX_OK;
}

extern "C" int mgx_kv_append(const void* qkv, void* kcache, void* vcache, int B, int R, int H, int dh, int Lmax,
                             int pos0, int dtype, void* stream) {
    MGX_REQUIRE(qkv && kcache && vcache, MGX_EINVAL, "mgx_kv_append: null pointer");
    MGX_REQUIRE(B > 0 && R > 0 && H > 0 && dh > 0, MGX_EINVAL, "mgx_kv_append: bad shape");
    MGX_REQUIRE(pos0 >= 0 && pos0 + R <= Lmax, MGX_EINVAL, "mgx_kv_append: pos0=%d R=%d beyond Lmax=%d", pos0, R, Lmax);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_kv_append: bad dtype %d", dtype);
    const long n = (long)B * R * H * dh;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((kv_append_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const float*)qkv, (float*)kcache, (float*)vcache, B, R, H, dh, Lmax, pos0);
    else
        hipLaunchKernelGGL((kv_append_kernel<bf16_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache, B, R, H, dh, Lmax, pos0);
    MGX_CHECK_LAUNCH("kv_append");
    return MGX_OK;
}
