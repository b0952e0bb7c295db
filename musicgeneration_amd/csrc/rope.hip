// Rotary position embedding applied in place to the q and k thirds of the packed projection
// qkv [B,S,3,H,dh] (old_models/rotary_transformer.py:9-21: rotate_every_two + apply_rotary_pos_emb):
//   x'[2i]   = x[2i] * cos_i(s) - x[2i+1] * sin_i(s)
//   x'[2i+1] = x[2i+1] * cos_i(s) + x[2i] * sin_i(s)
// with cos/sin tables [S, dh/2] fp32 built on the host exactly as FixedPositionalEmbedding does
// (rotary_transformer.py:24-42).  inverse != 0 applies the transposed rotation (sin negated): the
// backward of the forward application, used on dq | dk.  HBM-bound: 2/3 of qkv read and written once.
#include "mgx_common.h"

namespace {

template <typename T, int N>
struct RVec;
template <> struct RVec<float, 4> { typedef f32x4 type; };
template <> struct RVec<bf16_t, 8> { typedef bf16x8 type; };

template <typename T, int N>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qkv, const float* __restrict__ cs,
                                                    const float* __restrict__ sn, long rows, int S, int H, int dh,
                                                    float sign) {
    // one thread = one N-element vector of the q|k part of a row: 2*H*dh/N vectors per row
    const int vpr = 2 * H * dh / N;
    const long total = rows * vpr;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / vpr;
        const int v = (int)(e - row * vpr);
        const int col = v * N;                       // element inside [q | k] of this row
        const int d0 = col % dh;                     // first head-dim index of the vector (even)
        const int s = (int)(row % S);
        T* p = qkv + row * (3L * H * dh) + col;
        typename RVec<T, N>::type x = *reinterpret_cast<typename RVec<T, N>::type*>(p);
        const float* c = cs + (long)s * (dh / 2) + d0 / 2;
        const float* si = sn + (long)s * (dh / 2) + d0 / 2;
#pragma unroll
        for (int k = 0; k < N; k += 2) {
            const float a = to_f32(x[k]), b = to_f32(x[k + 1]);
            const float cc = c[k / 2], ss = si[k / 2] * sign;
            x[k] = from_f32<T>(a * cc - b * ss);
            x[k + 1] = from_f32<T>(b * cc + a * ss);
        }
        *reinterpret_cast<typename RVec<T, N>::type*>(p) = x;
    }
}

}  // namespace

extern "C" int mgx_rope_qk(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int H, int dh,
                           int inverse, int dtype, void* stream) {
    MGX_REQUIRE(qkv && cos_tab && sin_tab, MGX_EINVAL, "mgx_rope_qk: null pointer");
    MGX_REQUIRE(B > 0 && S > 0 && H > 0 && dh > 0, MGX_EINVAL, "mgx_rope_qk: B=%d S=%d H=%d dh=%d", B, S, H, dh);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_rope_qk: bad dtype %d", dtype);
    const int N = dtype == MGX_F32 ? 4 : 8;
    MGX_REQUIRE(dh % N == 0, MGX_EUNSUPPORTED, "mgx_rope_qk: dh=%d must be a multiple of %d", dh, N);
    const long rows = (long)B * S;
    const long total = rows * (2L * H * dh / N);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const float sign = inverse ? -1.f : 1.f;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((rope_kernel<float, 4>), dim3((unsigned)blocks), dim3(256), 0, st, (float*)qkv, cos_tab, sin_tab,
                           rows, S, H, dh, sign);
    else
        hipLaunchKernelGGL((rope_kernel<bf16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, st, (bf16_t*)qkv, cos_tab,
                           sin_tab, rows, S, H, dh, sign);
    MGX_CHECK_LAUNCH("rope_qk");
    return MGX_OK;
}
