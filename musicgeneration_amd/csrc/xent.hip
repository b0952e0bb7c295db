// Fused softmax cross-entropy over the MIDI vocabulary (V = 618, or 92/130/203
// for the separated heads): replaces F.cross_entropy(logits.transpose(-1,-2), y,
// weight, ignore_index) — alibi_unified.py:161, alibi_separated.py:193-195 —
// i.e. the transpose copy + log_softmax + nll_loss2d of the reference.
//
// One wave per token row: the V scores are read once (coalesced), max / sum-exp
// by wavefront reduction, fp32 throughout.  HBM-bound: V*sizeof(T) bytes per row
// forward, 2*V*sizeof(T) backward.  The mean uses the weighted normaliser
// sum(w[y]) exactly as torch does.
#include "mgx_common.h"

namespace {

constexpr int XENT_PARTIALS = 2048;   // forward blocks (4 rows in flight each): 8 per CU; 256 left one wave per SIMD (155 us at T = 32768)

template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* __restrict__ logits, const long long* __restrict__ tgt,
                                                        const float* __restrict__ weight, float* __restrict__ row_loss,
                                                        float* __restrict__ row_lse, float* __restrict__ partial,
                                                        long T_rows, int V, int ld, int col0, long long ignore_index) {
    __shared__ float s_loss[4], s_w[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float loss_acc = 0.f, w_acc = 0.f;
    for (long row = (long)blockIdx.x * 4 + w; row < T_rows; row += (long)gridDim.x * 4) {
        const T* x = logits + row * ld + col0;
        float m = -INFINITY;
        for (int c = lane; c < V; c += 64) m = fmaxf(m, to_f32(x[c]));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += expf(to_f32(x[c]) - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        const long long y = tgt[row];
        float wl = 0.f, wy = 0.f;
        if (y != ignore_index && y >= 0 && y < V) {
            wy = weight ? weight[y] : 1.f;
            wl = wy * (lse - to_f32(x[y]));
        }
        if (lane == 0) {
            row_lse[row] = lse;
            row_loss[row] = wl;
        }
        loss_acc += wl;
        w_acc += wy;
    }
    if (lane == 0) {
        s_loss[w] = loss_acc;
        s_w[w] = w_acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
        partial[2 * blockIdx.x + 1] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

__global__ void xent_sum_kernel(const float* __restrict__ partial, float* __restrict__ sums, int n) {
    // one wave, fixed order -> deterministic
    const int lane = threadIdx.x;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < n; i += 64) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        sums[0] = a;
        sums[1] = b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const T* __restrict__ logits, const long long* __restrict__ tgt,
                                                        const float* __restrict__ weight,
                                                        const float* __restrict__ row_lse,
                                                        const float* __restrict__ sums, const float* __restrict__ gscale,
                                                        T* __restrict__ dlogits, long T_rows, int V, int ld, int col0,
                                                        long long ignore_index) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float g = gscale[0] / sums[1];
    for (long row = (long)blockIdx.x * 4 + w; row < T_rows; row += (long)gridDim.x * 4) {
        const T* x = logits + row * ld + col0;
        T* dx = dlogits + row * ld + col0;
        const long long y = tgt[row];
        float wy = 0.f;
        if (y != ignore_index && y >= 0 && y < V) wy = weight ? weight[y] : 1.f;
        const float coef = g * wy;
        const float lse = row_lse[row];
        for (int c = lane; c < V; c += 64) {
            float p = expf(to_f32(x[c]) - lse);
            if (c == y) p -= 1.f;
            dx[c] = from_f32<T>(coef * p);
        }
    }
}

}  // namespace

extern "C" int mgx_xent_sums_len(void) { return 2 + 2 * XENT_PARTIALS; }

extern "C" int mgx_softmax_xent_fwd(const void* logits, const long long* targets, const float* weight,
                                    float* row_loss, float* row_lse, float* sums, long T, int V, int ld, int col0,
                                    long long ignore_index, int dtype, void* stream) {
    MGX_REQUIRE(logits && targets && row_loss && row_lse && sums, MGX_EINVAL, "mgx_softmax_xent_fwd: null pointer");
    MGX_REQUIRE(T > 0 && V > 0 && col0 >= 0 && col0 + V <= ld, MGX_EINVAL,
                "mgx_softmax_xent_fwd: T=%ld V=%d ld=%d col0=%d", T, V, ld, col0);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_softmax_xent_fwd: bad dtype %d", dtype);
    const int nblk = (int)((T + 3) / 4 < XENT_PARTIALS ? (T + 3) / 4 : XENT_PARTIALS);
    // two-stage deterministic reduction: per-block partials go to the tail of `sums`
    float* partial = sums + 2;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((xent_fwd_kernel<float>), dim3(nblk), dim3(256), 0, st, (const float*)logits, targets,
                           weight, row_loss, row_lse, partial, T, V, ld, col0, ignore_index);
    else
        hipLaunchKernelGGL((xent_fwd_kernel<bf16_t>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)logits, targets,
                           weight, row_loss, row_lse, partial, T, V, ld, col0, ignore_index);
    MGX_CHECK_LAUNCH("xent_fwd");
    hipLaunchKernelGGL(xent_sum_kernel, dim3(1), dim3(64), 0, st, partial, sums, nblk);
    MGX_CHECK_LAUNCH("xent_sum");
    return MGX_OK;
}

extern "C" int mgx_softmax_xent_bwd(const void* logits, const long long* targets, const float* weight,
                                    const float* row_lse, const float* sums, const float* gscale, void* dlogits,
                                    long T, int V, int ld, int col0, long long ignore_index, int dtype,
                                    void* stream) {
    MGX_REQUIRE(logits && targets && row_lse && sums && gscale && dlogits, MGX_EINVAL,
                "mgx_softmax_xent_bwd: null pointer");
    MGX_REQUIRE(T > 0 && V > 0 && col0 >= 0 && col0 + V <= ld, MGX_EINVAL,
                "mgx_softmax_xent_bwd: T=%ld V=%d ld=%d col0=%d", T, V, ld, col0);
    MGX_REQUIRE(dtype == MGX_F32 || dtype == MGX_BF16, MGX_EINVAL, "mgx_softmax_xent_bwd: bad dtype %d", dtype);
    long nblk = (T + 3) / 4;
    if (nblk > 4096) nblk = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGX_F32)
        hipLaunchKernelGGL((xent_bwd_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st, (const float*)logits,
                           targets, weight, row_lse, sums, gscale, (float*)dlogits, T, V, ld, col0, ignore_index);
    else
        hipLaunchKernelGGL((xent_bwd_kernel<bf16_t>), dim3((unsigned)nblk), dim3(256), 0, st, (const bf16_t*)logits,
                           targets, weight, row_lse, sums, gscale, (bf16_t*)dlogits, T, V, ld, col0, ignore_index);
    MGX_CHECK_LAUNCH("xent_bwd");
    return MGX_OK;
}
