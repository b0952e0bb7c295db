// bf16 MFMA flash attention for gfx950 (head dim 64 or 128), causal + ALiBi.
// Replaces SimpleAttention.multihead_attn (model_baseline/alibi_unified.py:40-54)
// and the [H,S,S] mask buffer (alibi_unified.py:26-31): the bias slope_h*(j-i) is
// folded into the initial MFMA accumulator, nothing S x S ever reaches HBM.
//
// Orientation ("query on the lane"): every product is arranged so that the
// v_mfma_f32_32x32x16_bf16 result has the QUERY index on the lane (column) and
// the reduced index of the NEXT product in its 16 registers (rows):
//     S^T[j][i]  = K . Q^T        A = K rows (LDS, ds_read_b128), B = Q (registers)
//     O^T[d][i] += V^T . P^T      A = V columns (LDS, ds_read_b64_tr_b16), B = P^T straight
//                                 from the S^T accumulators (no LDS round trip)
// so the softmax row statistics (one query per lane) need a single half-wave
// exchange per tile, and the rescale factor is a per-lane scalar.
//
// Workgroup = 4 waves x 32 queries = 128 queries; K/V tiles of 64 keys are
// staged global -> registers -> LDS with the loads of tile t+1 in flight under
// the MFMAs of tile t.  LDS image: 8-row x 32-column sub-tiles (512 B) with an
// XOR on the 16-byte chunk index, conflict-free for both the row reads and the
// transposed reads (cdna_hip_programming.md T10 image (a)).
#include <type_traits>
#include "attn_mfma_common.h"

// attn_bwd_pipe.hip
long mgx_attn_bwd_pipe_stats_floats(int B, int S, int H);
// attn_bwd_dq_pipe.hip
int mgx_attn_bwd_dq_pipe(const void* dout, const void* qkv, const float* lse, const float* nd32, const float* slopes,
                         void* dqkv, int B, int S, int H, int dh, hipStream_t st);
int mgx_attn_bwd_dkv_pipe(const void* dout, const void* qkv, const void* out, const float* lse, const float* slopes,
                          void* dqkv, float* stats, int B, int S, int H, int dh, hipStream_t st);

namespace {

// ===========================================================================
// forward
// ===========================================================================
constexpr int FWD_GROUP = 8;        // (batch, head) pairs whose query blocks an XCD runs together

template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                                const float* __restrict__ slopes,
                                                                bf16_t* __restrict__ out, float* __restrict__ lse,
                                                                int S, int H, float c /* scale*log2e */, int npairs) {
    constexpr int NT = 64 * NW, QB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[4 * TB];      // 2 buffers x (K image, V image)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    // 1-D grid; workgroup n runs on XCD n & 7 (round-robin dispatch).  An XCD walks its (batch, head) pairs in groups
    // of FWD_GROUP: inside a group the query blocks heaviest first, the group's pairs fastest - so the blocks in flight
    // on an XCD (64 workgroups) belong to 8 pairs whose K/V (1 MB each at S 2048, dh 128) they share through its L2
    // (the (H, B, q-block) grid of rounds 1-2 had 16 pairs x 4 blocks in flight: 205 -> 200 us), and the dispatcher's
    // first-free-slot order still ends every slot at the same time (list-scheduling the 4..32-tile blocks of 16 pairs
    // on 64 slots in this order: makespan 36 tiles = the ideal; pair-major order: 56).
    const int nqb = (S + QB - 1) / QB;
    const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3;
    const int grp = sl / (FWD_GROUP * nqb), r_ = sl % (FWD_GROUP * nqb);
    const int qb = nqb - 1 - r_ / FWD_GROUP;
    const int pair = 8 * (grp * FWD_GROUP + r_ % FWD_GROUP) + xcd;
    if (pair >= npairs) return;                          // whole workgroups; before any barrier
    const int b = pair / H, h = pair % H;
    const int q0 = qb * QB, qw0 = q0 + 32 * w;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const LaneAddr<DH> la(lane);

    const int qi = qw0 + l31;
    const bool qok = qi < S;
    bf16x8 qf[DH / 16];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = qok ? *reinterpret_cast<const bf16x8*>(base + (long)qi * rs + ks * 16 + 8 * h5) : z;
    }
    const float sb = slopes[h] * LOG2E / c;              // bias per key step, in raw-score units

    float m = -INFINITY, l = 0.f;
    f32x16 o[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

    const int kmax = min(q0 + QB, S);                    // keys [0, kmax)
    const int ntiles = (kmax + 63) / 64;
    // Keys are walked from the diagonal tile BACKWARDS: with ALiBi the largest scores sit next to the diagonal, so the
    // running maximum is right after the first tile and the rescale below stays the exception (walking forwards, the
    // heads with large slopes raise it by more than 2^8 on every tile).
    // Double-buffered staging, one barrier per tile: while step n is consumed from buffer n&1, the tile of step n+1
    // (already in registers) is written to the other buffer and the tile of step n+2 is fetched.
    TileRegs<DH, NT> kr, vr;
    tile_load<DH, NT>(kr, kp, rs, (ntiles - 1) * 64, S, tid);
    tile_load<DH, NT>(vr, vp, rs, (ntiles - 1) * 64, S, tid);
    tile_store<DH, NT>(kr, smem, tid);
    tile_store<DH, NT>(vr, smem + TB, tid);
    if (ntiles > 1) {
        tile_load<DH, NT>(kr, kp, rs, (ntiles - 2) * 64, S, tid);
        tile_load<DH, NT>(vr, vp, rs, (ntiles - 2) * 64, S, tid);
    }
    __syncthreads();
    for (int n = 0; n < ntiles; ++n) {
        const int j0 = (ntiles - 1 - n) * 64;
        const char* Ks = smem + (n & 1) * 2 * TB;
        const char* Vs = Ks + TB;
        if (n + 1 < ntiles) {
            char* Kn = smem + ((n + 1) & 1) * 2 * TB;
            tile_store<DH, NT>(kr, Kn, tid);
            tile_store<DH, NT>(vr, Kn + TB, tid);
        }
        if (n + 2 < ntiles) {
            tile_load<DH, NT>(kr, kp, rs, j0 - 128, S, tid);
            tile_load<DH, NT>(vr, vp, rs, j0 - 128, S, tid);
        }
        if (j0 <= qw0 + 31) {                             // wave-uniform
            f32x16 s[2];
            const float lb = sb * (float)(j0 + 4 * h5 - qi);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[tb][r] = lb + sb * (float)(32 * tb + (r & 3) + 8 * (r >> 2));
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(tb, ks)), qf[ks],
                                                                   s[tb], 0, 0, 0);
            if (j0 + 63 > qw0) {                          // diagonal tile: causal mask
                const int dj = qi - j0 - 4 * h5;          // key offset (r-part) must be <= dj
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * tb + (r & 3) + 8 * (r >> 2) > dj) s[tb][r] = -INFINITY;
            }
            float mx = s[0][0];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tb][r]);
            mx = half_swap_max(mx);
            // Deferred rescale (T13): keep the stale running max while no row's max grew by more
            // than 2^RESCALE_LOG2; P is then bounded by 2^RESCALE_LOG2 instead of 1 (fp32 sums,
            // bf16 P keep their relative precision).  The decision is wave-uniform and taken
            // before this tile's P is formed, with the previous tile's P.V already accumulated.
            if (!__all((mx - m) * c <= RESCALE_LOG2)) {
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2((m - mn) * c);
                m = mn;
                l *= alpha;
#pragma unroll
                for (int d = 0; d < DH / 32; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            const float mc = m * c;
            float psum = 0.f;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(fmaf(s[tb][r], c, -mc));
                    s[tb][r] = p;
                    psum += p;
                }
            l += psum;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 pf = pack8(s[tb], 8 * s2);
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d) {
                        const bf16x8 vf = lds_read_tr_pair(Vs + la.tr_read(tb, s2, 0, d), Vs + la.tr_read(tb, s2, 1, d));
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }
    const float lt = half_swap_sum(l);
    const float inv = 1.f / lt;
    if (qok) {
        bf16_t* op = out + ((long)b * S + qi) * H * DH + (long)h * DH;
#pragma unroll
        for (int d = 0; d < DH / 32; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[d][4 * g + k] * inv);
                *reinterpret_cast<bf16x4*>(op + 32 * d + 8 * g + 4 * h5) = v;
            }
        if (h5 == 0) lse[((long)b * H + h) * S + qi] = (m * c + log2f(lt)) * LN2;
    }
}


// ===========================================================================
// backward: statistics pre-pass + dK/dV kernel (attn_bwd_pipe.hip), then the dQ kernel (attn_bwd_dq_pipe.hip)
// ===========================================================================
int launch_bwd_mfma(const void* dout, const void* qkv, const void* out, const float* lse, const float* slopes,
                    void* dqkv, float* ws, int B, int S, int H, int dh, hipStream_t st) {
    const int Sp = (S + 63) / 64 * 64;
    const int rc = mgx_attn_bwd_dkv_pipe(dout, qkv, out, lse, slopes, dqkv, ws, B, S, H, dh, st);
    if (rc != MGX_OK) return rc;
    const float* nd = ws + 8L * B * H * Sp;          // workspace: nla | nda (16 B per row each) | nd32
    return mgx_attn_bwd_dq_pipe(dout, qkv, lse, nd, slopes, dqkv, B, S, H, dh, st);
}

}  // namespace

// attn_rel_mfma.hip
int mgx_rel_fwd_mfma(const void* qkv, const void* E, void* out, float* lse, int B, int S, int H, int dh, int max_seq,
                     hipStream_t st);
int mgx_rel_bwd_mfma(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                     float* dE, float* delta, int B, int S, int H, int dh, int max_seq, hipStream_t st);
bool mgx_rel_bwd_mfma_ready();

bool mgx_attn_mfma_supported(int mode, int dh, int dtype, int S, int H, int max_seq, bool backward) {
    // 32-bit byte offsets inside one batch element: S * (3*H*dh) * 2 bytes must stay below 2^31
    if (!(dtype == MGX_BF16 && (dh == 64 || dh == 128) && (long)S * 3 * H * dh * 2 < (1L << 31))) return false;
    if (mode == 0) return true;
    // skew-relative mode: E rows are fetched as 16-byte pieces of whole rows (any max_seq would do since round 3, when the
    // transposed copy of E went away; the bound is kept as the tested envelope, other tables take the generic kernels)
    return mode == 1 && max_seq % 8 == 0 && max_seq >= 32 && (!backward || mgx_rel_bwd_mfma_ready());
}

int mgx_attn_fwd_mfma(int mode, const void* qkv, const float* slopes, const void* E, void* out, float* lse, int B,
                      int S, int H, int dh, int max_seq, hipStream_t st) {
    if (mode == 1) return mgx_rel_fwd_mfma(qkv, E, out, lse, B, S, H, dh, max_seq, st);
    MGX_REQUIRE(mode == 0, MGX_EUNSUPPORTED, "mgx_attn_fwd_mfma: mode %d", mode);
    const float c = LOG2E / sqrtf((float)dh);
    constexpr int NW = 8;                                // waves (x 32 queries) per workgroup
    const int nqb = (S + 32 * NW - 1) / (32 * NW);
    const long ppx = ((B * H + 7) / 8 + FWD_GROUP - 1) / FWD_GROUP * FWD_GROUP;       // pair slots per XCD
    MGX_REQUIRE(8 * ppx * nqb < (1L << 31), MGX_EUNSUPPORTED, "mgx_attn_fwd_mfma: too many workgroups");
    dim3 grid((unsigned)(8 * ppx * nqb), 1, 1);
    if (dh == 64)
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<64, NW>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv, slopes,
                           (bf16_t*)out, lse, S, H, c, B * H);
    else
        hipLaunchKernelGGL((attn_fwd_mfma_kernel<128, NW>), grid, dim3(64 * NW), 0, st, (const bf16_t*)qkv, slopes,
                           (bf16_t*)out, lse, S, H, c, B * H);
    MGX_CHECK_LAUNCH("attn_fwd_mfma");
    return MGX_OK;
}

int mgx_attn_bwd_mfma(int mode, const void* dout, const void* qkv, const void* out, const float* lse,
                      const float* slopes, const void* E, void* dqkv, float* dE, float* delta, int B, int S, int H,
                      int dh, int max_seq, hipStream_t st) {
    if (mode == 1) return mgx_rel_bwd_mfma(dout, qkv, out, lse, E, dqkv, dE, delta, B, S, H, dh, max_seq, st);
    MGX_REQUIRE(mode == 0, MGX_EUNSUPPORTED, "mgx_attn_bwd_mfma: mode %d", mode);
    return launch_bwd_mfma(dout, qkv, out, lse, slopes, dqkv, delta, B, S, H, dh, st);
}
