// bf16 MFMA flash attention with the skew relative-position term (head dim 64 or 128).
// Replaces RelativeAttention.relative_attn + multihead_attn
// (old_models/relative_transformer.py:18-44):
//     w[i,j] = (q_i.k_j + q_i.E[h, M-1-(i-j)]) * dh^-1/2,  causal softmax, @ v
// The reference materialises rel = q @ E^T as a [B,H,S,S] tensor and "skews" it with a
// pad/reshape.  Here the term is computed per tile: with Er[r] = E[h, M-1-r] (r = i-j, the
// distance), a 32-distance block G[r,i] = Er[r].q_i is one more MFMA product per key block (a
// block of G serves two neighbouring key blocks), and the skew — the move from (distance, query)
// to (key, query) coordinates — is a round trip through a per-wave LDS ring whose addressing
// makes one side of the trip a 16-byte aligned access:
//     ring[i_l][p],  p = (r - i_l - 1) mod 64          (i_l = query within the wave's 32)
// A reader in key coordinates (j = i - r) finds p = (R0 - 1 - j_l) mod 64, independent of i_l.
// The ring then receives dS in place (backward), is read back in distance coordinates, and
// feeds dQ += dG.Er.  Nothing S x S reaches HBM.
//
// Orientation and staging are those of attn_mfma.hip (query on the lane for forward and dQ, key
// on the lane for dK/dV).  dE has its own kernel (distance on the lane): a workgroup owns a band
// of distances and walks down the diagonal band of the score matrix, so the dE accumulators stay
// in registers for the whole launch: no atomics, bitwise reproducible.
#include "attn_mfma_common.h"

namespace {

constexpr int REL_NW = 4;        // waves (x 32 rows) per workgroup
constexpr int REL_DE_NW = 2;     // dE kernel: 64 distances per band x 2 query-half groups = 4 waves; bands paired
constexpr int SKS = 68;          // ring row stride in floats (272 B: 16-byte reads of 16 consecutive lanes spread over all banks)

// ---- the skew ring (query-on-lane kernels) ----------------------------------------------------
// G layout: accumulator of Er_blk . Q^T : lane = (query i_l, half h5), register e <-> distance
//           Rb + 8*(e>>2) + 4*h5 + (e&3)
// S layout: accumulator of K_blk . Q^T  : register e <-> key jb + 8*(e>>2) + 4*h5 + (e&3)
__device__ __forceinline__ void ring_put_g(float* ring, const f32x16& g, int Rb, int l31, int h5) {
    float* row = ring + l31 * SKS;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = Rb + 8 * (e >> 2) + 4 * h5 + (e & 3);
        row[(r - l31 - 1) & 63] = g[e];
    }
}
__device__ __forceinline__ void ring_get_s(const float* ring, f32x16& s, int R0, int l31, int h5) {
    const float* row = ring + l31 * SKS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + ((R0 - 8 * g - 4 * h5 - 4) & 63));
        s[4 * g + 0] = v[3];
        s[4 * g + 1] = v[2];
        s[4 * g + 2] = v[1];
        s[4 * g + 3] = v[0];
    }
}
__device__ __forceinline__ void ring_put_s(float* ring, const f32x16& s, int R0, int l31, int h5) {
    float* row = ring + l31 * SKS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[3] = s[4 * g + 0];
        v[2] = s[4 * g + 1];
        v[1] = s[4 * g + 2];
        v[0] = s[4 * g + 3];
        *reinterpret_cast<f32x4*>(row + ((R0 - 8 * g - 4 * h5 - 4) & 63)) = v;
    }
}

// Er rows Rb + l31 as an MFMA operand (row on the lane, 8 consecutive d per lane and k-step);
// Er[r] = E[h, M-1-r]; rows outside [0, M) are clamped (their products are never used).
template <int DH>
__device__ __forceinline__ void load_er_rows(bf16x8 (&ef)[DH / 16], const bf16_t* Eh, int M, int Rb, int l31, int h5) {
    int idx = M - 1 - (Rb + l31);
    idx = min(max(idx, 0), M - 1);
    const bf16_t* p = Eh + (long)idx * DH + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) ef[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// ===========================================================================
// forward
// ===========================================================================
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void rel_fwd_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                               const bf16_t* __restrict__ E,
                                                               bf16_t* __restrict__ out, float* __restrict__ lse,
                                                               int S, int H, int M, float c /* scale*log2e */) {
    constexpr int NT = 64 * NW, QB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 x (K image, V image), then NW rings
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    float* ring = reinterpret_cast<float*>(smem + 4 * TB) + w * 32 * SKS;
    const int qb = gridDim.z - 1 - blockIdx.z;
    const int q0 = qb * QB, qw0 = q0 + 32 * w;
    const int b = blockIdx.y, h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const bf16_t* Eh = E + (long)h * M * DH;
    const LaneAddr<DH> la(lane);

    const int qi = qw0 + l31;
    const bool qok = qi < S;
    bf16x8 qf[DH / 16];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = qok ? *reinterpret_cast<const bf16x8*>(base + (long)qi * rs + ks * 16 + 8 * h5) : z;
    }
    float m = -INFINITY, l = 0.f;
    f32x16 o[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;

    // distance blocks are consumed in descending order qw0, qw0-32, ..., 0: one block ahead in registers
    bf16x8 ef[DH / 16];
    auto g_block = [&](int Rb) {                           // G[Rb..Rb+32) -> ring; prefetch the next block's Er rows
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef[ks], qf[ks], g, 0, 0, 0);
        if (Rb >= 32) load_er_rows<DH>(ef, Eh, M, Rb - 32, l31, h5);
        ring_put_g(ring, g, Rb, l31, h5);
    };
    if (qw0 < S) {
        load_er_rows<DH>(ef, Eh, M, qw0, l31, h5);
        g_block(qw0);
    }

    const int kmax = min(q0 + QB, S);
    const int ntiles = (kmax + 63) / 64;
    TileRegs<DH, NT> kr, vr;
    tile_load<DH, NT>(kr, kp, rs, 0, S, tid);
    tile_load<DH, NT>(vr, vp, rs, 0, S, tid);
    tile_store<DH, NT>(kr, smem, tid);
    tile_store<DH, NT>(vr, smem + TB, tid);
    if (ntiles > 1) {
        tile_load<DH, NT>(kr, kp, rs, 64, S, tid);
        tile_load<DH, NT>(vr, vp, rs, 64, S, tid);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        const char* Ks = smem + (t & 1) * 2 * TB;
        const char* Vs = Ks + TB;
        if (t + 1 < ntiles) {
            char* Kn = smem + ((t + 1) & 1) * 2 * TB;
            tile_store<DH, NT>(kr, Kn, tid);
            tile_store<DH, NT>(vr, Kn + TB, tid);
        }
        if (t + 2 < ntiles) {
            tile_load<DH, NT>(kr, kp, rs, j0 + 128, S, tid);
            tile_load<DH, NT>(vr, vp, rs, j0 + 128, S, tid);
        }
        if (j0 <= qw0 + 31 && qw0 < S) {                  // wave-uniform
            f32x16 s[2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int R0 = qw0 - j0 - 32 * tb;        // distance of (query qw0, key block start)
                if (R0 >= 0) {
                    if (R0 >= 32) g_block(R0 - 32);
                    ring_get_s(ring, s[tb], R0, l31, h5);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[tb][e] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks)
                    s[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(tb, ks)), qf[ks],
                                                                   s[tb], 0, 0, 0);
            }
            if (j0 + 63 > qw0) {                          // diagonal tile: causal mask
                const int dj = qi - j0 - 4 * h5;
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
// backward
// ===========================================================================
// ErT[h][d][r] = Er[r][d] = E[h, M-1-r, d] for r < M, zero up to Mp = round_up(M, 32): the operand of
// dQ += dG . Er needs Er with the distance as the contiguous (reduced) index.
__global__ __launch_bounds__(256) void rel_ert_kernel(const bf16_t* __restrict__ E, bf16_t* __restrict__ ErT, int H,
                                                       int M, int Mp, int DH) {
    const long n = (long)H * DH * Mp;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int r = (int)(t % Mp);
    const long u = t / Mp;
    const int d = (int)(u % DH);
    const int h = (int)(u / DH);
    ErT[t] = r < M ? E[((long)h * M + (M - 1 - r)) * DH + d] : (bf16_t)0.f;
}

// dQ: query on the lane.  Per 32-key block (distance R0 = qw0 - jb of its first key):
//   S^T = K.Q^T + skew(G)   dP^T = V.dO^T - delta   dS^T = P^T*dP^T*scale   dQ^T += K^T.dS^T
//   ring <- dS^T (in place of the consumed G entries); block R0 of dG is then complete:
//   dQ^T += ErT[:, R0..R0+32) . dG^T
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void rel_bwd_dq_mfma_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
    const float* __restrict__ delta, const bf16_t* __restrict__ E, const bf16_t* __restrict__ ErT,
    bf16_t* __restrict__ dqkv, int S, int H, int M, int Mp, float c, float scale) {
    constexpr int NT = 64 * NW, QB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    float* ring = reinterpret_cast<float*>(smem + 4 * TB) + w * 32 * SKS;
    const int qb = gridDim.z - 1 - blockIdx.z;
    const int q0 = qb * QB, qw0 = q0 + 32 * w;
    const int b = blockIdx.y, h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const bf16_t* gp = dout + (long)b * S * H * DH + (long)h * DH;
    const bf16_t* Eh = E + (long)h * M * DH;
    const bf16_t* Et = ErT + (long)h * DH * Mp + (long)l31 * Mp + 8 * h5;     // + 32*dblk*Mp + Rb + 16*s2
    const LaneAddr<DH> la(lane);

    const int qi = qw0 + l31;
    const bool qok = qi < S;
    bf16x8 qf[DH / 16], gf[DH / 16];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = qok ? *reinterpret_cast<const bf16x8*>(base + (long)qi * rs + ks * 16 + 8 * h5) : z;
        gf[ks] = qok ? *reinterpret_cast<const bf16x8*>(gp + (long)qi * H * DH + ks * 16 + 8 * h5) : z;
    }
    const float lse2 = qok ? lse[((long)b * H + h) * S + qi] * LOG2E : INFINITY;
    const float ndelta = qok ? -delta[((long)b * H + h) * S + qi] : 0.f;

    f32x16 dq[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;

    bf16x8 ef[DH / 16];
    auto g_block = [&](int Rb) {
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks) g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ef[ks], qf[ks], g, 0, 0, 0);
        if (Rb >= 32) load_er_rows<DH>(ef, Eh, M, Rb - 32, l31, h5);
        ring_put_g(ring, g, Rb, l31, h5);
    };
    if (qw0 < S) {
        load_er_rows<DH>(ef, Eh, M, qw0, l31, h5);
        g_block(qw0);
    }

    const int kmax = min(q0 + QB, S);
    const int ntiles = (kmax + 63) / 64;
    TileRegs<DH, NT> kr, vr;
    tile_load<DH, NT>(kr, kp, rs, 0, S, tid);
    tile_load<DH, NT>(vr, vp, rs, 0, S, tid);
    tile_store<DH, NT>(kr, smem, tid);
    tile_store<DH, NT>(vr, smem + TB, tid);
    if (ntiles > 1) {
        tile_load<DH, NT>(kr, kp, rs, 64, S, tid);
        tile_load<DH, NT>(vr, vp, rs, 64, S, tid);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        const char* Ks = smem + (t & 1) * 2 * TB;
        const char* Vs = Ks + TB;
        if (t + 1 < ntiles) {
            char* Kn = smem + ((t + 1) & 1) * 2 * TB;
            tile_store<DH, NT>(kr, Kn, tid);
            tile_store<DH, NT>(vr, Kn + TB, tid);
        }
        if (t + 2 < ntiles) {
            tile_load<DH, NT>(kr, kp, rs, j0 + 128, S, tid);
            tile_load<DH, NT>(vr, vp, rs, j0 + 128, S, tid);
        }
        if (j0 <= qw0 + 31 && qw0 < S) {
            const int dj = qi - j0 - 4 * h5;
            const bool diag = j0 + 63 > qw0;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int R0 = qw0 - j0 - 32 * tb;
                if (R0 < 0) continue;                              // wave-uniform: block fully masked
                // operand of the dG product for this step's completed distance block, fetched early
                bf16x8 etf[2][DH / 32];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d)
                        etf[s2][d] = *reinterpret_cast<const bf16x8*>(Et + (long)(32 * d) * Mp + R0 + 16 * s2);
                if (R0 >= 32) g_block(R0 - 32);
                f32x16 sc, dp;
                ring_get_s(ring, sc, R0, l31, h5);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = ndelta;
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(tb, ks)), qf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Vs + la.row_read(tb, ks)), gf[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fast_exp2(fmaf(sc[r], c, -lse2));
                    if (diag && (32 * tb + (r & 3) + 8 * (r >> 2) > dj)) p = 0.f;
                    sc[r] = p * dp[r] * scale;
                }
                ring_put_s(ring, sc, R0, l31, h5);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 dsf = pack8(sc, 8 * s2);
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d) {
                        const bf16x8 kt = lds_read_tr_pair(Ks + la.tr_read(tb, s2, 0, d), Ks + la.tr_read(tb, s2, 1, d));
                        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, dsf, dq[d], 0, 0, 0);
                    }
                }
                // distance block [R0, R0+32) of dG^T is complete (its other half came from the previous key block)
                const float* row = ring + l31 * SKS;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 dgf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = R0 + 16 * s2 + 8 * h5 + e;
                        const float v = row[(r - l31 - 1) & 63];
                        dgf[e] = (bf16_t)((r > qi) ? 0.f : v);     // r > i: key index below 0, never written
                    }
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d)
                        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(etf[s2][d], dgf, dq[d], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (qok) {
        bf16_t* op = dqkv + ((long)b * S + qi) * rs + (long)h * DH;
#pragma unroll
        for (int d = 0; d < DH / 32; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (bf16_t)dq[d][4 * g + k];
                *reinterpret_cast<bf16x4*>(op + 32 * d + 8 * g + 4 * h5) = v;
            }
    }
}

// ---- skew buffer of the row-in-register kernels (dK/dV, dE) -----------------------------------
// buf[i_l][u]: a block in (query row i_l in registers, X on the lane) coordinates is written at
// u = x_l (+32 for the upper of two neighbouring 32-blocks) and read back at u = 32 + i_l - y_l:
// this turns distances into keys (dK/dV kernel) and keys into distances (dE kernel) alike, since
// key = query - distance.
constexpr int BKS = 72;
__device__ __forceinline__ void band_put(float* buf, const f32x16& a, int upper, int l31, int h5) {
#pragma unroll
    for (int e = 0; e < 16; ++e) buf[(8 * (e >> 2) + 4 * h5 + (e & 3)) * BKS + 32 * upper + l31] = a[e];
}
__device__ __forceinline__ void band_get(const float* buf, f32x16& a, int l31, int h5) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int il = 8 * (e >> 2) + 4 * h5 + (e & 3);
        a[e] = buf[il * BKS + 32 + il - l31];
    }
}

// dK, dV: key on the lane.  Per 32-query block (distance D = I0 - jw0 between block starts):
//   G'_lo, G'_hi = Q.Er^T for distances [D-32, D), [D, D+32)  -> skew to key coordinates
//   S = Q.K^T + skew(G') - lse   dP = dO.V^T - delta   P = exp2, dS = P*dP*scale
//   dV^T += dO^T.P               dK^T += Q^T.dS
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, (DH == 128 ? 1 : 2)) void rel_bwd_dkv_mfma_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
    const float* __restrict__ delta, const bf16_t* __restrict__ E, bf16_t* __restrict__ dqkv, int S, int H, int M,
    float c, float scale) {
    constexpr int NT = 64 * NW, KB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (Q, dO images), stats, NW skew buffers
    float* stats = reinterpret_cast<float*>(smem + 4 * TB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    float* skew = stats + 4 * 64 + w * 32 * BKS;
    const int jb0 = blockIdx.z * KB, jw0 = jb0 + 32 * w;
    const int b = blockIdx.y, h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const bf16_t* gp = dout + (long)b * S * H * DH + (long)h * DH;
    const float* lse_bh = lse + ((long)b * H + h) * S;
    const float* delta_bh = delta + ((long)b * H + h) * S;
    const bf16_t* Eh = E + (long)h * M * DH;
    const LaneAddr<DH> la(lane);

    const int j = jw0 + l31;
    const bool jok = j < S;
    bf16x8 kf[DH / 16], vf[DH / 16];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        kf[ks] = jok ? *reinterpret_cast<const bf16x8*>(kp + (long)j * rs + ks * 16 + 8 * h5) : z;
        vf[ks] = jok ? *reinterpret_cast<const bf16x8*>(vp + (long)j * rs + ks * 16 + 8 * h5) : z;
    }
    const float inv_c = 1.f / c;

    f32x16 dk[DH / 32], dv[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[d][r] = dv[d][r] = 0.f;

    // Er rows of the distance blocks [D-32, D) and [D, D+32); D grows by 32 per processed block
    bf16x8 er_lo[DH / 16], er_hi[DH / 16];
    load_er_rows<DH>(er_hi, Eh, M, 0, l31, h5);
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) er_lo[ks] = er_hi[ks];

    const int t0 = jb0 / 64;
    const int ntiles = (S + 63) / 64;
    TileRegs<DH, NT> qr, gr;
    float r_lse = 0.f, r_delta = 0.f;
    auto load_tile = [&](int i0) {
        tile_load<DH, NT>(qr, base, rs, i0, S, tid);
        tile_load<DH, NT>(gr, gp, (unsigned)(H * DH), i0, S, tid);
        if (tid < 64 && i0 + tid < S) {
            r_lse = lse_bh[i0 + tid];
            r_delta = delta_bh[i0 + tid];
        }
    };
    auto store_tile = [&](int i0, int buf) {
        tile_store<DH, NT>(qr, smem + buf * 2 * TB, tid);
        tile_store<DH, NT>(gr, smem + buf * 2 * TB + TB, tid);
        if (tid < 64) {
            const bool ok = i0 + tid < S;
            stats[buf * 128 + tid] = ok ? -(r_lse * LOG2E * inv_c) : -INFINITY;
            stats[buf * 128 + 64 + tid] = ok ? -r_delta : 0.f;
        }
    };
    load_tile(t0 * 64);
    store_tile(t0 * 64, 0);
    if (t0 + 1 < ntiles) load_tile(t0 * 64 + 64);
    __syncthreads();
    for (int t = t0; t < ntiles; ++t) {
        const int i0 = t * 64;
        const int buf = (t - t0) & 1;
        const char* Qs = smem + buf * 2 * TB;
        const char* Gs = Qs + TB;
        const float* Ls = stats + buf * 128;
        const float* Dn = Ls + 64;
        if (t + 1 < ntiles) store_tile(i0 + 64, buf ^ 1);
        if (t + 2 < ntiles) load_tile(i0 + 128);
        if (i0 + 63 >= jw0 && jw0 < S) {                           // wave-uniform
            const bool diag = i0 < jw0 + 31;
            const int di = j - i0 - 4 * h5;                        // masked when row-part < di
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int D = i0 + 32 * tb - jw0;
                if (D < 0) continue;                               // block entirely above the diagonal
                bf16x8 af[DH / 16];
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) af[ks] = lds_read16(Qs + la.row_read(tb, ks));
                {
                    f32x16 glo, ghi;
#pragma unroll
                    for (int e = 0; e < 16; ++e) glo[e] = ghi[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks) {
                        ghi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], er_hi[ks], ghi, 0, 0, 0);
                        if (D > 0) glo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], er_lo[ks], glo, 0, 0, 0);
                    }
                    band_put(skew, glo, 0, l31, h5);
                    band_put(skew, ghi, 1, l31, h5);
                    // next block's distances: [D, D+32) becomes the lower block
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks) er_lo[ks] = er_hi[ks];
                    load_er_rows<DH>(er_hi, Eh, M, D + 32, l31, h5);
                }
                f32x16 sc, dp;
                band_get(skew, sc, l31, h5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 lv = *reinterpret_cast<const f32x4*>(Ls + 32 * tb + 8 * g + 4 * h5);
                    const f32x4 dn = *reinterpret_cast<const f32x4*>(Dn + 32 * tb + 8 * g + 4 * h5);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        sc[4 * g + k] += lv[k];
                        dp[4 * g + k] = dn[k];
                    }
                }
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], kf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Gs + la.row_read(tb, ks)), vf[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = fast_exp2(sc[r] * c);
                    if (diag && (32 * tb + (r & 3) + 8 * (r >> 2) < di)) p = 0.f;
                    sc[r] = p;
                    dp[r] = p * dp[r] * scale;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 pf = pack8(sc, 8 * s2);
                    const bf16x8 dsf = pack8(dp, 8 * s2);
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d) {
                        const bf16x8 gt = lds_read_tr_pair(Gs + la.tr_read(tb, s2, 0, d), Gs + la.tr_read(tb, s2, 1, d));
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt, pf, dv[d], 0, 0, 0);
                        const bf16x8 qt = lds_read_tr_pair(Qs + la.tr_read(tb, s2, 0, d), Qs + la.tr_read(tb, s2, 1, d));
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dsf, dk[d], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (jok) {
        bf16_t* op = dqkv + ((long)b * S + j) * rs + (long)h * DH;
#pragma unroll
        for (int d = 0; d < DH / 32; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 a, v;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[k] = (bf16_t)dk[d][4 * g + k];
                    v[k] = (bf16_t)dv[d][4 * g + k];
                }
                *reinterpret_cast<bf16x4*>(op + (long)H * DH + 32 * d + 8 * g + 4 * h5) = a;
                *reinterpret_cast<bf16x4*>(op + 2L * H * DH + 32 * d + 8 * g + 4 * h5) = v;
            }
    }
}

// dE: distance on the lane.  A workgroup owns 32*NW distances of one head and walks, for every
// batch element, down the band of the score matrix those distances select: for the 32-query block
// at I0 the keys are j = i - r, two 32-key blocks starting at JB - 32 and JB = I0 - rw0.
//   S = Q.K^T, dP = dO.V^T on those key blocks -> skew to distance coordinates
//   W = skew(S) + Q.Er^T - lse   P = exp2   dW = P*(skew(dP) - delta)*scale
//   dEr^T[d, r] += Q^T . dW      (accumulators live in registers for the whole launch)
template <int DH, int NWD>
__device__ __forceinline__ void rel_de_band(
    char* smem, const int rb0, const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv,
    const float* __restrict__ lse, const float* __restrict__ delta, const bf16_t* __restrict__ E,
    float* __restrict__ dE, int B, int S, int H, int M, float c, float scale) {
    constexpr int NW = 2 * NWD, NT = 64 * NW, TB = Img<DH>::TILE_BYTES;
    static_assert(NWD % 2 == 0, "a band must cover whole 64-key tiles");
    // Keys/values of the band: staged through a 3-slot LDS ring of 64-key tiles (tiles T-1, T in use,
    // T+1 being filled) when it fits (dh 64); at dh 128 the ring would exceed 160 KB and each wave
    // fetches its two 32-key operand blocks straight from global memory instead.
    constexpr bool KVL = DH == 64;
    char* kvring = smem + 4 * TB;
    float* stats = reinterpret_cast<float*>(smem + (KVL ? 10 : 4) * TB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    float* skew = stats + 4 * 64 + w * 32 * BKS;
    const int rw0 = rb0 + 32 * (w % NWD);
    const int tb = w / NWD;                               // which 32-query half of every staged tile this wave takes
    const int h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* Eh = E + (long)h * M * DH;
    const LaneAddr<DH> la(lane);
    const int r = rw0 + l31;
    const float inv_c = 1.f / c;

    bf16x8 erf[DH / 16];
    load_er_rows<DH>(erf, Eh, M, rw0, l31, h5);
    f32x16 de[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) de[d][e] = 0.f;

    const int t0 = rb0 / 64;
    const int ntiles = (S + 63) / 64;
    for (int b = 0; b < B && rb0 < S; ++b) {
        const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
        const bf16_t* kp = base + (long)H * DH;
        const bf16_t* vp = base + 2L * H * DH;
        const bf16_t* gp = dout + (long)b * S * H * DH + (long)h * DH;
        const float* lse_bh = lse + ((long)b * H + h) * S;
        const float* delta_bh = delta + ((long)b * H + h) * S;
        TileRegs<DH, NT> qr, gr, kr, vr;
        float r_lse = 0.f, r_delta = 0.f;
        // query tile at i0 travels with key tile at i0 - rb0 (the keys at distance rb0.. below it)
        auto load_tile = [&](int i0) {
            tile_load<DH, NT>(qr, base, rs, i0, S, tid);
            tile_load<DH, NT>(gr, gp, (unsigned)(H * DH), i0, S, tid);
            if (KVL) {
                tile_load<DH, NT>(kr, kp, rs, i0 - rb0, S, tid);
                tile_load<DH, NT>(vr, vp, rs, i0 - rb0, S, tid);
            }
            if (tid < 64 && i0 + tid < S) {
                r_lse = lse_bh[i0 + tid];
                r_delta = delta_bh[i0 + tid];
            }
        };
        auto store_tile = [&](int i0, int buf) {
            tile_store<DH, NT>(qr, smem + buf * 2 * TB, tid);
            tile_store<DH, NT>(gr, smem + buf * 2 * TB + TB, tid);
            if (KVL) {
                char* slot = kvring + (((i0 - rb0) >> 6) % 3) * 2 * TB;
                tile_store<DH, NT>(kr, slot, tid);
                tile_store<DH, NT>(vr, slot + TB, tid);
            }
            if (tid < 64) {
                const bool ok = i0 + tid < S;
                stats[buf * 128 + tid] = ok ? -(r_lse * LOG2E * inv_c) : -INFINITY;
                stats[buf * 128 + 64 + tid] = ok ? -r_delta : 0.f;
            }
        };
        bf16x8 ka[DH / 16], kb[DH / 16], va[DH / 16], vb[DH / 16];      // !KVL: operand blocks from global
        auto load_kv = [&](int J) {                        // rows J-32.. (lower block) and J.. (upper block)
            const int ra = min(max(J - 32 + l31, 0), S - 1), rb = min(max(J + l31, 0), S - 1);
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) {
                ka[ks] = *reinterpret_cast<const bf16x8*>(kp + (long)ra * rs + 16 * ks + 8 * h5);
                va[ks] = *reinterpret_cast<const bf16x8*>(vp + (long)ra * rs + 16 * ks + 8 * h5);
                kb[ks] = *reinterpret_cast<const bf16x8*>(kp + (long)rb * rs + 16 * ks + 8 * h5);
                vb[ks] = *reinterpret_cast<const bf16x8*>(vp + (long)rb * rs + 16 * ks + 8 * h5);
            }
        };
        // the wave's first block is the first I0 = 64 t + 32 tb >= rw0
        if (!KVL) load_kv(((max(rw0 - 32 * tb, 0) + 63) / 64) * 64 + 32 * tb - rw0);
        __syncthreads();                                   // previous batch element's tiles are consumed
        load_tile(t0 * 64);
        store_tile(t0 * 64, 0);
        if (t0 + 1 < ntiles) load_tile(t0 * 64 + 64);
        __syncthreads();
        for (int t = t0; t < ntiles; ++t) {
            const int i0 = t * 64;
            const int buf = (t - t0) & 1;
            const char* Qs = smem + buf * 2 * TB;
            const char* Gs = Qs + TB;
            const float* Ls = stats + buf * 128;
            const float* Dn = Ls + 64;
            if (t + 1 < ntiles) store_tile(i0 + 64, buf ^ 1);
            if (t + 2 < ntiles) load_tile(i0 + 128);
            if (rw0 < S) {                                     // wave-uniform
                {
                    const int JB = i0 + 32 * tb - rw0;         // key of (query block start, distance rw0)
                    if (JB < 0) goto next_tile;
                    bf16x8 af[DH / 16], gfr[DH / 16];
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks) {
                        af[ks] = lds_read16(Qs + la.row_read(tb, ks));
                        gfr[ks] = lds_read16(Gs + la.row_read(tb, ks));
                    }
                    // upper key block [JB, JB+32) and lower [JB-32, JB): halves of ring tiles
                    const char* Kb = kvring + ((JB >> 6) % 3) * 2 * TB;
                    const int hb = (JB >> 5) & 1;
                    const int JA = max(JB - 32, 0);
                    const char* Ka = kvring + ((JA >> 6) % 3) * 2 * TB;
                    const int ha = (JA >> 5) & 1;
                    f32x16 sc, dp;
                    {
                        f32x16 sa, sb;
#pragma unroll
                        for (int e = 0; e < 16; ++e) sa[e] = sb[e] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < DH / 16; ++ks) {
                            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], KVL ? lds_read16(Kb + la.row_read(hb, ks)) : kb[ks], sb, 0, 0, 0);
                            if (JB > 0) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], KVL ? lds_read16(Ka + la.row_read(ha, ks)) : ka[ks], sa, 0, 0, 0);
                        }
                        band_put(skew, sa, 0, l31, h5);
                        band_put(skew, sb, 1, l31, h5);
                        band_get(skew, sc, l31, h5);
                    }
                    {
                        f32x16 pa, pb;
#pragma unroll
                        for (int e = 0; e < 16; ++e) pa[e] = pb[e] = 0.f;
#pragma unroll
                        for (int ks = 0; ks < DH / 16; ++ks) {
                            pb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[ks], KVL ? lds_read16(Kb + TB + la.row_read(hb, ks)) : vb[ks], pb, 0, 0, 0);
                            if (JB > 0) pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[ks], KVL ? lds_read16(Ka + TB + la.row_read(ha, ks)) : va[ks], pa, 0, 0, 0);
                        }
                        band_put(skew, pa, 0, l31, h5);
                        band_put(skew, pb, 1, l31, h5);
                        band_get(skew, dp, l31, h5);
                    }
                    if (!KVL) load_kv(JB + 64);                // this wave's next block (one tile further)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 lv = *reinterpret_cast<const f32x4*>(Ls + 32 * tb + 8 * g + 4 * h5);
                        const f32x4 dn = *reinterpret_cast<const f32x4*>(Dn + 32 * tb + 8 * g + 4 * h5);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            sc[4 * g + k] += lv[k];
                            dp[4 * g + k] += dn[k];
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks)
                        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], erf[ks], sc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int il = 8 * (e >> 2) + 4 * h5 + (e & 3);
                        float p = fast_exp2(sc[e] * c);
                        if (JB + il - l31 < 0) p = 0.f;        // key index below 0 (distance beyond the query)
                        sc[e] = p * dp[e] * scale;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 dwf = pack8(sc, 8 * s2);
#pragma unroll
                        for (int d = 0; d < DH / 32; ++d) {
                            const bf16x8 qt = lds_read_tr_pair(Qs + la.tr_read(tb, s2, 0, d), Qs + la.tr_read(tb, s2, 1, d));
                            de[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dwf, de[d], 0, 0, 0);
                        }
                    }
                }
            }
        next_tile:
            __syncthreads();
        }
    }
    // the two query-half groups hold partial sums of the same distances: add them in a fixed order
    {
        float* red = reinterpret_cast<float*>(smem) + (size_t)(w % NWD) * (DH / 32) * 16 * 64;
        if (tb == 1) {
#pragma unroll
            for (int d = 0; d < DH / 32; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(d * 16 + e) * 64 + lane] = de[d][e];
        }
        __syncthreads();
        if (tb == 0) {
#pragma unroll
            for (int d = 0; d < DH / 32; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) de[d][e] += red[(d * 16 + e) * 64 + lane];
        }
    }
    if (tb == 0 && r < M) {
        float* op = dE + ((long)h * M + (M - 1 - r)) * DH;
#pragma unroll
        for (int d = 0; d < DH / 32; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = de[d][4 * g + k];
                *reinterpret_cast<f32x4*>(op + 32 * d + 8 * g + 4 * h5) = v;
            }
    }
}

// Band y and band N-1-y have complementary lengths (S - r queries each): one workgroup takes both,
// so every workgroup of the launch carries the same amount of work.
template <int DH, int NWD>
__global__ __launch_bounds__(128 * NWD, 1) void rel_bwd_de_mfma_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
    const float* __restrict__ delta, const bf16_t* __restrict__ E, float* __restrict__ dE, int B, int S, int H, int M,
    int nbands, float c, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (Q, dO images), stats, NW skew buffers
    const int y = blockIdx.y, y2 = nbands - 1 - y;
    rel_de_band<DH, NWD>(smem, y * 32 * NWD, dout, qkv, lse, delta, E, dE, B, S, H, M, c, scale);
    if (y2 > y) {
        __syncthreads();
        rel_de_band<DH, NWD>(smem, y2 * 32 * NWD, dout, qkv, lse, delta, E, dE, B, S, H, M, c, scale);
    }
}

template <typename F>
void allow_smem(F* fn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int DH>
constexpr size_t rel_fwd_smem(int nw) { return (size_t)4 * Img<DH>::TILE_BYTES + (size_t)nw * 32 * SKS * sizeof(float); }

}  // namespace

int mgx_rel_fwd_mfma(const void* qkv, const void* E, void* out, float* lse, int B, int S, int H, int dh, int max_seq,
                     hipStream_t st) {
    const float c = LOG2E / sqrtf((float)dh);
    constexpr int NW = REL_NW;
    dim3 grid(H, B, (S + 32 * NW - 1) / (32 * NW));
    allow_smem(rel_fwd_mfma_kernel<64, NW>, rel_fwd_smem<64>(NW));
    allow_smem(rel_fwd_mfma_kernel<128, NW>, rel_fwd_smem<128>(NW));
    if (dh == 64)
        hipLaunchKernelGGL((rel_fwd_mfma_kernel<64, NW>), grid, dim3(64 * NW), rel_fwd_smem<64>(NW), st,
                           (const bf16_t*)qkv, (const bf16_t*)E, (bf16_t*)out, lse, S, H, max_seq, c);
    else
        hipLaunchKernelGGL((rel_fwd_mfma_kernel<128, NW>), grid, dim3(64 * NW), rel_fwd_smem<128>(NW), st,
                           (const bf16_t*)qkv, (const bf16_t*)E, (bf16_t*)out, lse, S, H, max_seq, c);
    MGX_CHECK_LAUNCH("rel_fwd_mfma");
    return MGX_OK;
}

bool mgx_rel_bwd_mfma_ready() { return true; }

namespace {
template <int DH>
int launch_rel_bwd(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                   float* dE, float* delta, int B, int S, int H, int max_seq, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)DH);
    const float c = LOG2E * scale;
    constexpr int NW = REL_NW;
    const int M = max_seq, Mp = (M + 31) / 32 * 32;
    // the dE buffer (fp32 [H,M,DH]) holds the transposed bf16 copy of E (H*DH*Mp*2 bytes <= its size
    // since Mp <= 2M) until the dE kernel, which runs last, overwrites it
    bf16_t* ErT = reinterpret_cast<bf16_t*>(dE);
    const long n_t = (long)H * DH * Mp;
    hipLaunchKernelGGL(rel_ert_kernel, dim3((unsigned)((n_t + 255) / 256)), dim3(256), 0, st, (const bf16_t*)E, ErT, H, M,
                       Mp, DH);
    MGX_CHECK_LAUNCH("rel_ert");
    const long n_chunks = (long)B * S * H * (DH / 8);
    hipLaunchKernelGGL((attn_delta_mfma_kernel<DH>), dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)dout, (const bf16_t*)out, delta, n_chunks, S, H);
    MGX_CHECK_LAUNCH("attn_delta_mfma");
    const size_t band_smem = (size_t)4 * Img<DH>::TILE_BYTES + 4 * 64 * sizeof(float) + (size_t)NW * 32 * BKS * sizeof(float);
    dim3 grid(H, B, (S + 32 * NW - 1) / (32 * NW));
    allow_smem(rel_bwd_dkv_mfma_kernel<DH, NW>, band_smem);
    hipLaunchKernelGGL((rel_bwd_dkv_mfma_kernel<DH, NW>), grid, dim3(64 * NW), band_smem, st, (const bf16_t*)dout,
                       (const bf16_t*)qkv, lse, delta, (const bf16_t*)E, (bf16_t*)dqkv, S, H, M, c, scale);
    MGX_CHECK_LAUNCH("rel_bwd_dkv_mfma");
    allow_smem(rel_bwd_dq_mfma_kernel<DH, NW>, rel_fwd_smem<DH>(NW));
    hipLaunchKernelGGL((rel_bwd_dq_mfma_kernel<DH, NW>), grid, dim3(64 * NW), rel_fwd_smem<DH>(NW), st,
                       (const bf16_t*)dout, (const bf16_t*)qkv, lse, delta, (const bf16_t*)E, (const bf16_t*)ErT,
                       (bf16_t*)dqkv, S, H, M, Mp, c, scale);
    MGX_CHECK_LAUNCH("rel_bwd_dq_mfma");
    constexpr int NWE = REL_DE_NW;
    const int nbands = (M + 32 * NWE - 1) / (32 * NWE);
    const size_t de_smem = (size_t)(DH == 64 ? 10 : 4) * Img<DH>::TILE_BYTES + 4 * 64 * sizeof(float) + (size_t)2 * NWE * 32 * BKS * sizeof(float);
    dim3 gride(H, (nbands + 1) / 2);
    allow_smem(rel_bwd_de_mfma_kernel<DH, NWE>, de_smem);
    hipLaunchKernelGGL((rel_bwd_de_mfma_kernel<DH, NWE>), gride, dim3(128 * NWE), de_smem, st, (const bf16_t*)dout,
                       (const bf16_t*)qkv, lse, delta, (const bf16_t*)E, dE, B, S, H, M, nbands, c, scale);
    MGX_CHECK_LAUNCH("rel_bwd_de_mfma");
    return MGX_OK;
}
}  // namespace

int mgx_rel_bwd_mfma(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                     float* dE, float* delta, int B, int S, int H, int dh, int max_seq, hipStream_t st) {
    if (dh == 64) return launch_rel_bwd<64>(dout, qkv, out, lse, E, dqkv, dE, delta, B, S, H, max_seq, st);
    return launch_rel_bwd<128>(dout, qkv, out, lse, E, dqkv, dE, delta, B, S, H, max_seq, st);
}
