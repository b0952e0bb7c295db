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
// on the lane for dK/dV).  dE = sum_i dG[i, r] q_i: the dQ kernel, which forms dG in distance
// coordinates for its own second term, also stores it (bf16 blocks), and a batched causal TN GEMM
// over those blocks (gemm_tn.hip, MODE 1) + a fixed-order sum over the batch give the table's
// gradient: no atomics, bitwise reproducible, no recomputation of S / dP for dE.
#include "attn_mfma_common.h"

namespace {

constexpr int REL_NW = 4;        // waves (x 32 rows) per workgroup

// ---- the skew ring (query-on-lane kernels) ----------------------------------------------------
// G layout: accumulator of Er_blk . Q^T : lane = (query i_l, half h5), register e <-> distance
//           Rb + 8*(e>>2) + 4*h5 + (e&3)
// S layout: accumulator of K_blk . Q^T  : register e <-> key jb + 8*(e>>2) + 4*h5 + (e&3)
// Physical image: the 64 positions of a row are two 32-position PLANES, RING_PLANE bytes apart; a wave's plane is
// 32 rows x 128 B whose 16-byte quads are XOR-swizzled with three bits of i_l, chosen (by search) so that the scalar
// accesses of 32 lanes, the 16-byte reads of 16 lanes and the 16-byte writes of 8 lanes each touch distinct banks.  Rb and R0 are multiples of 32, so a step of the walk only moves an access to
// the other plane: every address is a per-lane constant (RingMap, computed once) XOR a wave-uniform plane bit.  (The
// round-1 form computed (r - i_l - 1) & 63 per element: ~130 VALU instructions of address arithmetic per key block.)
constexpr unsigned RING_PLANE = 0x4000;            // >= REL_NW * 4096, one bit
constexpr unsigned RING_BYTES = 2 * RING_PLANE;    // all waves of the workgroup
__device__ __forceinline__ unsigned ring_flip(int Rb) { return (unsigned)(Rb & 32) << 9; }   // odd block -> RING_PLANE

struct RingMap {
    unsigned dist[16];   // distance coordinates: the lane's 16 accumulator elements of an even block (scalar accesses)
    unsigned key[4];     // key coordinates: the lane's 4 quads of an even block (16-byte accesses)
    // ring_off: byte offset of the ring area in the workgroup's LDS; a multiple of RING_BYTES
    __device__ __forceinline__ RingMap(unsigned ring_off, int w, int l31, int h5) {
        const unsigned row = ring_off + 4096u * w + 128u * l31;
        const unsigned f = ((l31 >> 1) & 3) | ((((l31 >> 3) ^ l31) & 1) << 2);   // conflict-free for all four access kinds
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int u = 8 * (e >> 2) + 4 * h5 + (e & 3) - l31 - 1;                  // position - Rb, in [-32, 30]
            dist[e] = row + (u < 0 ? RING_PLANE : 0u) + 16u * ((((unsigned)u & 31) >> 2) ^ f) + 4u * ((unsigned)u & 3);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) key[g] = row + RING_PLANE + 16u * ((unsigned)(7 - 2 * g - h5) ^ f);   // positions R0 - 8g - 4h5 - 4 ..
    }
};
__device__ __forceinline__ void ring_put_g(char* smem, const RingMap& rm, const f32x16& g, int Rb) {
    const unsigned fl = ring_flip(Rb);
#pragma unroll
    for (int e = 0; e < 16; ++e) *reinterpret_cast<float*>(smem + (rm.dist[e] ^ fl)) = g[e];
}
// the same 16 elements back, for the distance block [R0, R0 + 32) (register e <-> distance R0 + 8*(e>>2) + 4*h5 + (e&3))
__device__ __forceinline__ void ring_get_g(const char* smem, const RingMap& rm, f32x16& g, int R0) {
    const unsigned fl = ring_flip(R0);
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = *reinterpret_cast<const float*>(smem + (rm.dist[e] ^ fl));
}
__device__ __forceinline__ void ring_get_s(const char* smem, const RingMap& rm, f32x16& s, int R0) {
    const unsigned fl = ring_flip(R0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + (rm.key[g] ^ fl));
        s[4 * g + 0] = v[3];
        s[4 * g + 1] = v[2];
        s[4 * g + 2] = v[1];
        s[4 * g + 3] = v[0];
    }
}
__device__ __forceinline__ void ring_put_s(char* smem, const RingMap& rm, const f32x16& s, int R0) {
    const unsigned fl = ring_flip(R0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[3] = s[4 * g + 0];
        v[2] = s[4 * g + 1];
        v[1] = s[4 * g + 2];
        v[0] = s[4 * g + 3];
        *reinterpret_cast<f32x4*>(smem + (rm.key[g] ^ fl)) = v;
    }
}

// Er rows Rb + l31 as an MFMA operand (row on the lane, 8 consecutive d per lane and k-step), straight from HBM (the
// dK/dV kernel); Er[r] = E[h, M-1-r]; rows outside [0, M) are clamped (their products are never used).
template <int DH>
__device__ __forceinline__ void load_er_rows(bf16x8 (&ef)[DH / 16], const bf16_t* Eh, int M, int Rb, int l31, int h5) {
    int idx = M - 1 - (Rb + l31);
    idx = min(max(idx, 0), M - 1);
    const bf16_t* p = Eh + (long)idx * DH + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks) ef[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// ---- Er as a third tile stream (query-on-lane kernels) ---------------------------------------------------------
// Er tile T = distances [64 T, 64 T + 64) in the K / V image format (image row i <-> table row M-1-64T-i, clamped: a
// clamped row only meets masked keys).  A workgroup at key tile j0 touches the distance blocks (q0 - j0)/32 - 2 .. + 3:
// three tiles, of which ONE is new per key tile; they live in three LDS slots (tile T in slot T mod 3) and are moved
// with the K / V tiles (64 full lines per key tile and workgroup).  Round 1-2 had every wave gather its own 32 rows
// per key block (and the dQ kernel its ErT pieces) with per-lane 16-byte loads: 8-12 vector-memory instructions of 32
// cache lines each per block and wave, and the texture path - not MFMA, VALU or LDS - set the kernels' pace (measured
// by removing those loads: dQ 887 -> 525 us, forward 401 -> 316 us at B8 S4096 dh64).
template <int DH, int NT>
__device__ __forceinline__ void er_tile_load(TileRegs<DH, NT>& t, const bf16_t* Eh, int M, int T, int tid) {
#pragma unroll
    for (int n = 0; n < 8 * DH / NT; ++n) {
        int row, ch;
        tile_elem<DH>(tid + NT * n, row, ch);
        const int idx = min(max(M - 1 - (64 * T + row), 0), M - 1);
        t.v[n] = *reinterpret_cast<const bf16x8*>(Eh + ((unsigned)idx * (unsigned)DH + (unsigned)(ch * 8)));
    }
}
__device__ __forceinline__ int er_slot(int T) { return (T + 6) % 3; }      // T >= -6

// LDS of the query-on-lane kernels: rings | K | V | 3 Er tiles
template <int DH>
struct RelLds {
    static constexpr int TB = Img<DH>::TILE_BYTES;
    static constexpr int K = RING_BYTES, V = K + TB, ER = V + TB, BYTES = ER + 3 * TB;
};

// ===========================================================================
// forward
// ===========================================================================
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, (DH == 128 ? 1 : 2)) void rel_fwd_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                               const bf16_t* __restrict__ E,
                                                               bf16_t* __restrict__ out, float* __restrict__ lse,
                                                               int S, int H, int M, float c /* scale*log2e */) {
    constexpr int NT = 64 * NW, QB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    using L = RelLds<DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h5 = lane >> 5;
    static_assert(NW == 4 && NW * 4096 <= RING_PLANE, "the Er window is three tiles for 128 queries per workgroup");
    const RingMap rm(0, w, l31, h5);
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

    // distance blocks are consumed in descending order qw0, qw0-32, ..., 0
    auto g_block = [&](int Rb) {                           // G[Rb..Rb+32) = Er_blk . Q^T -> ring
        const char* ep = smem + L::ER + er_slot(Rb >> 6) * TB;
        const int tbe = (Rb >> 5) & 1;
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks)
            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(ep + la.row_read(tbe, ks)), qf[ks], g, 0, 0, 0);
        ring_put_g(smem, rm, g, Rb);
    };

    const int kmax = min(q0 + QB, S);
    const int ntiles = (kmax + 63) / 64;
    const int T00 = q0 >> 6;                               // Er tiles T00 - t - 1 .. T00 - t + 1 serve key tile t
    TileRegs<DH, NT> kr, vr, er;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
        er_tile_load<DH, NT>(er, Eh, M, T00 + i, tid);
        tile_store<DH, NT>(er, smem + L::ER + er_slot(T00 + i) * TB, tid);
    }
    tile_load<DH, NT>(kr, kp, rs, 0, S, tid);
    tile_load<DH, NT>(vr, vp, rs, 0, S, tid);
    tile_store<DH, NT>(kr, smem + L::K, tid);
    tile_store<DH, NT>(vr, smem + L::V, tid);
    if (ntiles > 1) {
        tile_load<DH, NT>(kr, kp, rs, 64, S, tid);
        tile_load<DH, NT>(vr, vp, rs, 64, S, tid);
        er_tile_load<DH, NT>(er, Eh, M, T00 - 2, tid);
    }
    __syncthreads();
    if (qw0 < S) g_block(qw0);
    const char* Ks = smem + L::K;
    const char* Vs = smem + L::V;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        if (j0 <= qw0 + 31 && qw0 < S) {                  // wave-uniform
            f32x16 s[2];
            if (qw0 - j0 >= 64) {
                // both key blocks lie wholly in the past: one straight-line block, the four products independent of
                // each other and of the ring's round trips (LDS operations of a wave are executed in issue order,
                // so the second G block may be written as soon as the first key block's reads have been ISSUED)
                const int R0 = qw0 - j0;
                const char* e0 = smem + L::ER + er_slot((R0 - 32) >> 6) * TB;
                const char* e1 = smem + L::ER + er_slot((R0 - 64) >> 6) * TB;
                const int t0 = ((R0 - 32) >> 5) & 1, t1 = t0 ^ 1;
                f32x16 g0, g1, r0, r1;
#pragma unroll
                for (int e = 0; e < 16; ++e) g0[e] = g1[e] = s[0][e] = s[1][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(e0 + la.row_read(t0, ks)), qf[ks], g0, 0, 0, 0);
                    g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(e1 + la.row_read(t1, ks)), qf[ks], g1, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(0, ks)), qf[ks], s[0], 0, 0, 0);
                    s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(1, ks)), qf[ks], s[1], 0, 0, 0);
                }
                ring_put_g(smem, rm, g0, R0 - 32);
                ring_get_s(smem, rm, r0, R0);
                ring_put_g(smem, rm, g1, R0 - 64);
                ring_get_s(smem, rm, r1, R0 - 32);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[0][e] += r0[e];
                    s[1][e] += r1[e];
                }
            } else
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int R0 = qw0 - j0 - 32 * tb;        // distance of (query qw0, key block start)
                if (R0 >= 0) {
                    if (R0 >= 32) g_block(R0 - 32);
                    ring_get_s(smem, rm, s[tb], R0);
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
        if (t + 1 == ntiles) break;
        __syncthreads();                                   // tile t and Er tile T00 - t + 1 are dead
        tile_store<DH, NT>(kr, smem + L::K, tid);
        tile_store<DH, NT>(vr, smem + L::V, tid);
        tile_store<DH, NT>(er, smem + L::ER + er_slot(T00 - t - 2) * TB, tid);
        if (t + 2 < ntiles) {
            tile_load<DH, NT>(kr, kp, rs, j0 + 128, S, tid);
            tile_load<DH, NT>(vr, vp, rs, j0 + 128, S, tid);
            er_tile_load<DH, NT>(er, Eh, M, T00 - t - 3, tid);
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
// dQ: query on the lane.  Per 32-key block (distance R0 = qw0 - jb of its first key):
//   S^T = K.Q^T + skew(G)   dP^T = V.dO^T - delta   dS^T = P^T*dP^T*scale   dQ^T += K^T.dS^T
//   ring <- dS^T (in place of the consumed G entries); block R0 of dG is then complete:
//   dQ^T += Er[R0..R0+32)^T . dG^T      (transposed reads of the Er tile that gave G one block earlier)
template <int DH, int NW>
__global__ __launch_bounds__(64 * NW, (DH == 128 ? 1 : 2)) void rel_bwd_dq_mfma_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv, const float* __restrict__ lse,
    const float* __restrict__ delta, const bf16_t* __restrict__ E, bf16_t* __restrict__ dqkv, char* __restrict__ dg,
    int nrb, int S, int H, int M, float c, float scale) {
    constexpr int NT = 64 * NW, QB = 32 * NW, TB = Img<DH>::TILE_BYTES;
    using L = RelLds<DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h5 = lane >> 5;
    static_assert(NW == 4 && NW * 4096 <= RING_PLANE, "the Er window is three tiles for 128 queries per workgroup");
    const RingMap rm(0, w, l31, h5);
    const int qb = gridDim.z - 1 - blockIdx.z;
    const int q0 = qb * QB, qw0 = q0 + 32 * w;
    const int b = blockIdx.y, h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const bf16_t* gp = dout + (long)b * S * H * DH + (long)h * DH;
    const bf16_t* Eh = E + (long)h * M * DH;
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

    auto g_block = [&](int Rb) {
        const char* ep = smem + L::ER + er_slot(Rb >> 6) * TB;
        const int tbe = (Rb >> 5) & 1;
        f32x16 g;
#pragma unroll
        for (int e = 0; e < 16; ++e) g[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks)
            g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(ep + la.row_read(tbe, ks)), qf[ks], g, 0, 0, 0);
        ring_put_g(smem, rm, g, Rb);
    };
    // dG of this wave's 32 queries goes to HBM block by block for the dE GEMM (gemm_tn.hip, MODE 1): 2 KiB blocks
    // [query unit: a staircase of rows, mgx_rel_dg_row][distance block][2 distance halves][32 queries][16 distances]; the
    // 16-byte piece of a lane is its
    // MFMA operand as it stands, so within a 16-distance half the stored position 8*h5 + 4*c + k holds distance
    // 8*c + 4*h5 + k (bits 2 and 3 swapped; rel_de_reduce_kernel reads the GEMM's rows through the same swap).  That
    // GEMM walks 256-distance tiles and reads, for the units inside a tile's own 256 queries, the blocks above the
    // diagonal, which nobody computes: zeros, written here.
    char* const dgp = dg + ((((long)b * H + h) * mgx_rel_dg_row(nrb) + mgx_rel_dg_row(qw0 >> 5)) << 11) + 32 * l31 + 16 * h5;
    if (qw0 < S) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int rb = (qw0 >> 5) + 1; rb <= min((qw0 >> 5) | 7, nrb - 1); ++rb) {
            *reinterpret_cast<bf16x8*>(dgp + ((long)rb << 11)) = z;
            *reinterpret_cast<bf16x8*>(dgp + ((long)rb << 11) + 1024) = z;
        }
    }

    const int kmax = min(q0 + QB, S);
    const int ntiles = (kmax + 63) / 64;
    const int T00 = q0 >> 6;                               // Er tiles T00 - t - 1 .. T00 - t + 1 serve key tile t
    TileRegs<DH, NT> kr, vr, er;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
        er_tile_load<DH, NT>(er, Eh, M, T00 + i, tid);
        tile_store<DH, NT>(er, smem + L::ER + er_slot(T00 + i) * TB, tid);
    }
    tile_load<DH, NT>(kr, kp, rs, 0, S, tid);
    tile_load<DH, NT>(vr, vp, rs, 0, S, tid);
    tile_store<DH, NT>(kr, smem + L::K, tid);
    tile_store<DH, NT>(vr, smem + L::V, tid);
    if (ntiles > 1) {
        tile_load<DH, NT>(kr, kp, rs, 64, S, tid);
        tile_load<DH, NT>(vr, vp, rs, 64, S, tid);
        er_tile_load<DH, NT>(er, Eh, M, T00 - 2, tid);
    }
    __syncthreads();
    if (qw0 < S) g_block(qw0);
    const char* Ks = smem + L::K;
    const char* Vs = smem + L::V;
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * 64;
        if (j0 <= qw0 + 31 && qw0 < S) {
            const int dj = qi - j0 - 4 * h5;
            const bool diag = j0 + 63 > qw0;                       // wave-uniform
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int R0 = qw0 - j0 - 32 * tb;
                if (R0 < 0) continue;                              // wave-uniform: block fully masked
                if (R0 >= 32) g_block(R0 - 32);
                f32x16 sc, dp;
                ring_get_s(smem, rm, sc, R0);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = ndelta;
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Ks + la.row_read(tb, ks)), qf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Vs + la.row_read(tb, ks)), gf[ks], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = fast_exp2(fmaf(sc[r], c, -lse2));
                if (diag) {                                        // the diagonal tile only: a real branch
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * tb + (r & 3) + 8 * (r >> 2) > dj) sc[r] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] *= dp[r] * scale;
                ring_put_s(smem, rm, sc, R0);
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
                f32x16 dgv;
                ring_get_g(smem, rm, dgv, R0);
                if (R0 == qw0) {                                   // key block 0: distances r > i (keys below 0) were never written
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (8 * (e >> 2) + 4 * h5 + (e & 3) > l31) dgv[e] = 0.f;
                }
                const char* ep = smem + L::ER + er_slot(R0 >> 6) * TB;
                const int tbe = (R0 >> 5) & 1;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 dgf = pack8(dgv, 8 * s2);
                    // written once, read once by the dE GEMM: non-temporal, so that the 1.07 GB do not push the K / V tiles
                    // out of L2 (relative backward 1.61 -> 1.55 ms per layer at B8 S4096)
                    __builtin_nontemporal_store(dgf, reinterpret_cast<bf16x8*>(dgp + ((long)(R0 >> 5) << 11) + 1024 * s2));
#pragma unroll
                    for (int d = 0; d < DH / 32; ++d) {
                        const bf16x8 et = lds_read_tr_pair(ep + la.tr_read(tbe, s2, 0, d), ep + la.tr_read(tbe, s2, 1, d));
                        dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(et, dgf, dq[d], 0, 0, 0);
                    }
                }
            }
        }
        if (t + 1 == ntiles) break;
        __syncthreads();                                   // tile t and Er tile T00 - t + 1 are dead
        tile_store<DH, NT>(kr, smem + L::K, tid);
        tile_store<DH, NT>(vr, smem + L::V, tid);
        tile_store<DH, NT>(er, smem + L::ER + er_slot(T00 - t - 2) * TB, tid);
        if (t + 2 < ntiles) {
            tile_load<DH, NT>(kr, kp, rs, j0 + 128, S, tid);
            tile_load<DH, NT>(vr, vp, rs, j0 + 128, S, tid);
            er_tile_load<DH, NT>(er, Eh, M, T00 - t - 3, tid);
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
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
                    // (D = 0: the lower block holds distances < 0, i.e. keys above the diagonal, masked below; er_lo is
                    // the clamped row set of the start, its product finite and unused)
                    f32x16 glo, ghi;
#pragma unroll
                    for (int e = 0; e < 16; ++e) glo[e] = ghi[e] = 0.f;
                    // the two waves of a SIMD bunch their MFMAs: while one is inside a group it keeps the issue priority, so
                    // the groups alternate instead of interleaving instruction by instruction (dK/dV 547 -> 534 us)
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ks = 0; ks < DH / 16; ++ks) {
                        ghi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], er_hi[ks], ghi, 0, 0, 0);
                        glo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], er_lo[ks], glo, 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
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
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < DH / 16; ++ks) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], kf[ks], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read16(Gs + la.row_read(tb, ks)), vf[ks], dp, 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = fast_exp2(sc[r] * c);
                if (diag) {                                        // the diagonal tile only: a real branch
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * tb + (r & 3) + 8 * (r >> 2) < di) sc[r] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] *= sc[r] * scale;
                __builtin_amdgcn_s_setprio(1);
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
                __builtin_amdgcn_s_setprio(0);
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

// dE[h, M-1-r, :] = sum_b slab[b, h][r, :] for r < Mp (the distances a query can reach), zero for the rest of the
// table; fixed summation order.  The slabs come from the dE GEMM over the stored dG blocks (gemm_tn.hip, MODE 1).
__global__ __launch_bounds__(256) void rel_de_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dE, int B,
                                                             int H, int M, int Mp, int dh) {
    const long n4 = (long)H * M * dh / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const int per_row = dh / 4;
    const int c4 = (int)(idx % per_row);
    const long hr = idx / per_row;
    const int row = (int)(hr % M), h = (int)(hr / M);
    const int r = M - 1 - row;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (r < Mp) {
        const int rp = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);     // the dQ kernel's storage order of a 16-distance half
        const float* p = slabs + ((long)h * Mp + rp) * dh + 4 * c4;
        for (int b = 0; b < B; ++b) s += *reinterpret_cast<const f32x4*>(p + (long)b * H * Mp * dh);
    }
    *reinterpret_cast<f32x4*>(dE + idx * 4) = s;
}

template <typename F>
void allow_smem(F* fn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int DH>
constexpr size_t rel_fwd_smem(int) { return (size_t)RelLds<DH>::BYTES; }

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

// workspace of the backward, in floats: delta [B,H,S] | dE slabs [B*H][32 nrb][dh] | dG blocks [B*H][staircase] x 2 KiB
long mgx_rel_bwd_mfma_workspace_floats(int B, int S, int H, int dh) {
    const int nrb = (S + 31) / 32;
    const long delta = ((long)B * H * S + 63) / 64 * 64;
    return delta + (long)B * H * nrb * 32 * dh + (long)B * H * mgx_rel_dg_row(nrb) * 512;
}
int mgx_rel_de_from_dg(const void* dg, const void* qkv, float* slabs, int B, int S, int H, int dh, int nrb, hipStream_t st);   // gemm_tn.hip

namespace {
template <int DH>
int launch_rel_bwd(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                   float* dE, float* ws, int B, int S, int H, int max_seq, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)DH);
    const float c = LOG2E * scale;
    constexpr int NW = REL_NW;
    const int M = max_seq;
    const int nrb = (S + 31) / 32;
    float* delta = ws;
    float* slabs = ws + ((long)B * H * S + 63) / 64 * 64;
    char* dg = reinterpret_cast<char*>(slabs + (long)B * H * nrb * 32 * DH);
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
                       (const bf16_t*)dout, (const bf16_t*)qkv, lse, delta, (const bf16_t*)E, (bf16_t*)dqkv, dg, nrb, S, H,
                       M, c, scale);
    MGX_CHECK_LAUNCH("rel_bwd_dq_mfma");
    // dE: a batched, causal TN GEMM over the dG blocks (one slab per (b, h)), then the sum over b into the table
    if (int rc = mgx_rel_de_from_dg(dg, qkv, slabs, B, S, H, DH, nrb, st)) return rc;
    const long n4 = (long)H * M * DH / 4;
    hipLaunchKernelGGL(rel_de_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, slabs, dE, B, H, M,
                       nrb * 32, DH);
    MGX_CHECK_LAUNCH("rel_de_reduce");
    return MGX_OK;
}
}  // namespace

int mgx_rel_bwd_mfma(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                     float* dE, float* delta, int B, int S, int H, int dh, int max_seq, hipStream_t st) {
    if (dh == 64) return launch_rel_bwd<64>(dout, qkv, out, lse, E, dqkv, dE, delta, B, S, H, max_seq, st);
    return launch_rel_bwd<128>(dout, qkv, out, lse, E, dqkv, dE, delta, B, S, H, max_seq, st);
}
