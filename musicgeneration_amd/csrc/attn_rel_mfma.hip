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

#ifndef MGX_REL_NW
#define MGX_REL_NW 4
#endif
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
    constexpr int NW = MGX_REL_NW;
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

bool mgx_rel_bwd_mfma_ready() { return false; }

int mgx_rel_bwd_mfma(const void* dout, const void* qkv, const void* out, const float* lse, const void* E, void* dqkv,
                     float* dE, float* delta, int B, int S, int H, int dh, int max_seq, hipStream_t st) {
    MGX_REQUIRE(false, MGX_EUNSUPPORTED, "mgx_rel_bwd_mfma: not built");
    return MGX_OK;
}
