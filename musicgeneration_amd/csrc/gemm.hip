// bf16 MFMA GEMMs for the projections of the decoder block (gfx950):
//   NT:  C[M,N] = A[M,K] . Bt[N,K]^T  (+ bias[N]) (+ residual[M,N]) (-> activation)   forward / dgrad
//   TN:  C[M,N] = A[K,M]^T . B[K,N]   (split over K, fp32 partial slabs)               weight gradients
// Replaces Conv1d.forward (base_layers.py:54-57: reshape -> matmul -> +b) and the nn.Linear
// calls of DecoderBlock.forward (alibi_unified.py:90,92) plus their autograd.
//
// Tile 256 x 256 x 64, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 8 x 4 accumulators of
// v_mfma_f32_16x16x32_bf16 (128 accumulator registers; the chip holds a higher clock on this shape
// than on 32x32x16 at the same operand traffic: +12 % measured).  Operand tiles are staged by LDS-DMA
// (global_load_lds_dwordx4, no VGPR round trip), double-buffered (2 x 64 KiB).  Both kernels run the
// two waves of every SIMD one interval apart (one reads fragments while the other issues MFMAs, see
// the main loops).  LDS images: NT = 128-byte rows with the chunk index XORed by (row>>1)&7 (full
// cache lines per DMA piece, conflict-free ds_read_b128); TN = 8-row x 32-column sub-tiles with an
// XOR (the attention kernels' image) for conflict-free ds_read_b64_tr_b16 transposed reads.  The NT
// epilogue goes through LDS so that C is written in full 512-byte row segments with 16-byte stores,
// with bias / residual / GELU / dropout applied on the way.  Workgroup ids are remapped so that each
// XCD works on a compact set of tiles that share operand panels (L2 reuse).
// The TN kernel stays on v_mfma_f32_32x32x16_bf16: on 16x16x32 it was no faster (data-movement bound) and
// its transposed-read addressing cost three spilled registers.
#include "mgx_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int TILE_BYTES = 256 * BK * 2;          // one operand tile: 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A + B
constexpr int CPITCH = 64 * 2 + 16;               // epilogue staging: 64 bf16 per row + pad (16-B aligned)

// ---- LDS images -----------------------------------------------------------------------------
// K-contiguous operand tile [256 rows][64 k] (NT): plain 128-byte rows, 16-byte chunk index XORed
// with (row>>1)&7.  One LDS-DMA wave-instruction (1 KiB) = 8 whole rows = 8 full 128-byte cache
// lines, each fetched by 8 adjacent lanes; ds_read_b128 row reads of 32 rows are conflict-free
// (two rows per 256-byte bank row, 8 chunk slots each).
__device__ __forceinline__ int img64_off(int row, int ch) {
    return 1024 * (row >> 3) + 128 * (row & 7) + 16 * (ch ^ ((row >> 1) & 7));
}
// K-major operand tile [64 k rows][256 cols] as two [64][128] images (16 chunks per row each)
__device__ __forceinline__ int img128_off(int row, int ch) {
    return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

__device__ __forceinline__ bf16x8 lds16(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds_tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    union {
        struct { s16x4 lo, hi; } s;
        bf16x8 v;
    } u;
    u.s.lo = a;
    u.s.hi = b;
    return u.v;
}

__device__ const bf16_t g_zero16[8] = {};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    void* C;                 // bf16 [M,N] (NT) or fp32 slabs [splits][M,N] (TN)
    bf16_t* C2;              // optional second output (activation), NT only
    const float* bias;       // [N] or null
    const bf16_t* res;       // [M,N] or null
    int M, N, K;
    int lda, ldb, ldc;
    int act;                 // -1 none, else MGX_ACT_*: C = pre-activation, C2 = dropout(act(C))
    float p_drop;
    unsigned long long seed;
    int ksplit_len;          // TN: K elements per split
};

// bijective XCD remap: blocks that share an XCD (id % 8) get a contiguous run of tile indices
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ float gelu_fast(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    const float erfv = copysignf(fmaf(-poly * t, e, 1.f), z);
    return 0.5f * x * (1.f + erfv);
}

// =============================================================================================
// NT kernel
// =============================================================================================
// ACT: -1 none, MGX_ACT_*; DROP: dropout after the activation.  Compile-time: with run-time switches
// hipcc branched per element in the epilogue.
template <int ACT, bool DROP>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, nbm * nbn);
    const int bm = tile / nbn, bn = tile - bm * nbn;         // n fastest: neighbours share the A panel
    const int m0 = bm * BM, n0 = bn * BN;

    // Staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction writes 1 KiB = one 8-row
    // group of the image, linearly (LDS address = wave-uniform base + 16*lane), so the image's
    // swizzle is applied to the per-lane SOURCE address: lane -> 16-B slot -> (row, chunk).
    // 32 pieces per operand tile, 4 per wave.
    const bf16_t* ap[4];
    const bf16_t* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = 4 * w + i;                          // 8-row group of the tile
        const int row = 8 * piece + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        ap[i] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + ch * 8;
        bp[i] = g.B + (long)min(n0 + row, g.N - 1) * g.ldb + ch * 8;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    const int nk_total = g.K / BK;
    auto stage = [&](int k0, int buf) {
#ifdef MGX_GEMM_NOLOAD                      // timing experiment only: results are wrong
        if (k0 > 0) return;
#endif
#ifdef MGX_GEMM_SAMETILE                    // timing experiment only: every load re-reads K tile 0 (cache hits)
        k0 = 0;
#endif
        char* sa = smem + buf * STAGE_BYTES + 4096 * w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const void*)(ap[i] + k0), (lds_void*)(sa + 1024 * i), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)(bp[i] + k0), (lds_void*)(sa + TILE_BYTES + 1024 * i), 16, 0, 0);
        }
    };

    // v_mfma_f32_16x16x32_bf16 (two 32-deep k steps per K tile): lane l holds row (l & 15) of a 16-row block,
    // k = 8 * (l >> 4) .. +7 of the step = 16-byte chunk 4 * step + (l >> 4).  Same operand bytes as the
    // 32x32x16 shape, but the chip holds a higher clock on it: +12 % on this kernel at equal instruction
    // traffic (timing experiment of round 1; MI355X_MICROARCH.md "DVFS give-back" item 7).
    const int l15 = lane & 15, g4 = lane >> 4;
    int rk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rk[ks] = img64_off(l15, 4 * ks + g4);
    const int a_base = 1024 * (16 * wm);               // wave's first A row: 128 * wm (8 rows per KiB)
    const int b_base = TILE_BYTES + 1024 * (8 * wn);   // wave's first Bt row: 64 * wn

    f32x4 acc[8][4];                                   // [16-row block of M][16-row block of N]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Main loop: the two waves of each SIMD (wave groups 0-3 and 4-7) run half a K tile apart.
    // Between two consecutive barriers one group is in its READ interval (24 ds_read_b128: every
    // fragment of the K tile into registers, plus its LDS-DMA issue) while the other group is in its
    // MFMA interval (32 back-to-back MFMAs on registers only), then they swap: the matrix pipe of
    // every SIMD always has one wave feeding it.  Raw s_barrier (no vmcnt drain): tile t+1's LDS-DMA
    // is issued two intervals before its first read and each wave waits for its own pieces
    // (vmcnt(0)) right before the barrier that precedes that read.
    const int nk = nk_total;
    const int grp = __builtin_amdgcn_readfirstlane(w) >> 2;
    stage(0, 0);
    if (grp == 1 && nk > 1) {
        stage(BK, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // tile 0 landed; tile 1's 8 pieces may still fly
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                  // stagger by one interval
    for (int kt = 0; kt < nk; ++kt) {
        const char* s = smem + (kt & 1) * STAGE_BYTES;
        // ---- READ interval
        bf16x8 bf[2][4], af[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ko = rk[ks];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[ks][j] = lds16(s + b_base + 2048 * j + ko);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[ks][i] = lds16(s + a_base + 2048 * i + ko);
        }
        // group 0 issues its LDS-DMA behind its own fragment reads (their latency covers the issue cost)
        if (grp == 0 && kt + 1 < nk) {
            stage((kt + 1) * BK, (kt + 1) & 1);
        }
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA interval (group 1 interleaves its LDS-DMA issue with the MFMAs)
        const bool g1stage = grp == 1 && kt + 2 < nk;
        __builtin_amdgcn_s_setprio(1);
        // the Bt rows go in as the MFMA "A" operand: D[row = n][col = m], so each lane (= one C row)
        // holds 4 consecutive n per accumulator
#pragma unroll
        for (int q = 0; q < 4; ++q) {                          // quarter of the K tile's MFMAs: step q >> 1, M blocks 4*(q&1)..
            const int ks = q >> 1;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * (q & 1) + ii;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
                }
            if (g1stage) {
                char* sa = smem + (kt & 1) * STAGE_BYTES + 4096 * w;
                const int k0 = (kt + 2) * BK;
                __builtin_amdgcn_global_load_lds((const void*)(ap[q] + k0), (lds_void*)(sa + 1024 * q), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const void*)(bp[q] + k0), (lds_void*)(sa + TILE_BYTES + 1024 * q), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---- epilogue through LDS: wave-private [64 rows][64 cols] slab, two passes (mi pairs) ----
    char* cs = smem + w * (64 * CPITCH);
    constexpr bool has_act = ACT >= 0;
    const unsigned thr16 = mgx_drop_thr16(g.p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * pass + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (bf16_t)acc[i][j][k];
                // row = 16*ii + l15 (m), col = 16*j + 4*g4 (n)
                *reinterpret_cast<bf16x4*>(cs + (16 * ii + l15) * CPITCH + (16 * j + 4 * g4) * 2) = v;
            }
        }
        // wave-private region: no barrier needed, only the wave's own LDS ordering
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        // read back: 8 lanes per row (16 B each), 8 rows per wave-instruction, 8 instructions
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 8 * it + (lane >> 3), c8 = (lane & 7) * 8;
            const int m = m0 + 128 * wm + 64 * pass + row;
            const int n = n0 + 64 * wn + c8;
            bf16x8 raw = *reinterpret_cast<const bf16x8*>(cs + row * CPITCH + c8 * 2);
            if (m < g.M && n < g.N) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (float)raw[k];
                const bool full = n + 8 <= g.N;
                if (g.bias) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += g.bias[min(n + k, g.N - 1)];
                }
                if (g.res) {
                    if (full) {
                        const bf16x8 rr = *reinterpret_cast<const bf16x8*>(g.res + (long)m * g.ldc + n);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += (float)rr[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (n + k < g.N) v[k] += (float)g.res[(long)m * g.ldc + n + k];
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (bf16_t)v[k];
                bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
                if (full) {
                    *reinterpret_cast<bf16x8*>(cp) = o;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (n + k < g.N) cp[k] = o[k];
                }
                if (has_act) {
                    bf16x8 o2;
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        // the activation sees the stored (rounded) pre-activation, as the unfused kernel does
                        float x0 = (float)o[k], x1 = (float)o[k + 1];
                        float y0 = ACT == MGX_ACT_RELU ? fmaxf(x0, 0.f) : gelu_fast(x0);
                        float y1 = ACT == MGX_ACT_RELU ? fmaxf(x1, 0.f) : gelu_fast(x1);
                        if (DROP) {
                            bool k0, k1;
                            mgx_keep_pair(g.seed, (unsigned long long)((long)m * g.N + n + k), thr16, k0, k1);
                            y0 = k0 ? y0 * keep_scale : 0.f;
                            y1 = k1 ? y1 * keep_scale : 0.f;
                        }
                        o2[k] = (bf16_t)y0;
                        o2[k + 1] = (bf16_t)y1;
                    }
                    bf16_t* cp2 = g.C2 + (long)m * g.ldc + n;
                    if (full) {
                        *reinterpret_cast<bf16x8*>(cp2) = o2;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (n + k < g.N) cp2[k] = o2[k];
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
}

// =============================================================================================
// TN kernel (weight gradients): C_split[M,N] = sum_{k in split} A[k,m] * B[k,n], fp32 slabs
// =============================================================================================
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, h5 = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BM - 1) / BM;
    const int ntile = nbm * nbn;
    // Workgroups that share an XCD (blockIdx % 8) take a contiguous run of (split, tile) pairs, the
    // smaller tile dimension fastest: the ~32 workgroups resident on one XCD then cover a compact
    // rectangle of tiles of ONE K split and walk K together, so every A / B panel slice is fetched
    // into that XCD's L2 once and re-used by the rectangle's row / column (the operand stream of
    // this kernel is otherwise ~2 GB per launch from beyond L2).
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = wid / ntile;
    const int tile = wid - split * ntile;
    int bm, bn;
    if (nbm <= nbn) {
        bn = tile / nbm;
        bm = tile - bn * nbm;
    } else {
        bm = tile / nbn;
        bn = tile - bm * nbn;
    }
    const int m0 = bm * BM, n0 = bn * BN;
    const int kbeg = split * g.ksplit_len;
    const int kend = min(g.K, kbeg + g.ksplit_len);

    // LDS-DMA staging: tile [64 k][256 cols] = two [64][128] images of 16 pieces (1 KiB) each;
    // piece p covers rows 8*(p>>1).., column chunks 8*(p&1)..; 4 pieces per operand per wave.
    int arow[4], acol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = 4 * w + i;                          // 0..31: image half = piece >> 4
        const int pp = piece & 15;
        const int row = 8 * (pp >> 1) + ((lane >> 2) & 7);
        const int ch = 8 * (pp & 1) + 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));
        arow[i] = row;
        acol[i] = (piece >> 4) * 128 + ch * 8;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto stage_piece = [&](int k0, int buf, int i) {
        char* sa = smem + buf * STAGE_BYTES + 4096 * w;
        // rows past the end of this split read a 16-byte block of zeros instead
        const bool ok = k0 + arow[i] < kend;
        const bf16_t* pa = ok ? g.A + (long)(k0 + arow[i]) * g.lda + min(m0 + acol[i], g.M - 8) : g_zero16;
        const bf16_t* pb = ok ? g.B + (long)(k0 + arow[i]) * g.ldb + min(n0 + acol[i], g.N - 8) : g_zero16;
        __builtin_amdgcn_global_load_lds((const void*)pa, (lds_void*)(sa + 1024 * i), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)pb, (lds_void*)(sa + TILE_BYTES + 1024 * i), 16, 0, 0);
    };
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece(k0, buf, i);
    };

    // transposed-read lane offsets (see attn_mfma.hip LaneAddr): columns on the lanes, k reduced
    const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
    const int x = 2 * g1 + (p >> 1);
    const int rbase = 64 * (4 * h5 + q) + 8 * (p & 1);
    const int tr0 = rbase + 16 * (x ^ h5), tr1 = rbase + 16 * ((x ^ h5) ^ 2);
    // wave's columns: A image half wm (128 m columns = 4 blocks of 32), B columns 64*wn .. +64
    const int a_base = wm * (TILE_BYTES / 2);
    const int b_base = TILE_BYTES + (wn >> 1) * (TILE_BYTES / 2) + 512 * (2 * (wn & 1));

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Main loop: same two-group schedule as gemm_nt_kernel (waves 0-3 and 4-7 run one interval apart:
    // one group reads every fragment of its K tile into registers while the other issues 32 MFMAs);
    // the fragment reads are transposed (ds_read_b64_tr_b16 pairs) because both operands are K-major.
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int grp = __builtin_amdgcn_readfirstlane(w) >> 2;
    stage(kbeg, 0);
    if (grp == 1 && nk > 1) {
        stage(kbeg + BK, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const char* s = smem + (kt & 1) * STAGE_BYTES;
        // ---- READ interval
        bf16x8 bf[4][2], af[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                      // 16 k per step: 8-row groups 2ks, 2ks+1
            const int k0o = 2048 * (2 * ks), k1o = 2048 * (2 * ks + 1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[ks][j] = lds_tr_pair(s + b_base + 512 * j + tr0 + k0o, s + b_base + 512 * j + tr1 + k1o);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[ks][i] = lds_tr_pair(s + a_base + 512 * i + tr0 + k0o, s + a_base + 512 * i + tr1 + k1o);
        }
        if (grp == 0 && kt + 1 < nk) stage(kbeg + (kt + 1) * BK, (kt + 1) & 1);
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA interval (group 1 issues the LDS-DMA of tile kt+2 between its MFMAs)
        const bool g1stage = grp == 1 && kt + 2 < nk;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            if (g1stage) stage_piece(kbeg + (kt + 2) * BK, kt & 1, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    // fp32 slab store: D[row = n][col = m]: lane = m, 4 consecutive n per register group (16-B stores)
    float* C = reinterpret_cast<float*>(g.C) + (long)split * g.M * g.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 128 * wm + 32 * i + l31;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + 64 * wn + 32 * j + 8 * gq + 4 * h5;
                if (n + 4 <= g.N) {
                    f32x4 v = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                    *reinterpret_cast<f32x4*>(C + (long)m * g.ldc + n) = v;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (n + k < g.N) C[(long)m * g.ldc + n + k] = acc[i][j][4 * gq + k];
                }
            }
    }
}

// out[i] = sum_s slabs[s][i]  (fixed order: deterministic), fp32 or bf16 output
template <typename O>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, O* __restrict__ out,
                                                             long n, int splits) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 256 * 4) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
        for (int k = 1; k < splits; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(slabs + (long)k * n + i);
            s += t;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) out[i + k] = from_f32<O>(s[k]);
    }
}

}  // namespace

extern "C" int mgx_gemm_nt_r1_bf16(const void* A, const void* Bt, void* C, void* C2, const float* bias, const void* res,
                                int M, int N, int K, int lda, int ldb, int ldc, int act, float p_drop,
                                unsigned long long seed, void* stream) {
    MGX_REQUIRE(A && Bt && C, MGX_EINVAL, "mgx_gemm_nt_bf16: null pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0, MGX_EINVAL, "mgx_gemm_nt_bf16: M=%d N=%d K=%d", M, N, K);
    MGX_REQUIRE(K % BK == 0, MGX_EUNSUPPORTED, "mgx_gemm_nt_bf16: K=%d must be a multiple of %d", K, BK);
    MGX_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N, MGX_EUNSUPPORTED,
                "mgx_gemm_nt_bf16: leading dimensions must be multiples of 8 (16-byte rows): lda=%d ldb=%d ldc=%d",
                lda, ldb, ldc);
    MGX_REQUIRE(act >= -1 && act <= MGX_ACT_RELU, MGX_EINVAL, "mgx_gemm_nt_bf16: bad act %d", act);
    MGX_REQUIRE(act < 0 || C2, MGX_EINVAL, "mgx_gemm_nt_bf16: activation requested without a second output");
    MGX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, MGX_EINVAL, "mgx_gemm_nt_bf16: p_drop=%f", p_drop);
    GemmArgs g{(const bf16_t*)A, (const bf16_t*)Bt, C, (bf16_t*)C2, bias, (const bf16_t*)res, M, N, K, lda, ldb, ldc,
               act, p_drop, seed, 0};
    const long nblk = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    MGX_REQUIRE(nblk < (1L << 31), MGX_EUNSUPPORTED, "mgx_gemm_nt_bf16: too many tiles");
#define MGX_NT(ACT, DROP) \
    hipLaunchKernelGGL((gemm_nt_kernel<ACT, DROP>), dim3((unsigned)nblk), dim3(NTHREADS), 0, (hipStream_t)stream, g)
    const bool drop = p_drop > 0.f;
    if (act < 0) MGX_NT(-1, false);
    else if (act == MGX_ACT_GELU) { if (drop) MGX_NT(MGX_ACT_GELU, true); else MGX_NT(MGX_ACT_GELU, false); }
    else { if (drop) MGX_NT(MGX_ACT_RELU, true); else MGX_NT(MGX_ACT_RELU, false); }
#undef MGX_NT
    MGX_CHECK_LAUNCH("gemm_nt");
    return MGX_OK;
}

extern "C" long mgx_gemm_tn_workspace_floats(int M, int N, int K, int* splits_out) {
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // one workgroup per CU (128 KiB of LDS each) and equal work per workgroup: a single round of at
    // most 256 workgroups has no tail and writes the fewest fp32 slabs
    int splits = (int)(256 / tiles);
    const int max_splits = (K + 4 * BK - 1) / (4 * BK);     // at least 4 K tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits_out) *splits_out = splits;
    return (long)splits * M * N;
}

extern "C" int mgx_gemm_tn_bf16(const void* A, const void* B, void* C, float* workspace, int M, int N, int K, int lda,
                                int ldb, int out_dtype, void* stream) {
    MGX_REQUIRE(A && B && C && workspace, MGX_EINVAL, "mgx_gemm_tn_bf16: null pointer");
    MGX_REQUIRE(M >= 8 && N >= 8 && K > 0, MGX_EINVAL, "mgx_gemm_tn_bf16: M=%d N=%d K=%d", M, N, K);
    MGX_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N, MGX_EUNSUPPORTED,
                "mgx_gemm_tn_bf16: M, N and leading dimensions must be multiples of 8: M=%d N=%d lda=%d ldb=%d", M, N,
                lda, ldb);
    MGX_REQUIRE(out_dtype == MGX_F32 || out_dtype == MGX_BF16, MGX_EINVAL, "mgx_gemm_tn_bf16: bad out dtype");
    int splits = 1;
    mgx_gemm_tn_workspace_floats(M, N, K, &splits);
    int len = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (K + len - 1) / len;
    GemmArgs g{(const bf16_t*)A, (const bf16_t*)B, workspace, nullptr, nullptr, nullptr, M, N, K, lda, ldb, N,
               -1, 0.f, 0ull, len};
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * splits)), dim3(NTHREADS), 0, st, g);
    MGX_CHECK_LAUNCH("gemm_tn");
    const long n = (long)M * N;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (out_dtype == MGX_F32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, workspace, (float*)C, n, splits);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, workspace, (bf16_t*)C, n, splits);
    MGX_CHECK_LAUNCH("splitk_reduce");
    return MGX_OK;
}
