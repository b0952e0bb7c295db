// bf16 MFMA GEMM, NT form, for the projections of the decoder block (gfx950):
//   C[M,N] = A[M,K] . Bt[N,K]^T  (+ bias[N]) (+ residual[M,N]) (-> activation / activation backward)
// The activation epilogue writes dropout(act(pre)) and, for the backward, the derivative factor act'(pre) * mask as ONE
// BYTE per element (a uniform code, see FacCode); the activation-backward dgrad is then (A . Bt^T) * fac / (1 - p): a
// byte conversion and two multiplies per element instead of erf + exp + the dropout hash (the epilogue is serial with
// the main loop on a CU), and half the bytes of a bf16 side tensor in both directions.  In the GELU forward epilogue
// both results of an element come from an LDS table indexed by the bf16 pre-activation (gelu_tab_off below).
// Replaces Conv1d.forward (base_layers.py:54-57: reshape -> matmul -> +b), the nn.Linear calls of
// DecoderBlock.forward (alibi_unified.py:90,92) and the dgrad GEMMs autograd derives from them.
//
// Tile 256 x 256 x 64, 8 waves.  A K tile is four 16-KiB "half tiles" (A rows 0-127 / 128-255, Bt rows
// 0-127 / 128-255), each staged by two LDS-DMA instructions per thread (global_load_lds_dwordx4, no VGPR
// round trip).  A K tile is computed in four phases, one 128 x 128 quadrant of the block tile per phase:
// every wave owns a 64 x 32 piece of each quadrant (16 v_mfma_f32_16x16x32_bf16 per phase), so a phase reads
// ONE A half tile and ONE Bt half tile and a half tile is dead as soon as the phase that moved it to
// registers is over.  Dead half tiles are restaged at once for the K tile two ahead: 4-5 half tiles
// (64-80 KiB) are in flight per CU, waited for with counted vmcnt (never 0 inside the loop) and raw
// s_barrier.  The two waves of every SIMD (waves w and w+4) run one barrier apart: while one issues its 16
// MFMAs the other issues its fragment reads and its LDS-DMA.  Rules the schedule obeys (program order of a
// wave: reads, stage, [wait], barrier a, MFMAs, barrier b):
//   WAR: a half tile is restaged >= 2 phases after the phase that read it;
//   RAW: a half tile is read >= 1 phase after the phase whose counted wait retired it.
// The epilogue leaves from the accumulators: v_permlane16_swap pairs make 16 contiguous bytes per lane
// (64-byte row segments per store instruction), no LDS and no barrier, so a workgroup's stores overlap the
// main loops of the others.
#include "mgx_common.h"
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int HT = 16384;                          // one half tile: 128 rows x 64 k x 2 B
constexpr int LDS_B = 65536;                       // Bt half tiles start here; slot (X, d) at (2X + d) * HT

struct NtArgs {
    const char* A;
    const char* B;
    bf16_t* C;               // [M,N]; EPI_ACT: the derivative-factor codes, one BYTE per element, row stride ldc bytes
    bf16_t* C2;              // second output (EPI_ACT) or null
    const float* bias;       // [N] or null
    const bf16_t* res;       // [M,N] or null (EPI_PLAIN)
    const unsigned char* fac;  // [M,N] derivative-factor codes written by the EPI_ACT forward (EPI_DFAC)
    float* colpart;          // [gridM, N] per-row-tile column sums of C (EPI_DFAC) or null
    int M, N, K;
    int lda, ldb, ldc;
    float p_drop;
    unsigned long long seed;
};

// first column (inside the block tile) of wave column wc's 32-column slice in Bt half tile X.  Measured and not kept:
// 64 (wc) + 32 (X), which makes a wave's two slices one 128-byte line of C (bias epilogue 208 -> 214 us, activation
// epilogue 320 -> 350 us at M=32768 N=4096 K=1024, same box)
#define NT_COL0(wc, X) (128 * (X) + 32 * (wc))
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DFAC = 2 };

// bijective XCD remap: blocks that share an XCD (id % 8) get a contiguous run of tile indices
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// gelu(x) = x * Phi(x) and d/dx gelu(x) = Phi(x) + x * phi(x) from one evaluation of erf (Abramowitz-Stegun 7.1.26,
// |error| < 1.5e-7) and of exp(-x^2 / 2).  Split in the part that depends on |x| only (the table build evaluates it
// once for +x and -x) and the signed tail.
struct GeluCore {
    float erf_abs, e;                                  // erf(|x| / sqrt 2), exp(-x^2 / 2)
};
__device__ __forceinline__ GeluCore gelu_core(float ax) {
    const float az = ax * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    GeluCore c;
    c.e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    c.erf_abs = fmaf(-poly * t, c.e, 1.f);
    return c;
}
__device__ __forceinline__ void gelu_tail(float x, const GeluCore& c, float& y, float& dy) {
    const float cdf = fmaf(0.5f, copysignf(c.erf_abs, x), 0.5f);
    y = x * cdf;
    dy = fmaf(x * 0.3989422804014327f, c.e, cdf);
}
__device__ __forceinline__ void gelu_val_grad(float x, float& y, float& dy) { gelu_tail(x, gelu_core(fabsf(x)), y, dy); }

// Derivative factor as one byte: act'(pre) * mask = (code - ZERO) * STEP.  GELU' lies in [-0.1288, 1.1288]: STEP ~ 0.005
// covers [-0.135, 1.139] with |error| <= 0.0025 (bf16 rounds a factor near 1 by up to 0.0039); its low 8 mantissa bits
// are cleared, so code * STEP and ZERO * STEP are exact in fp32 and code ZERO (a dropped element) decodes to exactly
// 0.  ReLU' is 0 or 1: STEP = 2^-7 makes both exact.  The 1 / (1 - p) of dropout is applied by the consumer.
template <int ACT>
struct FacCode {
    static constexpr int ZERO = 27;
    static constexpr float STEP = ACT == MGX_ACT_RELU ? 0.0078125f : __builtin_bit_cast(float, 0x3BA3D700u);
    // four factors -> four codes in one dword: v_cvt_pk_u8_f32 rounds to nearest, saturates to [0, 255] and drops the
    // byte into place (one fma + one conversion per element)
    __device__ static __forceinline__ unsigned enc4(const float (&f)[4]) {
        unsigned w = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) w = __builtin_amdgcn_cvt_pk_u8_f32(fmaf(f[r], 1.0f / STEP, (float)ZERO), r, w);
        return w;
    }
};

// GELU by table (EPI_ACT, bf16 pre-activations).  The epilogue's operand is the pre-activation ROUNDED TO bf16, so
// gelu(x) * keep_scale (bf16) and the code of gelu'(x) are functions of 16 bits: the workgroup tabulates them in its
// prologue (gelu_val_grad's own arithmetic, 7 pairs +-x per thread, under the wait for the first K tile) for
// 2^-20 <= |x| < 128 and the epilogue replaces erf + exp + encode (~20 VALU per element, the largest part of this epilogue:
// 58 of the 127 us it adds at M=32768 N=4096 K=1024) by one 4-byte LDS gather.  Entry = bf16(y) | code << 16; one more
// entry per sign holds the dropped element (+-0, code ZERO), so dropout is a select on the index.  Elements outside the
// table (|x| < 2^-20 or >= 128, inf, nan: about one per million) raise a flag and the wave redoes that 16 x 32 block on
// the arithmetic path: the results are bit-identical to that path everywhere.
constexpr int GT_LO = 107 << 7, GT_HI = (133 << 7) | 127, GT_NE = GT_HI - GT_LO + 1;   // exponent|mantissa of |x|
constexpr int GT_SIGN = 16384;                       // byte distance of the negative half (> 4 * (GT_NE + 1))
constexpr int GT_OFF = 2 * LDS_B;                    // the table follows the staging buffers
constexpr int GT_BYTES = GT_SIGN + 4 * (GT_NE + 1);
static_assert(4 * (GT_NE + 1) <= GT_SIGN && (GT_OFF & (2 * GT_SIGN - 1)) == 0, "the sign bit of the offset must be free");
static_assert(GT_OFF + GT_BYTES <= 160 * 1024, "LDS of a CU");
// byte offset (from smem) of the entry of the bf16 value with exponent|mantissa em and sign mask sgn (0 / -1); odd
// becomes non-zero for the values the table does not hold
__device__ __forceinline__ unsigned gelu_tab_off(unsigned em, int sgn, bool keep, unsigned& odd) {
    unsigned t = min(max(em, (unsigned)GT_LO), (unsigned)GT_HI);        // v_med3_u32
    odd |= t ^ em;                                                       // outside the table (one v_bitop3_b32)
    t = keep ? t : (unsigned)(GT_HI + 1);
    return ((unsigned)sgn & (unsigned)GT_SIGN) | ((t << 2) + (unsigned)(GT_OFF - 4 * GT_LO));
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
    union { bf16x2 v; unsigned u; } x;
    x.v[0] = (bf16_t)a;
    x.v[1] = (bf16_t)b;
    return x.u;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

#define NT_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define NT_LGKM0()                                             \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// HALF: the block tile is 128 x 256 (only the A0 half tile exists: the two quadrants of A1 are not computed, its
// staging instructions keep their place in the schedule but all read one 16-byte piece).  For launches whose 256-row
// tiles would leave a large part of the CUs idle (M = 8192 token rows against N = 1024: 128 tiles on 256 CUs).
template <int EPI, int ACT, bool DROP, bool HALF>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(NtArgs g) {
    constexpr int BMT = HALF ? 128 : BM, NQA = HALF ? 1 : 2;
    constexpr bool TAB = EPI == EPI_ACT && ACT == MGX_ACT_GELU;
    __shared__ __attribute__((aligned(16))) char smem[2 * LDS_B + (TAB ? GT_BYTES : 0)];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int l15 = lane & 15, g4 = lane >> 4;

    // tile order: bands of 8 tile rows, column-major inside a band, so that the ~32 workgroups that share
    // an XCD at any time cover a compact 8 x 4 rectangle (12 operand panels per K tile for 32 tiles)
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + BMT - 1) / BMT;
    const int tile = xcd_remap(blockIdx.x, nbm * nbn);
    const int band = tile / (8 * nbn);
    const int bh = min(8, nbm - 8 * band);
    const int rem = tile - band * 8 * nbn;
    const int bn = rem / bh, bm = 8 * band + (rem - bn * bh);
    const int m0 = bm * BMT, n0 = bn * BN;

    // ---- staging: per-lane source byte offsets (32 bit), [half X][piece i]; piece 2w+i = rows 16w+8i.. of the half tile
    unsigned sa[2][2], sb[2][2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * w + 8 * i + (lane >> 3);
            const int ch = (lane & 7) ^ ((row >> 1) & 7);
            sa[X][i] = (unsigned)(((long)min(m0 + 128 * X + row, g.M - 1) * g.lda + ch * 8) * 2);
            if (HALF && X == 1) sa[X][i] = (unsigned)((long)m0 * g.lda * 2);
            sb[X][i] = (unsigned)(((long)min(n0 + NT_COL0(row >> 5, X) + (row & 31), g.N - 1) * g.ldb + ch * 8) * 2);
        }
    const int nk = g.K / BK;
    char* const stage_base = smem + 2048 * w;
    auto stage_a = [&](int X, int d, int kt) {
        const char* src = g.A + (long)min(kt, nk - 1) * (BK * 2);
        char* dst = stage_base + (2 * X + d) * HT;
        __builtin_amdgcn_global_load_lds((const void*)(src + sa[X][0]), (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(src + sa[X][1]), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    auto stage_b = [&](int X, int d, int kt) {
        const char* src = g.B + (long)min(kt, nk - 1) * (BK * 2);
        char* dst = stage_base + LDS_B + (2 * X + d) * HT;
        __builtin_amdgcn_global_load_lds((const void*)(src + sb[X][0]), (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(src + sb[X][1]), (lds_void*)(dst + 1024), 16, 0, 0);
    };

    // ---- fragment reads: v_mfma_f32_16x16x32_bf16 operand = row (l & 15) of a 16-row block, 16-byte chunk
    // 4*ks + (l >> 4) of the 128-byte row; image: chunk index XORed with (row >> 1) & 7 (conflict-free b128 reads)
    const char* ard[2];
    const char* brd[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int lo = 1024 * (l15 >> 3) + 128 * (l15 & 7) + 16 * ((4 * ks + g4) ^ ((l15 >> 1) & 7));
        ard[ks] = smem + lo + 8192 * wr;                 // wave's 64 A rows of the half tile
        brd[ks] = smem + LDS_B + lo + 4096 * wc;         // wave's 32 Bt rows of the half tile
    }
    bf16x8 a[4][2], b0[2][2], b1[2][2];
    auto read_a = [&](int X, int d) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i][ks] = *reinterpret_cast<const bf16x8*>(ard[ks] + (2 * X + d) * HT + 2048 * i);
    };
    auto read_b = [&](bf16x8 (&b)[2][2], int X, int d) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j][ks] = *reinterpret_cast<const bf16x8*>(brd[ks] + (2 * X + d) * HT + 2048 * j);
    };

    f32x4 acc[2][2][4][2];                               // [qa][qb][16-row block of M][16-row block of N]
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[qa][qb][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the Bt rows go in as the MFMA "A" operand: D[row = n][col = m], so each lane (one C row m) holds 4
    // consecutive n per accumulator
#define NT_MMA(QA, QB, BF)                                                                                         \
    do {                                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                    acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], a[i][ks], acc[QA][QB][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                             \
    } while (0)

    // ---- prologue: K tile 0 whole, A0 / B0 of K tile 1
    stage_a(0, 0, 0);
    stage_b(0, 0, 0);
    stage_b(1, 0, 0);
    stage_a(1, 0, 0);
    stage_a(0, 1, 1);
    stage_b(0, 1, 1);
    if constexpr (TAB) {
        // the GELU table, while the first K tile is on its way; every wave passes an lgkmcnt(0) and a barrier in the
        // first K step, long before the epilogue reads it
        unsigned* const tab = reinterpret_cast<unsigned*>(smem + GT_OFF);
        const float ks = mgx_keep_scale(mgx_drop_thr16(g.p_drop));
#pragma unroll 1
        for (int em = tid; em < GT_NE; em += NTHREADS) {
            const float x = __uint_as_float((unsigned)(GT_LO + em) << 16);
            const GeluCore c = gelu_core(x);
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                float y, f;
                gelu_tail(sgn ? -x : x, c, y, f);
                if (DROP) y *= ks;
                const float fa[4] = {f, 0.f, 0.f, 0.f};
                tab[sgn * (GT_SIGN / 4) + em] = (pack2(y, 0.f) & 0xFFFFu) | ((FacCode<ACT>::enc4(fa) & 0xFFu) << 16);
            }
        }
        if (tid < 2) tab[tid * (GT_SIGN / 4) + GT_NE] = ((unsigned)tid << 15) | ((unsigned)FacCode<ACT>::ZERO << 16);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // A0(0), B0(0), B1(0) landed
    NT_BAR();
    if (wr == 1) NT_BAR();                                     // waves 4-7 run one barrier behind waves 0-3

    auto tile_step = [&](auto dtag, int t) {
        constexpr int d = decltype(dtag)::value;
        // phase 0: quadrant (A0, B0)
        read_a(0, d);
        read_b(b0, 0, d);
        stage_b(1, d ^ 1, t + 1);
        NT_BAR();
        NT_LGKM0();
        NT_MMA(0, 0, b0);
        NT_BAR();
        // phase 1: quadrant (A0, B1)
        read_b(b1, 1, d);
        stage_a(1, d ^ 1, t + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // A1(t) landed (read in phase 2)
        NT_BAR();
        NT_LGKM0();
        NT_MMA(0, 1, b1);
        NT_BAR();
        // phase 2: quadrant (A1, B1)
        if constexpr (!HALF) read_a(1, d);
        stage_a(0, d, t + 2);
        NT_BAR();
        NT_LGKM0();
        if constexpr (!HALF) NT_MMA(1, 1, b1);
        NT_BAR();
        // phase 3: quadrant (A1, B0)
        stage_b(0, d, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // A0, B0, B1 of K tile t+1 landed
        NT_BAR();
        if constexpr (!HALF) NT_MMA(1, 0, b0);
        NT_BAR();
    };
    for (int t = 0; t < nk; t += 2) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) tile_step(std::integral_constant<int, 1>{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the clamped re-loads of the last K tiles
    if (wr == 0) NT_BAR();
#undef NT_MMA

    // ---- epilogue from the accumulators -----------------------------------------------------------------
    // lane (m = l15, g4) holds n = 16 j + 4 g4 + r of the wave's 32 columns; after the 16-lane-row swap of the
    // packed j = 0 / j = 1 words, even g4 hold n = 4 g4 .. +8 and odd g4 hold n = 16 + 4 (g4 - 1) .. +8
    const unsigned thr16 = mgx_drop_thr16(g.p_drop);
    const float keep_scale = mgx_keep_scale(thr16);
    const int nsw = (g4 & 1) ? 16 + 4 * (g4 - 1) : 4 * g4;
    // the second operand of the epilogue (residual / derivative factor) for BOTH of the wave's 256 x 32 slices,
    // requested up front with clamped addresses (no branch: the 16 loads leave back to back into the registers the
    // fragments have left, and one round trip is exposed per tile), 16 bytes per lane in the layout of the stores
    // below and turned back into the accumulator layout by the same 16-lane-row swap
    const bool has_side = EPI == EPI_DFAC || (EPI == EPI_PLAIN && g.res);
    uint4 raw[2][2][4];                                 // EPI_DFAC uses .x / .y only: 8 codes per lane
    if (has_side) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int qa = 0; qa < NQA; ++qa)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int nc = min(n0 + NT_COL0(wc, qb) + nsw, g.N - 8);
                    const int mc = min(m0 + 128 * qa + 64 * wr + 16 * i + l15, g.M - 1);
                    if (EPI == EPI_DFAC) {
                        const uint2 t = *reinterpret_cast<const uint2*>(g.fac + (long)mc * g.ldc + nc);
                        raw[qb][qa][i] = make_uint4(t.x, t.y, 0u, 0u);
                    } else {
                        raw[qb][qa][i] = *reinterpret_cast<const uint4*>(g.res + (long)mc * g.ldc + nc);
                    }
                }
    }
    // EPI_DFAC: decode (code - ZERO) * STEP / (1 - p) as fma(code, fs, -ZERO * fs); ACT carries the code grid.  fs keeps
    // 16 mantissa bits, so code * fs and ZERO * fs are exact and a dropped element (code ZERO) decodes to exactly 0
    const float fs = __uint_as_float(__float_as_uint(FacCode<ACT>::STEP * keep_scale) & 0xFFFFFF00u);
    const float fz = -(float)FacCode<ACT>::ZERO * fs;
    f32x4 bv[2][2];
    float csum[2][2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bv[qb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + NT_COL0(wc, qb) + 16 * j + 4 * g4;
            if (EPI != EPI_DFAC && g.bias && n < g.N) bv[qb][j] = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[qb][j][r] = 0.f;
        }
    unsigned odd = 0;                                    // (TAB) non-zero: an element the table does not hold
    auto block = [&](int qa, int qb, int i, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int nb = n0 + NT_COL0(wc, qb);
        const int m = m0 + 128 * qa + 64 * wr + 16 * i + l15;
        const bool row_ok = m < g.M;
        const int mc = row_ok ? m : g.M - 1;
        float v[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = acc[qa][qb][i][j][r] + bv[qb][j][r];
        if (EPI == EPI_DFAC) {
            // C = v * fac: the lane's 8 codes go back to the accumulator layout by the 16-lane-row swap of the stores
            const uint4 rw = raw[qb][qa][i];
            auto sw = __builtin_amdgcn_permlane16_swap(rw.x, rw.y, false, false);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[j][r] *= fmaf((float)((sw[j] >> (8 * r)) & 0xFFu), fs, fz);
        } else if (has_side) {
            // residual: C = v + res (columns past N: some real value, never stored)
            const uint4 rw = raw[qb][qa][i];
            auto s0 = __builtin_amdgcn_permlane16_swap(rw.x, rw.z, false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(rw.y, rw.w, false, false);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float x[4] = {bf_lo(s0[j]), bf_hi(s0[j]), bf_lo(s1[j]), bf_hi(s1[j])};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] += x[r];
            }
        }
        unsigned o[2][2], o2[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            o[j][0] = pack2(v[j][0], v[j][1]);
            o[j][1] = pack2(v[j][2], v[j][3]);
        }
        if (EPI == EPI_DFAC) {
            // column sums of the ROUNDED values (what a separate pass over C would add up)
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    csum[qb][j][0] += bf_lo(o[j][0]);
                    csum[qb][j][1] += bf_hi(o[j][0]);
                    csum[qb][j][2] += bf_lo(o[j][1]);
                    csum[qb][j][3] += bf_hi(o[j][1]);
                }
            }
        }
        if (EPI == EPI_ACT) {
            // o2 = dropout(act(pre)), o[j][0] = the four codes of act'(pre) * mask; both see the pre-activation rounded
            // to bf16 (the value an unfused pass would read back)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nb + 16 * j + 4 * g4;
                if constexpr (FAST) {
                    bool k[4] = {true, true, true, true};
                    if (DROP) {
                        mgx_keep_pair(g.seed, (unsigned long long)((long)mc * g.N + n), thr16, k[0], k[1]);
                        mgx_keep_pair(g.seed, (unsigned long long)((long)mc * g.N + n + 2), thr16, k[2], k[3]);
                    }
                    unsigned e[4];
#pragma unroll
                    for (int hw = 0; hw < 2; ++hw) {
                        const unsigned wd = o[j][hw];
                        const unsigned a0 = gelu_tab_off(wd & 0x7FFFu, (int)(wd << 16) >> 31, k[2 * hw], odd);
                        const unsigned a1 = gelu_tab_off((wd >> 16) & 0x7FFFu, (int)wd >> 31, k[2 * hw + 1], odd);
                        e[2 * hw] = *reinterpret_cast<const unsigned*>(smem + a0);
                        e[2 * hw + 1] = *reinterpret_cast<const unsigned*>(smem + a1);
                    }
                    o2[j][0] = __builtin_amdgcn_perm(e[1], e[0], 0x05040100u);
                    o2[j][1] = __builtin_amdgcn_perm(e[3], e[2], 0x05040100u);
                    o[j][0] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e[3], e[2], 0x0C0C0602u),
                                                    __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0602u), 0x05040100u);
                    continue;
                }
                float y[4], f[4];
                const float x[4] = {bf_lo(o[j][0]), bf_hi(o[j][0]), bf_lo(o[j][1]), bf_hi(o[j][1])};
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float y0, y1, f0, f1;
                    if (ACT == MGX_ACT_RELU) {
                        y0 = fmaxf(x[r], 0.f);
                        y1 = fmaxf(x[r + 1], 0.f);
                        f0 = x[r] > 0.f ? 1.f : 0.f;
                        f1 = x[r + 1] > 0.f ? 1.f : 0.f;
                    } else {
                        gelu_val_grad(x[r], y0, f0);
                        gelu_val_grad(x[r + 1], y1, f1);
                    }
                    if (DROP) {
                        bool k0, k1;
                        mgx_keep_pair(g.seed, (unsigned long long)((long)mc * g.N + n + r), thr16, k0, k1);
                        y0 *= k0 ? keep_scale : 0.f;
                        f0 = k0 ? f0 : 0.f;
                        y1 *= k1 ? keep_scale : 0.f;
                        f1 = k1 ? f1 : 0.f;
                    }
                    y[r] = y0;
                    y[r + 1] = y1;
                    f[r] = f0;
                    f[r + 1] = f1;
                }
                o2[j][0] = pack2(y[0], y[1]);
                o2[j][1] = pack2(y[2], y[3]);
                o[j][0] = FacCode<ACT>::enc4(f);
            }
        }
        // 16-lane-row swap: odd rows of the j = 0 word <-> even rows of the j = 1 word
        const int n = nb + nsw;
        if (EPI == EPI_ACT) {
            auto sw = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
            if (row_ok && n < g.N)
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(g.C) + (long)m * g.ldc + n) = make_uint2(sw[0], sw[1]);
        } else {
            auto s0 = __builtin_amdgcn_permlane16_swap(o[0][0], o[1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(o[0][1], o[1][1], false, false);
            if (row_ok && n < g.N)
                *reinterpret_cast<uint4*>(g.C + (long)m * g.ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
        if (EPI == EPI_ACT) {
            auto s0 = __builtin_amdgcn_permlane16_swap(o2[0][0], o2[1][0], false, false);
            auto s1 = __builtin_amdgcn_permlane16_swap(o2[0][1], o2[1][1], false, false);
            if (row_ok && n < g.N)
                *reinterpret_cast<uint4*>(g.C2 + (long)m * g.ldc + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    };
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int qa = 0; qa < NQA; ++qa)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                odd = 0;
                block(qa, qb, i, std::bool_constant<TAB>{});
                if constexpr (TAB) {
                    // wave-uniform and rare: the block again on the arithmetic path (same stores, same values for the
                    // elements the table holds)
                    if (__builtin_amdgcn_ballot_w64(odd != 0) != 0) block(qa, qb, i, std::false_type{});
                }
            }
    if (EPI == EPI_DFAC && g.colpart) {
        // wave's partial column sums over its 128 rows of this tile: reduce the 16 lanes of a row group
        // (xor 1, 2, 4, 8), then lane l15 == 0 of each g4 adds 4 values per j.  Two wave rows (wr) share a
        // column: fixed slots [bm][wr][N], summed by nt_colpart_reduce_kernel in a fixed order.
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int nb = n0 + NT_COL0(wc, qb);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = csum[qb][j][r];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    csum[qb][j][r] = s;
                }
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = nb + 16 * j + 4 * g4;
                    if (n < g.N)
                        *reinterpret_cast<f32x4*>(g.colpart + ((long)bm * 2 + wr) * g.N + n) =
                            f32x4{csum[qb][j][0], csum[qb][j][1], csum[qb][j][2], csum[qb][j][3]};
                }
            }
        }
    }
}

// out[c] = sum_r part[r][c]: 64 columns x 16 row slices per workgroup, fixed summation order
__global__ __launch_bounds__(1024) void nt_colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                  int rows, int n) {
    __shared__ float sh[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < n) {
        int r = slice;
        for (; r + 16 < rows; r += 32) {
            s0 += part[(size_t)r * n + c];
            s1 += part[(size_t)(r + 16) * n + c];
        }
        if (r < rows) s0 += part[(size_t)r * n + c];
    }
    sh[slice][lane] = s0 + s1;
    __syncthreads();
    if (slice == 0 && c < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sh[k][lane];
        out[c] = s;
    }
}

}  // namespace

static int nt_check(const char* fn, const void* A, const void* Bt, const void* C, int M, int N, int K, int lda, int ldb,
                    int ldc) {
    MGX_REQUIRE(A && Bt && C, MGX_EINVAL, "%s: null pointer", fn);
    MGX_REQUIRE(M > 0 && N > 0 && K > 0, MGX_EINVAL, "%s: M=%d N=%d K=%d", fn, M, N, K);
    MGX_REQUIRE(K % BK == 0, MGX_EUNSUPPORTED, "%s: K=%d must be a multiple of %d", fn, K, BK);
    MGX_REQUIRE(N % 8 == 0, MGX_EUNSUPPORTED, "%s: N=%d must be a multiple of 8 (pad the weight)", fn, N);
    MGX_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N, MGX_EUNSUPPORTED,
                "%s: leading dimensions must be multiples of 8 (16-byte rows): lda=%d ldb=%d ldc=%d", fn, lda, ldb, ldc);
    MGX_REQUIRE((long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31), MGX_EUNSUPPORTED,
                "%s: operand larger than 4 GiB (32-bit staging offsets)", fn);
    const long nblk = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    MGX_REQUIRE(nblk < (1L << 31), MGX_EUNSUPPORTED, "%s: too many tiles", fn);
    return MGX_OK;
}

// 128-row tiles where 256-row tiles would fill at most three quarters of the current device's CUs (256 on MI355X)
static inline bool nt_half_tiles(int M, int N) {
    const long nblk = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int cus = mgx_cu_count() > 0 ? mgx_cu_count() : 256;
    return 4 * nblk <= 3L * cus && M > 128;
}
static inline long nt_blocks(int M, int N) {
    const int bm = nt_half_tiles(M, N) ? 128 : BM;
    return (long)((M + bm - 1) / bm) * ((N + BN - 1) / BN);
}
#define NT_LAUNCH(EPI, ACT, DROP)                                                                              \
    do {                                                                                                       \
        if (nt_half_tiles(M, N))                                                                               \
            hipLaunchKernelGGL((gemm_nt_kernel<EPI, ACT, DROP, true>), dim3((unsigned)nblk), dim3(NTHREADS), 0, \
                               (hipStream_t)stream, g);                                                        \
        else                                                                                                   \
            hipLaunchKernelGGL((gemm_nt_kernel<EPI, ACT, DROP, false>), dim3((unsigned)nblk), dim3(NTHREADS), 0, \
                               (hipStream_t)stream, g);                                                        \
    } while (0)

extern "C" int mgx_gemm_nt_bf16(const void* A, const void* Bt, void* C, void* C2, const float* bias, const void* res,
                                int M, int N, int K, int lda, int ldb, int ldc, int act, float p_drop,
                                unsigned long long seed, void* stream) {
    if (int rc = nt_check("mgx_gemm_nt_bf16", A, Bt, C, M, N, K, lda, ldb, ldc)) return rc;
    MGX_REQUIRE(act >= -1 && act <= MGX_ACT_RELU, MGX_EINVAL, "mgx_gemm_nt_bf16: bad act %d", act);
    MGX_REQUIRE(act < 0 || C2, MGX_EINVAL, "mgx_gemm_nt_bf16: activation requested without a second output");
    MGX_REQUIRE(act < 0 || !res, MGX_EINVAL, "mgx_gemm_nt_bf16: residual and activation epilogues exclude each other");
    MGX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, MGX_EINVAL, "mgx_gemm_nt_bf16: p_drop=%f", p_drop);
    NtArgs g{(const char*)A, (const char*)Bt, (bf16_t*)C, (bf16_t*)C2, bias, (const bf16_t*)res, nullptr, nullptr,
             M, N, K, lda, ldb, ldc, p_drop, seed};
    const long nblk = nt_blocks(M, N);
    const bool drop = p_drop > 0.f;
    if (act < 0) NT_LAUNCH(EPI_PLAIN, 0, false);
    else if (act == MGX_ACT_GELU) { if (drop) NT_LAUNCH(EPI_ACT, MGX_ACT_GELU, true); else NT_LAUNCH(EPI_ACT, MGX_ACT_GELU, false); }
    else { if (drop) NT_LAUNCH(EPI_ACT, MGX_ACT_RELU, true); else NT_LAUNCH(EPI_ACT, MGX_ACT_RELU, false); }
    MGX_CHECK_LAUNCH("gemm_nt");
    return MGX_OK;
}

// rows of colpart a call writes: two per row tile (one per wave row), the row tile being 128 or 256 rows
extern "C" int mgx_gemm_nt_dfac_colpart_rows(int M, int N) {
    return 2 * (nt_half_tiles(M, N) ? (M + 127) / 128 : (M + BM - 1) / BM);
}
extern "C" long mgx_gemm_nt_dfac_colpart_floats(int M, int N) { return (long)mgx_gemm_nt_dfac_colpart_rows(M, N) * N; }

extern "C" int mgx_gemm_nt_dfac_bf16(const void* A, const void* Bt, const void* fac, void* C, float* colpart, float* dbias,
                                     int M, int N, int K, int lda, int ldb, int ldc, int act, float p_drop, void* stream) {
    if (int rc = nt_check("mgx_gemm_nt_dfac_bf16", A, Bt, C, M, N, K, lda, ldb, ldc)) return rc;
    MGX_REQUIRE(fac, MGX_EINVAL, "mgx_gemm_nt_dfac_bf16: null derivative factor");
    MGX_REQUIRE(act == MGX_ACT_GELU || act == MGX_ACT_RELU, MGX_EINVAL, "mgx_gemm_nt_dfac_bf16: bad act %d", act);
    MGX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, MGX_EINVAL, "mgx_gemm_nt_dfac_bf16: p_drop=%f", p_drop);
    MGX_REQUIRE(!dbias || colpart, MGX_EINVAL, "mgx_gemm_nt_dfac_bf16: dbias needs the colpart workspace");
    NtArgs g{(const char*)A, (const char*)Bt, (bf16_t*)C, nullptr, nullptr, nullptr, (const unsigned char*)fac, colpart,
             M, N, K, lda, ldb, ldc, p_drop, 0ull};
    const long nblk = nt_blocks(M, N);
    if (act == MGX_ACT_GELU) NT_LAUNCH(EPI_DFAC, MGX_ACT_GELU, false);
    else NT_LAUNCH(EPI_DFAC, MGX_ACT_RELU, false);
    MGX_CHECK_LAUNCH("gemm_nt_dfac");
    if (dbias) {
        const int rows = mgx_gemm_nt_dfac_colpart_rows(M, N);
        hipLaunchKernelGGL(nt_colpart_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, (hipStream_t)stream, colpart,
                           dbias, rows, N);
        MGX_CHECK_LAUNCH("gemm_nt_dfac_reduce");
    }
    return MGX_OK;
}
