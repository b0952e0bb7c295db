// bf16 MFMA GEMM, TN form, for the weight gradients of every projection (gfx950):
//   C[M,N] = A[K,M]^T . B[K,N]     K = token rows (32,768 in the training step), split over workgroups into fp32
//   slabs that a second kernel adds in a fixed order (deterministic, no atomics).
// Autograd of Conv1d.forward (base_layers.py:54-57) and of the nn.Linear calls of DecoderBlock.forward
// (alibi_unified.py:90,92): dW = x^T . dy.
//
// Same schedule as gemm_nt.hip (tile 256 x 256 x 64, four 16-KiB half tiles per K tile, one 128 x 128 quadrant
// per phase, half tiles restaged as soon as they are dead, counted vmcnt + raw s_barrier, the two waves of a SIMD
// one barrier apart).  Both operands are K-major here, so a half tile is [64 k][128 columns] stored as 8-row x
// 32-column sub-tiles of 512 B with an XOR on the 16-byte slot, and MFMA fragments are read with
// ds_read_b64_tr_b16 pairs (conflict-free on this image for the 16x16x32 operand).
#include "mgx_common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int HT = 16384;
constexpr int LDS_B = 65536;

struct TnProb {
    const char* A;
    const char* B;
    float* C;                // fp32 slabs [splits][M, N]
    int M, N;
    int lda, ldb;
};
struct TnArgs {
    TnProb p[2];             // MODE 0: one or two problems share the launch (tiles of p[0] first); MODE 1: p[0]
    int tiles0;              // tiles of p[0]; 0 tiles of p[1] when there is one problem
    int tiles1;
    int K;
    int ksplit_len;          // K elements per split (multiple of 64)
    // MODE 1 (dE of the skew relative attention from the dG blocks the dQ kernel stored, attn_rel_mfma.hip):
    // A = blocked dG, B = the packed qkv (q third of b = 0, h = 0), C = fp32 slabs [B*H][M][N]; M = padded number
    // of distances, N = dh, K = S (queries)
    int bk_bh, bk_h;           // B*H, H
    long bk_blocks;            // 2-KiB dG blocks per (b, h): mgx_rel_dg_row(nrb), the staircase layout of attn_rel_mfma.hip
};

__device__ const bf16_t g_tn_zero16[8] = {};

__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ds_read_b64_tr_b16 as inline asm: behind the builtin hipcc (ROCm 7.2) waits vmcnt(0) before the first transposed
// read after an LDS-DMA (it cannot tell that the read does not alias the DMA's destination), which drains the whole
// staging pipeline every K tile.  The asm form is invisible to that pass; its data is waited for by the explicit
// s_waitcnt lgkmcnt(0) + sched_barrier(0) that follows every group of reads (TN_LGKM0).
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(unsigned long)(lds_char*)p; }
template <int OFF>
__device__ __forceinline__ s16x4 tr64(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_tr_pair(unsigned a0, unsigned a1) {
    union {
        struct { s16x4 lo, hi; } s;
        bf16x8 v;
    } u;
    u.s.lo = tr64<OFF>(a0);
    u.s.hi = tr64<OFF>(a1);
    return u.v;
}
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#define TN_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define TN_LGKM0()                                             \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

// TAIL: the split's K range may end inside a K tile (rows past the end are read from a block of zeros)
// MODE 0: weight gradients (split K, fp32 slabs).
// MODE 1: dEr[b,h][r, :] = sum_i dG[b,h][i, r] . q[b,i,h, :] of the skew relative attention, from the dG blocks the dQ
//   kernel stored: [32-query unit][32-distance block][2 distance halves][32 queries][16 distances] bf16, 2 KiB each,
//   i.e. K-major in both operands (this kernel's form) with a blocked address map on the A side.  One workgroup per
//   (b, h, 256-distance tile), longest K range first; the K range starts at the tile's first distance (r <= i);
//   only the first dh columns of the block tile exist, so the two quadrants of the second B half tile are not
//   computed; fp32 result rows go to the slab of (b, h), summed over b by rel_de_reduce_kernel (attn_rel_mfma.hip).
template <bool TAIL, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(TnArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * LDS_B];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int l15 = lane & 15, g4 = lane >> 4;

    // Workgroups that share an XCD (blockIdx % 8) take a contiguous run of (split, tile) pairs, the smaller tile
    // dimension fastest: the ~32 workgroups resident on one XCD cover a compact rectangle of tiles of ONE K split
    // and walk K together, so every A / B panel slice is fetched into that XCD's L2 once.
    int split = 0, m0, n0, kbeg, kend;
    int sel = 0;
    if constexpr (MODE == 0) {
        // two problems in one launch (the weight gradients of linear1 + linear2, of c_attn + c_proj): together they have
        // enough tiles for half as many K splits, i.e. half the fp32 slab traffic per gradient
        const int ntile = g.tiles0 + g.tiles1;
        const int wid = xcd_remap(blockIdx.x, gridDim.x);
        split = wid / ntile;
        int tile = wid - split * ntile;
        sel = tile >= g.tiles0;
        tile -= sel ? g.tiles0 : 0;
        const int pM = sel ? g.p[1].M : g.p[0].M, pN = sel ? g.p[1].N : g.p[0].N;
        const int nbn = (pN + BN - 1) / BN, nbm = (pM + BM - 1) / BM;
        int bm, bn;
        if (nbm <= nbn) {
            bn = tile / nbm;
            bm = tile - bn * nbm;
        } else {
            bm = tile / nbn;
            bn = tile - bm * nbn;
        }
        m0 = bm * BM;
        n0 = bn * BN;
        kbeg = split * g.ksplit_len;
        kend = min(g.K, kbeg + g.ksplit_len);
    }
    const TnProb P = sel ? g.p[1] : g.p[0];
    const char* Ab = P.A;
    const char* Bb = P.B;
    if constexpr (MODE == 1) {
        // dispatch order: every (b, h) of the tile row with the longest K range first; (b, h) % 8 = the XCD, the
        // same for all of its tiles (its q rows stay in that XCD's L2)
        const int ord = blockIdx.x / g.bk_bh, bh = blockIdx.x - ord * g.bk_bh;
        m0 = ord * BM;
        n0 = 0;
        kbeg = m0;
        kend = g.K;
        split = bh;
        const int b = bh / g.bk_h, h = bh - b * g.bk_h;
        Ab += (long)bh * g.bk_blocks * 2048;
        Bb += ((long)b * g.K * P.ldb + (long)h * P.N) * 2;
    }
    const int nk = (kend - kbeg + BK - 1) / BK;

    // ---- staging: half tile X = [64 k][128 columns]; piece 2w+i = k rows 8w.., 16-byte chunks 8i..8i+7
    const int srow = 8 * w + ((lane >> 2) & 7);
    unsigned sa[2][2], sb[2][2];
#pragma unroll
    for (int X = 0; X < 2; ++X)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = 8 * i + 4 * (lane >> 5) + ((lane & 3) ^ ((srow >> 2) & 3));
            if constexpr (MODE == 0) {
                sa[X][i] = (unsigned)(((long)srow * P.lda + min(m0 + 128 * X + ch * 8, P.M - 8)) * 2);
            } else {
                // query row srow of the K tile: unit srow >> 5, row srow & 31 of the block; 8 distances of block m >> 5
                const int m = min(m0 + 128 * X + ch * 8, P.M - 8);
                // (the unit's row of blocks is added per stage: mgx_rel_dg_row, wave-uniform)
                sa[X][i] = (unsigned)(((m >> 5) * 2048) + ((m >> 4) & 1) * 1024 + (srow & 31) * 32 + (m & 15) * 2);
            }
            sb[X][i] = (unsigned)(((long)srow * P.ldb + min(n0 + 128 * X + ch * 8, P.N - 8)) * 2);
        }
    char* const stage_base = smem + 2048 * w;
    auto stage_a = [&](int X, int d, int kt) {
        const int k0 = kbeg + min(kt, nk - 1) * BK;
        // MODE 1: query unit (k0 >> 5) + (srow >> 5), and srow >> 5 == w >> 2
        const char* src = MODE == 0 ? Ab + (long)k0 * P.lda * 2 : Ab + mgx_rel_dg_row((k0 >> 5) + (w >> 2)) * 2048L;
        char* dst = stage_base + (2 * X + d) * HT;
        const char* p0 = src + sa[X][0];
        const char* p1 = src + sa[X][1];
        if (TAIL && k0 + srow >= kend) p0 = p1 = (const char*)g_tn_zero16;
        if constexpr (MODE == 1) {
            // the dG blocks are read exactly once: non-temporal (aux = 2), the q rows keep the L2 to themselves
            __builtin_amdgcn_global_load_lds((const void*)p0, (lds_void*)(dst), 16, 0, 2);
            __builtin_amdgcn_global_load_lds((const void*)p1, (lds_void*)(dst + 1024), 16, 0, 2);
            return;
        }
        __builtin_amdgcn_global_load_lds((const void*)p0, (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)p1, (lds_void*)(dst + 1024), 16, 0, 0);
    };
    auto stage_b = [&](int X, int d, int kt) {
        const int k0 = kbeg + min(kt, nk - 1) * BK;
        const char* src = Bb + (long)k0 * P.ldb * 2;
        char* dst = stage_base + LDS_B + (2 * X + d) * HT;
        const char* p0 = src + sb[X][0];
        const char* p1 = src + sb[X][1];
        if (TAIL && k0 + srow >= kend) p0 = p1 = (const char*)g_tn_zero16;
        if (MODE == 1 && X == 1) p0 = p1 = (const char*)g_tn_zero16;   // no second B half tile: same instruction count, no traffic
        __builtin_amdgcn_global_load_lds((const void*)p0, (lds_void*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)p1, (lds_void*)(dst + 1024), 16, 0, 0);
    };

    // ---- transposed fragment reads.  16x16x32 operand: lane (column c = l & 15, k group gk = l >> 4) holds
    // k = 8 gk .. 8 gk + 7 of its column: two ds_read_b64_tr_b16, each a 4 (k) x 16 (columns) block per 16-lane
    // group; lane 4q+p of the group supplies the address of block row q, columns 4p..4p+3.
    //   image: off(row, ch) = 2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3))
    // with row = 32 ks + 8 gk + q (+ 4 for the second read), ch = 2 jb + (p >> 1) for 16-column block jb.
    // The XOR makes the slot depend on the parity of jb and on which read: four lane bases.
    const int q = (lane >> 2) & 3, p = lane & 3, gk = g4;
    int lb[2][2];                                        // [parity of the column block][first / second read]
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 8 * gk + q + 4 * rd;
            const int chl = 2 * par + (p >> 1);
            lb[par][rd] = 2048 * (row >> 3) + 64 * (row & 7) + 16 * (chl ^ ((row >> 2) & 3)) + 8 * (p & 1);
        }
    const unsigned abase = lds_addr(smem) + 1024 * wr;          // A side: column blocks 4 wr + i -> 512 * ((4 wr + i) >> 1)
    const unsigned bbase = lds_addr(smem) + LDS_B + 512 * wc;   // B side: column blocks 2 wc + j -> 512 * wc
    unsigned la[2][2], lbb[2][2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            la[par][rd] = abase + lb[par][rd];
            lbb[par][rd] = bbase + LDS_B * 0 + lb[par][rd];
        }
    bf16x8 a[4][2], b0[2][2], b1[2][2];
    // the B region starts at LDS_B = 65536: beyond the 16-bit offset field, so it lives in the base registers
    auto read_a = [&](auto xt, auto dt) {
        constexpr int slot = (2 * decltype(xt)::value + decltype(dt)::value) * HT;
        static_for<2>([&](auto kst) {
            static_for<4>([&](auto it) {
                constexpr int ks = decltype(kst)::value, i = decltype(it)::value;
                a[i][ks] = lds_tr_pair<slot + 8192 * ks + 512 * (i >> 1)>(la[i & 1][0], la[i & 1][1]);
            });
        });
    };
    auto read_b = [&](bf16x8 (&b)[2][2], auto xt, auto dt) {
        constexpr int slot = (2 * decltype(xt)::value + decltype(dt)::value) * HT;
        static_for<2>([&](auto kst) {
            static_for<2>([&](auto jt) {
                constexpr int ks = decltype(kst)::value, j = decltype(jt)::value;
                b[j][ks] = lds_tr_pair<slot + 8192 * ks>(lbb[j][0], lbb[j][1]);
            });
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[qa][qb][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define TN_MMA(QA, QB, BF)                                                                                         \
    do {                                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
                    acc[QA][QB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], a[i][ks], acc[QA][QB][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                             \
    } while (0)

    // prologue / main loop: identical to gemm_nt.hip (see the schedule rules there)
    stage_a(0, 0, 0);
    stage_b(0, 0, 0);
    stage_b(1, 0, 0);
    stage_a(1, 0, 0);
    stage_a(0, 1, 1);
    stage_b(0, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    TN_BAR();
    if (wr == 1) TN_BAR();

    auto tile_step = [&](auto dtag, int t) {
        constexpr int d = decltype(dtag)::value;
        read_a(I0{}, dtag);
        read_b(b0, I0{}, dtag);
        stage_b(1, d ^ 1, t + 1);
        TN_BAR();
        TN_LGKM0();
        TN_MMA(0, 0, b0);
        TN_BAR();
        if constexpr (MODE == 0) read_b(b1, I1{}, dtag);
        stage_a(1, d ^ 1, t + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        TN_BAR();
        TN_LGKM0();
        if constexpr (MODE == 0) TN_MMA(0, 1, b1);
        TN_BAR();
        read_a(I1{}, dtag);
        stage_a(0, d, t + 2);
        TN_BAR();
        TN_LGKM0();
        if constexpr (MODE == 0) TN_MMA(1, 1, b1);
        TN_BAR();
        stage_b(0, d, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        TN_BAR();
        TN_MMA(1, 0, b0);
        TN_BAR();
    };
    for (int t = 0; t < nk; t += 2) {
        tile_step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) tile_step(std::integral_constant<int, 1>{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) TN_BAR();
#undef TN_MMA

    // fp32 slab store: D[row = n][col = m]: lane m = l15 holds 4 consecutive n per accumulator (16-byte stores,
    // 64 contiguous bytes per row and instruction)
    float* C = P.C + (long)split * P.M * P.N;
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 128 * qa + 64 * wr + 16 * i + l15;
            if (m >= P.M) continue;
#pragma unroll
            for (int qb = 0; qb < (MODE == 0 ? 2 : 1); ++qb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + 128 * qb + 32 * wc + 16 * j + 4 * g4;
                    if (n + 4 <= P.N) *reinterpret_cast<f32x4*>(C + (long)m * P.N + n) = acc[qa][qb][i][j];
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

static long tn_tiles(int M, int N) { return (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN); }
// one workgroup per CU (128 KiB of LDS each) and equal work per workgroup: a single round of at most one workgroup
// per CU of the current device (256 on MI355X) has no tail and writes the fewest fp32 slabs
static int tn_splits(long tiles, int K) {
    const int cus = mgx_cu_count() > 0 ? mgx_cu_count() : 256;
    int splits = (int)(cus / tiles);
    const int max_splits = (K + 4 * BK - 1) / (4 * BK);     // at least 4 K tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int len = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    return (K + len - 1) / len;
}
static int tn_check(const char* fn, const void* A, const void* B, const void* C, int M, int N, int K, int lda, int ldb) {
    MGX_REQUIRE(A && B && C, MGX_EINVAL, "%s: null pointer", fn);
    MGX_REQUIRE(M >= 8 && N >= 8 && K > 0, MGX_EINVAL, "%s: M=%d N=%d K=%d", fn, M, N, K);
    MGX_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N, MGX_EUNSUPPORTED,
                "%s: M, N and leading dimensions must be multiples of 8: M=%d N=%d lda=%d ldb=%d", fn, M, N, lda, ldb);
    MGX_REQUIRE(64L * lda < (1L << 31) && 64L * ldb < (1L << 31), MGX_EUNSUPPORTED, "%s: leading dimension too large", fn);
    return MGX_OK;
}
static int tn_reduce(const float* slabs, void* C, long n, int splits, int out_dtype, hipStream_t st) {
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (out_dtype == MGX_F32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, slabs, (float*)C, n, splits);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, slabs, (bf16_t*)C, n, splits);
    MGX_CHECK_LAUNCH("splitk_reduce");
    return MGX_OK;
}

extern "C" long mgx_gemm_tn_workspace_floats(int M, int N, int K, int* splits_out) {
    const int splits = tn_splits(tn_tiles(M, N), K);
    if (splits_out) *splits_out = splits;
    return (long)splits * M * N;
}

extern "C" int mgx_gemm_tn_bf16(const void* A, const void* B, void* C, float* workspace, int M, int N, int K, int lda,
                                int ldb, int out_dtype, void* stream) {
    if (int rc = tn_check("mgx_gemm_tn_bf16", A, B, C, M, N, K, lda, ldb)) return rc;
    MGX_REQUIRE(workspace, MGX_EINVAL, "mgx_gemm_tn_bf16: null workspace");
    MGX_REQUIRE(out_dtype == MGX_F32 || out_dtype == MGX_BF16, MGX_EINVAL, "mgx_gemm_tn_bf16: bad out dtype");
    const long tiles = tn_tiles(M, N);
    const int splits = tn_splits(tiles, K);
    const int len = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    TnArgs g{{{(const char*)A, (const char*)B, workspace, M, N, lda, ldb}, {nullptr, nullptr, nullptr, 8, 8, 8, 8}},
             (int)tiles, 0, K, len, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    if (K % BK == 0)
        hipLaunchKernelGGL((gemm_tn_kernel<false, 0>), dim3((unsigned)(tiles * splits)), dim3(NTHREADS), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<true, 0>), dim3((unsigned)(tiles * splits)), dim3(NTHREADS), 0, st, g);
    MGX_CHECK_LAUNCH("gemm_tn");
    return tn_reduce(workspace, C, (long)M * N, splits, out_dtype, st);
}

// Two weight gradients over the same K (token) dimension in one launch: together they have enough tiles for fewer K
// splits (linear1 + linear2 of the block: 128 tiles x 2 splits instead of 64 x 4 each), i.e. half the slab traffic.
extern "C" long mgx_gemm_tn_pair_workspace_floats(int M0, int N0, int M1, int N1, int K) {
    return (long)tn_splits(tn_tiles(M0, N0) + tn_tiles(M1, N1), K) * ((long)M0 * N0 + (long)M1 * N1);
}

extern "C" int mgx_gemm_tn_pair_bf16(const void* A0, const void* B0, void* C0, int M0, int N0, int lda0, int ldb0,
                                     const void* A1, const void* B1, void* C1, int M1, int N1, int lda1, int ldb1, int K,
                                     float* workspace, int out_dtype, void* stream) {
    if (int rc = tn_check("mgx_gemm_tn_pair_bf16", A0, B0, C0, M0, N0, K, lda0, ldb0)) return rc;
    if (int rc = tn_check("mgx_gemm_tn_pair_bf16", A1, B1, C1, M1, N1, K, lda1, ldb1)) return rc;
    MGX_REQUIRE(workspace, MGX_EINVAL, "mgx_gemm_tn_pair_bf16: null workspace");
    MGX_REQUIRE(out_dtype == MGX_F32 || out_dtype == MGX_BF16, MGX_EINVAL, "mgx_gemm_tn_pair_bf16: bad out dtype");
    const long t0 = tn_tiles(M0, N0), t1 = tn_tiles(M1, N1);
    const int splits = tn_splits(t0 + t1, K);
    const int len = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    float* ws1 = workspace + (long)splits * M0 * N0;
    TnArgs g{{{(const char*)A0, (const char*)B0, workspace, M0, N0, lda0, ldb0},
              {(const char*)A1, (const char*)B1, ws1, M1, N1, lda1, ldb1}},
             (int)t0, (int)t1, K, len, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((t0 + t1) * splits);
    if (K % BK == 0)
        hipLaunchKernelGGL((gemm_tn_kernel<false, 0>), dim3(grid), dim3(NTHREADS), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<true, 0>), dim3(grid), dim3(NTHREADS), 0, st, g);
    MGX_CHECK_LAUNCH("gemm_tn_pair");
    if (int rc = tn_reduce(workspace, C0, (long)M0 * N0, splits, out_dtype, st)) return rc;
    return tn_reduce(ws1, C1, (long)M1 * N1, splits, out_dtype, st);
}

// skew relative attention backward, dEr slabs from the dG blocks of rel_bwd_dq_mfma_kernel (attn_rel_mfma.hip).
// dg: [B*H][staircase of 2-KiB blocks: unit qu holds distance blocks 0 .. qu | 7, mgx_rel_dg_row]; slabs: [B*H][32 nrb][dh] fp32
int mgx_rel_de_from_dg(const void* dg, const void* qkv, float* slabs, int B, int S, int H, int dh, int nrb, hipStream_t st) {
    MGX_REQUIRE(dh == 64 || dh == 128, MGX_EUNSUPPORTED, "rel dE from dG: dh=%d", dh);
    const int Mp = nrb * 32;
    const int rs = 3 * H * dh;
    MGX_REQUIRE(64L * rs < (1L << 30), MGX_EUNSUPPORTED, "rel dE from dG: row stride too large");
    TnArgs g{{{(const char*)dg, (const char*)qkv, slabs, Mp, dh, 0, rs}, {nullptr, nullptr, nullptr, 8, 8, 8, 8}},
             0, 0, S, 0, B * H, H, mgx_rel_dg_row(nrb)};
    const int nT = (min(Mp, (S + 31) / 32 * 32) + BM - 1) / BM;       // distance tiles that hold a distance <= S - 1
    const unsigned grid = (unsigned)(B * H * nT);
    if (S % BK == 0)
        hipLaunchKernelGGL((gemm_tn_kernel<false, 1>), dim3(grid), dim3(NTHREADS), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<true, 1>), dim3(grid), dim3(NTHREADS), 0, st, g);
    MGX_CHECK_LAUNCH("rel_de_from_dg");
    return MGX_OK;
}
