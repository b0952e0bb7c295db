// bf16 MFMA flash-attention backward for gfx950, software-pipelined inside the wave (causal + ALiBi).
// Autograd of SimpleAttention.multihead_attn (model_baseline/alibi_unified.py:40-54).
//
// dK/dV kernel.  One wave per SIMD (4 waves, 32 keys each; K and V fragments and the 32 x dh dK / dV accumulators
// stay in registers).  A workgroup is PERSISTENT: it walks a list of (batch, head, 128-key block) items, heavy and
// light key blocks paired so that every workgroup gets the same number of query rows, and the LDS-DMA stream of
// query tiles (64 rows of Q and dO + their row statistics, ring of 4 slots) runs on across item boundaries.
// Queries are consumed in UNITS of 32 rows.  Per unit u and wave:
//     A(u)  S = Q.K^T + nl, dP = dO.V^T + nd       2*(dh/16 + 1) MFMAs; the row statistics enter as one extra k-step
//                                                   (A = the statistic split into three bf16, B = ones), so no
//                                                   per-row value has to be fetched into registers for the VALU
//     B(u)  P = exp2(c S + bias), dS = P * dP       plain fp32 VALU (v_pk_* does not run beside the MFMA pipe)
//     C(u)  dV^T += dO^T.P, dK^T += Q^T.dS          4*dh/32 MFMAs   (P, dS straight from the accumulators)
// and the loop body issues  X(u) = A(u+1) with the exponentials of B(u) between the MFMAs, then  Y(u) = C(u) with
// the dS half of B(u) between them; every gap between two MFMAs also carries ONE KiB of LDS fragment reads for a
// later stage (more than that per gap does not hide under an MFMA: tools/mfma_overlap_lab2.cpp).
// All LDS reads are inline asm with explicit s_waitcnt (hipcc would otherwise drain vmcnt before every read that
// follows an LDS-DMA).
#include "attn_pipe_common.h"

namespace {

// Row statistics for the backward, rows padded to Sp (a multiple of 64):
//   nla[b,h,i] (16 B) = -(slope_h * (i mod 32) + lse) * log2e / c  split into three bf16 (hi, mid, lo) + five zeros:
//                       the initial value of S in raw-score units, relative to the start of the 32-row unit;
//                       -1e30 for S <= i < Sp (P becomes exactly 0)
//   nda[b,h,i] (16 B) = -sum_d dout*out, split the same way; 0 for the padding rows
//   nd32[b,h,i]       = the same value in fp32 (dQ kernel)
__device__ __forceinline__ void split3(float x, bf16_t* dst) {
    const bf16_t hi = (bf16_t)x;
    const float r1 = x - (float)hi;
    const bf16_t mid = (bf16_t)r1;
    const bf16_t lo = (bf16_t)(r1 - (float)mid);
    bf16x8 v = {hi, mid, lo, 0, 0, 0, 0, 0};
    *reinterpret_cast<bf16x8*>(dst) = v;
}
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const bf16_t* __restrict__ dout,
                                                              const bf16_t* __restrict__ out,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ slopes, bf16_t* __restrict__ nla,
                                                              bf16_t* __restrict__ nda, float* __restrict__ nd32,
                                                              long n_chunks, int S, int Sp, int H, float inv_c) {
    constexpr int LPH = DH / 8;
    const long ch = (long)blockIdx.x * 256 + threadIdx.x;      // chunk of the padded [B, Sp, H, dh/8] index space
    const long bh = ch / LPH;                                  // (b*Sp + i)*H + h
    const int h = (int)(bh % H);
    const long bi = bh / H;
    const int i = (int)(bi % Sp);
    const long b = bi / Sp;
    const bool live = ch < n_chunks && i < S;
    float s = 0.f;
    if (live) {
        const long src = (((b * S + i) * H + h) * LPH + ch % LPH) * 8;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dout + src);
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(out + src);
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf((float)a[k], (float)o[k], s);
    }
#pragma unroll
    for (int o = 1; o < LPH; o <<= 1) s += __shfl_xor(s, o, 64);
    if (ch < n_chunks && (ch % LPH) == 0) {
        const long dst = (b * H + h) * Sp + i;
        const float xl = i < S ? -(slopes[h] * (float)(i & 31) + lse[(b * H + h) * S + i]) * LOG2E * inv_c : -1e30f;
        const float xd = i < S ? -s : 0.f;
        split3(xl, nla + dst * 8);
        split3(xd, nda + dst * 8);
        nd32[dst] = xd;
    }
}

template <int DH>
struct DkvCfg {
    static constexpr int NB = 4;                       // LDS ring slots (one 64-query tile each)
    static constexpr int UB = 32 * DH * 2;             // bytes of one 32 x dh bf16 image (a unit)
    static constexpr int TB = 2 * UB;                  // a 64-row tile of Q or dO
    static constexpr int AUG = 2 * TB;                 // offset of the statistics: nla[64] then nda[64], 16 B per row
    static constexpr int SLOT = 2 * TB + 2048;
    static constexpr int VM_PER_TILE = 4 * (DH / 64) + 1;   // LDS-DMA instructions per wave and tile
    static constexpr int KS = DH / 16, DB = DH / 32;
    static constexpr int NG_X = 2 * KS + 2;            // MFMAs of stage A (S and dP chains, statistics step last)
    static constexpr int NG_Y = 4 * DB;                // MFMAs of stage C: NG_Y / 2 for dV, then NG_Y / 2 for dK
    static constexpr int HF = NG_Y / 2;
    static constexpr int EPG = DH == 128 ? 1 : 2;      // elements of P per gap of X
    // AGPR map
    static constexpr int R_DK = 0, R_DV = 16 * DB, R_KF = 32 * DB, R_VF = R_KF + 4 * KS, R_TV = R_VF + 4 * KS;
    static constexpr int R_TK = R_TV + 4 * HF, R_END = R_TK + 4 * HF;   // 256 at dh 128, 128 at dh 64
};

struct DkvArgs {
    const bf16_t* dout;
    const bf16_t* qkv;
    const bf16_t* nla;
    const bf16_t* nda;
    const float* slopes;
    bf16_t* dqkv;
    int B, S, Sp, H;
    float c, scale;
    int nkb, np;        // 128-key blocks per (b,h); pairs of them
    int nwork;          // B*H*np pairs in all
};

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_pipe_kernel(DkvArgs g) {
    using Cfg = DkvCfg<DH>;
    constexpr int NB = Cfg::NB, UB = Cfg::UB, TB = Cfg::TB, SLOT = Cfg::SLOT, G8 = Img<DH>::G8;
    constexpr int KS = Cfg::KS, DB = Cfg::DB, HF = Cfg::HF, NGX = Cfg::NG_X, VM = Cfg::VM_PER_TILE;
    static_assert(Cfg::R_END <= 256, "AGPR map");
    static_assert(2 * VM <= NGX, "one LDS-DMA piece per odd gap of X");
    constexpr int EPI = NB * SLOT;                       // 4 KiB per wave: transposition buffer of the item epilogue
    __shared__ __attribute__((aligned(1024))) char smem[NB * SLOT + 4 * 4096];
    // the AGPRs named in the asm below, made known to the compiler's resource accounting (it allocates none itself)
    asm volatile("" ::: "a0", "a127", "a255");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h5 = lane >> 5;
    const int S = g.S, Sp = g.Sp, H = g.H;
    const unsigned rs = 3u * H * DH;
    const unsigned rs2 = rs * 2u, gs2 = (unsigned)(H * DH * 2);
    const LaneAddr<DH> la(lane);
    const int tend = Sp / 64;                            // tiles of an item: [2 kb, tend)
    const int tlast_real = (S - 1) / 64;                 // last tile that holds real rows

    // ---- this workgroup's items.  Work pair q = (b*H + h, p) -> key blocks p (heavy) and nkb-1-p (light); the
    // workgroups of one XCD (blockIdx % 8) take consecutive pairs, i.e. the key blocks of the same few (b, h), so
    // that one Q / dO stream serves them from that XCD's L2.
    const int G = gridDim.x;
    const int gq = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    const int nseq = 2 * ((g.nwork + G - 1) / G);        // sequence numbers [0, nseq)
    auto item_at = [&](int n, int& bh, int& kb) -> bool {
        const int q = (n >> 1) * G + gq;
        if (n >= nseq || q >= g.nwork) return false;
        bh = q / g.np;
        const int p = q - bh * g.np;
        kb = (n & 1) ? g.nkb - 1 - p : p;
        return !(n & 1) || kb > p;
    };
    auto next_item = [&](int n, int& bh, int& kb) -> int {   // first valid sequence number > n, or nseq
        for (++n; n < nseq; ++n)
            if (item_at(n, bh, kb)) return n;
        return nseq;
    };

    // ---- producer: LDS-DMA of one 64-row tile.  Wave w brings rows 16w..16w+15 of Q and of dO (two 8-row groups of
    // DH/64 KiB-pieces each) and one KiB of statistics (even waves nla, odd waves nda; written twice, same bytes).
    // Lane l of a piece lands at byte 16 l: 512-B sub-tile l >> 5, row (l >> 2) & 7, slot l & 3 of the image
    //   off(row, ch) = G8 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)).
    const int rowin = (lane >> 2) & 7;
    unsigned offq[2][DH / 64], offg[2][DH / 64];         // per-lane byte offsets from the tile's first row
    int prow[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        prow[gi] = 16 * w + 8 * gi + rowin;
        const int sx = (lane & 3) ^ ((prow[gi] >> 2) & 3);
#pragma unroll
        for (int hf = 0; hf < DH / 64; ++hf) {
            const unsigned colb = (unsigned)((4 * (2 * hf + (lane >> 5)) + sx) * 16);
            offq[gi][hf] = (unsigned)prow[gi] * rs2 + colb;
            offg[gi][hf] = (unsigned)prow[gi] * gs2 + colb;
        }
    }
    int p_n, p_bh = 0, p_kb = 0, p_tile = 0;             // producer cursor: item sequence number, tile inside it
    int p_slot = 0;
    const char* p_q = nullptr;                           // Q rows, dO rows, statistics of the producer's (b, h)
    const char* p_g = nullptr;
    const char* p_a = nullptr;
    auto producer_set_item = [&]() {
        const int b = p_bh / H, h = p_bh - b * H;
        p_q = (const char*)(g.qkv + (long)b * S * rs + (long)h * DH);
        p_g = (const char*)(g.dout + (long)b * S * H * DH + (long)h * DH);
        p_a = (const char*)(((w & 1) ? g.nda : g.nla) + ((long)b * H + h) * Sp * 8) + 16 * lane;
        p_tile = 2 * p_kb;
    };
    // One tile = VM pieces per wave.  produce_begin fixes the source / destination of the tile at the cursor,
    // produce_piece<k> issues one LDS-DMA instruction, produce_end advances the cursor.  The pieces are issued one
    // per MFMA gap: nine of them back to back stall the wave for ~600 cycles (the four waves' 36 KiB-instructions
    // queue up behind one texture-address unit).
    const char* p_q0 = nullptr;
    const char* p_g0 = nullptr;
    int p_back[2] = {0, 0};
    auto produce_begin = [&]() {
        const int tq = min(p_tile, tlast_real);          // padding tiles re-read real rows (their P is exactly 0)
        p_q0 = p_q + (long)tq * 64 * rs2;
        p_g0 = p_g + (long)tq * 64 * gs2;
        // rows past the end re-read row S-1
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) p_back[gi] = max(tq * 64 + prow[gi] - (S - 1), 0);
    };
    auto produce_piece = [&](auto kt) {
        constexpr int k = decltype(kt)::value;
        if constexpr (k == VM - 1) {
            __builtin_amdgcn_global_load_lds((const void*)(p_a + (long)p_tile * 1024),
                                             (lds_void_t*)(smem + p_slot * SLOT + Cfg::AUG + 1024 * (w & 1)), 16, 0, 0);
        } else {
            constexpr int which = k & 1, hf = (k >> 1) % (DH / 64), gi = (k >> 1) / (DH / 64);
            char* dst = smem + p_slot * SLOT + 2 * G8 * w + G8 * gi + 1024 * hf;
            if constexpr (which)
                __builtin_amdgcn_global_load_lds((const void*)(p_g0 + (offg[gi][hf] - (unsigned)p_back[gi] * gs2)),
                                                 (lds_void_t*)(dst + TB), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const void*)(p_q0 + (offq[gi][hf] - (unsigned)p_back[gi] * rs2)),
                                                 (lds_void_t*)dst, 16, 0, 0);
        }
    };
    auto produce_end = [&]() {
        p_slot = (p_slot + 1) & (NB - 1);
        // advance; past the last item the last tile is fetched again (harmless, keeps the counts uniform)
        if (p_tile + 1 < tend) {
            ++p_tile;
        } else if (p_n < nseq) {
            int bh, kb;
            const int n = next_item(p_n, bh, kb);
            p_n = n;
            if (n < nseq) {
                p_bh = bh;
                p_kb = kb;
                producer_set_item();
            }
        }
    };
    auto produce = [&]() {
        produce_begin();
        sfor<VM>([&](auto kt) { produce_piece(kt); });
        produce_end();
    };

    // ---- consumer state
    const unsigned lds0 = lds_addr32(smem);
    bf16x8 rq[KS + 1], rg[KS + 1];   // row fragments (A operands of S / dP; index KS: the statistics step)
    bf16x8 ones;                     // B operand of the statistics step: 1 at k = 0, 1, 2 (lanes of the first half)
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (bf16_t)((h5 == 0 && k < 3) ? 1.0f : 0.0f);

    // row fragment fg of the unit at byte `ub` (its slot + 0 / UB): fg even: Q rows (S chain), odd: dO rows (dP
    // chain); k-step fg >> 1, the last one being the statistics row (then ub = slot + AUG + 512 * unit-in-tile)
    auto read_row = [&](auto gt, unsigned ub) {
        constexpr int fg = decltype(gt)::value, ks = fg >> 1;
        if constexpr (ks == KS) {
            const unsigned a = ub + 16 * l31;
            if constexpr (fg & 1)
                rg[KS] = lds_b128<1024>(a);
            else
                rq[KS] = lds_b128<0>(a);
        } else {
            const unsigned a = ub + ((ks & 1) ? la.row1 : la.row0);
            if constexpr (fg & 1)
                rg[ks] = lds_b128<TB + 512 * (ks >> 1)>(a);
            else
                rq[ks] = lds_b128<512 * (ks >> 1)>(a);
        }
    };
    // transposed fragment f of stage C of the unit at byte `ub`: f < HF: dO^T for dV -> a[R_TV + 4 f ..],
    // f >= HF: Q^T for dK -> a[R_TK + 4 (f - HF) ..]; (s2, d) = ((f % HF) / DB, f % DB)
    auto read_tr = [&](auto ft, unsigned ub) {
        constexpr int f = decltype(ft)::value;
        constexpr int which = f / HF, d = f % DB, s2 = (f % HF) / DB;
        constexpr int o0 = (which ? 0 : TB) + G8 * (2 * s2) + 512 * d, o1 = o0 + G8;
        constexpr int reg = which ? Cfg::R_TK + 4 * (f - HF) : Cfg::R_TV + 4 * f;
        lds_tr128_agpr<reg, o0, o1>(ub + la.tr0, ub + la.tr1);
    };
    // stage A: S and dP chains, alternating (gap index fg: even = S, odd = dP)
    auto mfma_a_step = [&](auto gt, f32x16& sc, f32x16& dp) {
        constexpr int fg = decltype(gt)::value, ks = fg >> 1;
        if constexpr (fg == 0)
            mfma_s0<Cfg::R_KF>(sc, rq[0]);
        else if constexpr (fg == 1)
            mfma_s0<Cfg::R_VF>(dp, rg[0]);
        else if constexpr (ks == KS) {
            if constexpr (fg & 1)
                mfma_sv(dp, rg[KS], ones);
            else
                mfma_sv(sc, rq[KS], ones);
        } else if constexpr (fg & 1)
            mfma_s<Cfg::R_VF + 4 * ks>(dp, rg[ks]);
        else
            mfma_s<Cfg::R_KF + 4 * ks>(sc, rq[ks]);
    };
    auto load_kv = [&](int bh, int kb) {
        // K / V fragments of this lane's key straight into AGPRs; keys past the end read row S-1 (finite data, and
        // their columns of dK / dV are never stored)
        const int b = bh / H, h = bh - b * H;
        const bf16_t* kp = g.qkv + (long)b * S * rs + (long)h * DH + (long)H * DH;
        const bf16_t* vp = kp + (long)H * DH;
        const long jr = (long)min(kb * 128 + 32 * w + l31, S - 1) * rs + 8 * h5;
        sfor<KS>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            global_b128_to_agpr<Cfg::R_KF + 4 * ks>(kp + jr + ks * 16);
            global_b128_to_agpr<Cfg::R_VF + 4 * ks>(vp + jr + ks * 16);
        });
    };

    // ---- start: first item, its K / V, three tiles in flight
    int c_n, c_bh = 0, c_kb = 0;
    c_n = next_item(-1, c_bh, c_kb);
    if (c_n >= nseq) return;                             // (uniform) nothing to do for this workgroup
    p_n = c_n;
    p_bh = c_bh;
    p_kb = c_kb;
    producer_set_item();
    load_kv(c_bh, c_kb);
    produce();
    produce();
    produce();
    sfor<32 * DB>([&](auto rt) { agpr_zero<decltype(rt)::value>(); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VM) : "memory");      // K, V and the first tile
    PIPE_FENCE();
    __builtin_amdgcn_s_barrier();
    PIPE_FENCE();

    int slot_c = 0;                                      // slot of the tile being consumed
    f32x16 scA, dpA, scB, dpB;
    while (true) {
        // ================= one item: key block c_kb of (b, h) = c_bh
        const int b = c_bh / H, h = c_bh - b * H;
        const int j = c_kb * 128 + 32 * w + l31;
        const float csb = g.slopes[h] * LOG2E;           // bias per key step in log2 units
        const int u0 = 4 * c_kb;
        const int umask = u0 + 4;                        // units [u0, umask) can touch the causal diagonal
        float cjc = csb * (float)(j - 32 * u0);          // bias of key j against the first row of the unit
        int n_bh = 0, n_kb = 0;
        const int n_n = next_item(c_n, n_bh, n_kb);
        {
            // not overlapped: A(u0), then the row fragments of unit u0+1 and the dO^T fragments of unit u0
            const unsigned ub = lds0 + slot_c * SLOT;
            sfor<NGX>([&](auto gt) {
                constexpr int fg = decltype(gt)::value;
                read_row(gt, (fg >> 1) == KS ? ub + Cfg::AUG : ub);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIPE_FENCE();
            sfor<NGX>([&](auto gt) { mfma_a_step(gt, scA, dpA); });
            PIPE_FENCE();
            sfor<NGX>([&](auto gt) {
                constexpr int fg = decltype(gt)::value;
                read_row(gt, (fg >> 1) == KS ? ub + Cfg::AUG + 512 : ub + UB);
            });
            sfor<HF>([&](auto ft) { read_tr(ft, ub); });
            PIPE_FENCE();
        }

        // One unit (tile t, half HB).  (sc, dp) hold S / dP of unit u (from the previous X), (scn, dpn) receive u + 1.
        auto unit = [&](int t, auto hb_tag, f32x16& sc, f32x16& dp, f32x16& scn, f32x16& dpn) {
            constexpr int HB = decltype(hb_tag)::value;
            const int u = 2 * t + HB;
            const int slot_n = (slot_c + 1) & (NB - 1);
            const unsigned ucur = lds0 + slot_c * SLOT + HB * UB;            // this unit's images
            const unsigned unext = lds0 + slot_n * SLOT + HB * UB;           // unit u + 2
            const unsigned u1 = HB ? lds0 + slot_n * SLOT : ucur + UB;       // unit u + 1
            const unsigned anext = lds0 + slot_n * SLOT + Cfg::AUG + 512 * HB;
            // rows of u+1 (read during X(u-1)) must have landed; the newest reads (dO^T of unit u) may fly
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * HF < 15 ? 2 * HF : 15) : "memory");
            PIPE_FENCE();
            if (HB && t + 1 == tend && n_n < nseq) load_kv(n_bh, n_kb);      // K / V are dead: A(u+1) is never used
            if (u < umask) {                              // wave-uniform, at most 4 units per item
                const int di = j - 32 * u - 4 * h5;       // row part < di  <=>  query before key
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA of A(u) must have landed
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((e & 3) + 8 * (e >> 2) < di) sc[e] = -INFINITY;
            }
            // ---------------- X(u)
            PIPE_FENCE();
            float pe[16];
            unsigned pf[8], dsf[8];                       // bf16 pairs: P and dS of unit u
            sfor<NGX>([&](auto gt) {
                constexpr int fg = decltype(gt)::value;
                mfma_a_step(gt, scn, dpn);
                PIPE_FENCE();
                if constexpr (fg == 0 && HB == 0) {
                    // tile top, in the shadow of the first MFMA: the next tile must be complete before its rows are
                    // read below, and the slot of the previous tile is free once every wave is here
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
                    PIPE_FENCE();
                    __builtin_amdgcn_s_barrier();
                    PIPE_FENCE();
                    produce_begin();
                    PIPE_FENCE();
                }
                if constexpr (HB == 0 && (fg & 1) && (fg >> 1) < VM) {
                    produce_piece(std::integral_constant<int, (fg >> 1)>{});
                    if constexpr ((fg >> 1) == VM - 1) produce_end();
                }
                read_row(gt, (fg >> 1) == KS ? anext : unext);
                if constexpr (fg < 16 / Cfg::EPG) {
#pragma unroll
                    for (int q = 0; q < Cfg::EPG; ++q) {
                        constexpr int e0 = fg * Cfg::EPG;
                        const int e = e0 + q;
                        pe[e] = fast_exp2(fma_np(sc[e], g.c, cjc));
                        if (e & 1) pf[e >> 1] = pack2(pe[e - 1], pe[e]);
                    }
                }
                PIPE_FENCE();
            });
            // ---------------- Y(u): dV MFMAs (with dS of unit u and the Q^T fragments of unit u between them), then dK
            // MFMAs (with the dO^T fragments of unit u+1 between them)
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NGX < 15 ? NGX : 15) : "memory");   // dO^T fragments of unit u
            PIPE_FENCE();
            bf16x8 pb[2] = {words_to_frag(pf[0], pf[1], pf[2], pf[3]), words_to_frag(pf[4], pf[5], pf[6], pf[7])};
            pin_frags(pb[0], pb[1]);                      // VALU write of P -> MFMA read
            bf16x8 db[2];
            PIPE_FENCE();
            sfor<Cfg::NG_Y>([&](auto gt) {
                constexpr int f = decltype(gt)::value;
                constexpr int d = f % DB, s2 = (f % HF) / DB;
                if constexpr (f >= HF) {
                    if constexpr (f == HF) {
                        db[0] = words_to_frag(dsf[0], dsf[1], dsf[2], dsf[3]);
                        db[1] = words_to_frag(dsf[4], dsf[5], dsf[6], dsf[7]);
                        pin_frags(db[0], db[1]);                  // VALU write of dS -> MFMA read
                    }
                    // Q^T fragment f - HF was read in gap f - HF; 2 (HF - 1) reads have been issued since
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (HF - 1)) : "memory");
                    PIPE_FENCE();
                    mfma_acc<Cfg::R_DK + 16 * d, Cfg::R_TK + 4 * (f - HF)>(db[s2]);
                } else {
                    mfma_acc<Cfg::R_DV + 16 * d, Cfg::R_TV + 4 * f>(pb[s2]);
                }
                PIPE_FENCE();
                if constexpr (f < HF) {
#pragma unroll
                    for (int q = 0; q < 8 / HF; ++q) {
                        constexpr int pr0 = f * (8 / HF);
                        const int pr = pr0 + q, e = 2 * pr;
                        dsf[pr] = pack2(mul_np(pe[e], dp[e]), mul_np(pe[e + 1], dp[e + 1]));
                    }
                    read_tr(std::integral_constant<int, f + HF>{}, ucur);
                } else {
                    read_tr(std::integral_constant<int, f - HF>{}, u1);
                }
                PIPE_FENCE();
            });
            cjc -= 32.f * csb;
        };

        for (int t = 2 * c_kb; t < tend; ++t) {
            unit(t, std::integral_constant<int, 0>{}, scA, dpA, scB, dpB);
            unit(t, std::integral_constant<int, 1>{}, scB, dpB, scA, dpA);
            slot_c = (slot_c + 1) & (NB - 1);
        }

        // ---- item done.  The last MFMAs must have written dK / dV before they are read back (16 passes; nothing
        // tells the compiler about the dependency).
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // The next item's resident rows (requested a unit ago, younger than every tile request in flight) are waited
        // for HERE, before the epilogue's stores are issued: after them a vmcnt(0) would also wait for the stores'
        // acknowledgements.  The counted tile waits of the next item stay valid with stores outstanding: loads
        // retire in order among themselves, stores only make the count conservative.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_FENCE();
        {
            // dK^T / dV^T have the key on the lane and dh down the registers: written out directly, every store
            // instruction would scatter 8-byte pieces over 64 rows.  Each wave turns its 32 keys x 64 columns at a time
            // through 4 KiB of LDS (16-byte chunks XOR-swizzled with the row) and stores 128-byte row segments.
            const unsigned eb = lds0 + EPI + 4096 * w;
            const int erow = lane >> 3, ech = lane & 7;
            bf16_t* const orow = g.dqkv + ((long)b * S + c_kb * 128 + 32 * w) * rs + (long)h * DH;
            sfor<2 * (DH / 64)>([&](auto pt) {
                constexpr int mat = decltype(pt)::value / (DH / 64), half = decltype(pt)::value % (DH / 64);   // mat 0: dK, 1: dV
                sfor<2>([&](auto dt) {
                    constexpr int db = 2 * half + decltype(dt)::value;
                    sfor<4>([&](auto gt) {
                        constexpr int gg = decltype(gt)::value;
                        constexpr int r0 = (mat ? Cfg::R_DV : Cfg::R_DK) + 16 * db + 4 * gg;
                        const float sc_ = mat ? 1.f : g.scale;
                        bf16x4 v;
                        v[0] = (bf16_t)(agpr_get<r0 + 0>() * sc_);
                        v[1] = (bf16_t)(agpr_get<r0 + 1>() * sc_);
                        v[2] = (bf16_t)(agpr_get<r0 + 2>() * sc_);
                        v[3] = (bf16_t)(agpr_get<r0 + 3>() * sc_);
                        const int c16 = 4 * (db & 1) + gg;
                        lds_write_b64(eb + l31 * 128 + ((c16 ^ (l31 & 7)) * 16) + 8 * h5, v);
                    });
                });
                // the wave reads back what it wrote itself: LDS operations of one wave complete in order
                bf16x8 ov[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + erow;
                    ov[i] = lds_b128<0>(eb + row * 128 + ((ech ^ (row & 7)) * 16));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PIPE_FENCE();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + erow;
                    if (c_kb * 128 + 32 * w + row < S)
                        *reinterpret_cast<bf16x8*>(orow + (long)row * rs + (long)(1 + mat) * H * DH + 64 * half + 8 * ech) = ov[i];
                }
            });
        }
        PIPE_FENCE();
        if (n_n >= nseq) break;
        c_n = n_n;
        c_bh = n_bh;
        c_kb = n_kb;
        sfor<32 * DB>([&](auto rt) { agpr_zero<decltype(rt)::value>(); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIPE_FENCE();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-fetched tail tiles
}

}  // namespace

long mgx_attn_bwd_pipe_stats_floats(int B, int S, int H) { return 9L * B * H * ((S + 63) / 64 * 64); }

// statistics pre-pass + dK/dV; `stats` holds mgx_attn_bwd_pipe_stats_floats(B, S, H) floats: nla | nda | nd32
int mgx_attn_bwd_dkv_pipe(const void* dout, const void* qkv, const void* out, const float* lse, const float* slopes,
                          void* dqkv, float* stats, int B, int S, int H, int dh, hipStream_t st) {
    const int Sp = (S + 63) / 64 * 64;
    const long rows = (long)B * H * Sp;
    bf16_t* nla = (bf16_t*)stats;
    bf16_t* nda = (bf16_t*)(stats + 4 * rows);
    float* nd32 = stats + 8 * rows;
    const float scale = 1.0f / sqrtf((float)dh);
    const float c = LOG2E * scale;
    const long n_chunks = (long)B * Sp * H * (dh / 8);
    dim3 gs((unsigned)((n_chunks + 255) / 256));
    DkvArgs a{(const bf16_t*)dout, (const bf16_t*)qkv, nla, nda, slopes, (bf16_t*)dqkv, B, S, Sp, H, c, scale, 0, 0, 0};
    a.nkb = (S + 127) / 128;
    a.np = (a.nkb + 1) / 2;
    a.nwork = B * H * a.np;
    const int n_cu = mgx_cu_count();
    MGX_REQUIRE(n_cu > 0, MGX_ELAUNCH, "attention backward: cannot query the device");
    // Persistent workgroups, one resident per CU (registers), TWO per CU in the grid: the second half is handed out by
    // the dispatcher as CUs come free.  On an idle chip that costs nothing measurable (639 against 641 us per backward);
    // when another kernel holds some CUs -- RCCL's reduction kernels during the data-parallel backward -- the work
    // lists are half as long, so the workgroups that had to wait add half as much.
    constexpr int kGridPerCu = 2;
    const int grid = a.nwork < n_cu * kGridPerCu ? a.nwork : n_cu * kGridPerCu;
    if (dh == 64) {
        hipLaunchKernelGGL((attn_bwd_stats_kernel<64>), gs, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                           lse, slopes, nla, nda, nd32, n_chunks, S, Sp, H, 1.f / c);
        MGX_CHECK_LAUNCH("attn_bwd_stats");
        hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<64>), dim3(grid), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_stats_kernel<128>), gs, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                           lse, slopes, nla, nda, nd32, n_chunks, S, Sp, H, 1.f / c);
        MGX_CHECK_LAUNCH("attn_bwd_stats");
        hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<128>), dim3(grid), dim3(256), 0, st, a);
    }
    MGX_CHECK_LAUNCH("attn_bwd_dkv_pipe");
    return MGX_OK;
}
