// bf16 MFMA flash-attention backward for gfx950, dQ kernel: the mirror image of attn_bwd_pipe.hip (same pipeline,
// see the notes there).  Autograd of SimpleAttention.multihead_attn (model_baseline/alibi_unified.py:40-54).
//
// One wave per SIMD (4 waves, 32 queries each: this lane's query row of Q and dO and the 32 x dh dQ accumulators stay
// in registers); persistent workgroups walk (batch, head, 128-query block) items, late (long) and early (short) query
// blocks paired; K / V tiles of 64 keys stream through the 4-slot LDS-DMA ring across item boundaries.  Per unit u of
// 32 keys and wave:
//     A(u)  S^T = K.Q^T + bias - lse, dP^T = V.dO^T - delta     2*(dh/16 + 1) MFMAs: bias, lse and delta enter as one
//                                                                extra k-step whose operands are constants of the item
//     B(u)  P^T = exp2(c S^T + cu), dS^T = P^T * dP^T           plain fp32 VALU
//     C(u)  dQ^T += K^T.dS^T                                     2*dh/32 MFMAs (dS^T straight from the accumulators)
// X(u) = A(u+1) with B(u) and the K / V row fragments of unit u+2 between the MFMAs; Y(u) = C(u) with the K^T
// fragments of unit u+1 between them.
#include "attn_pipe_common.h"

namespace {

// the lgkmcnt field is 4 bits
constexpr int lgkm_cap(int n) { return n < 15 ? n : 15; }

template <int DH>
struct DqCfg {
    static constexpr int NB = 4;                       // LDS ring slots (one 64-key tile each)
    static constexpr int UB = 32 * DH * 2;             // bytes of one 32 x dh bf16 image (a unit)
    static constexpr int TB = 2 * UB;                  // a 64-row tile of K or V
    static constexpr int SLOT = 2 * TB;
    static constexpr int VM_PER_TILE = 4 * (DH / 64);  // LDS-DMA instructions per wave and tile
    static constexpr int KS = DH / 16, DB = DH / 32;
    static constexpr int NG_X = 2 * KS + 2;            // MFMAs of stage A (S^T and dP^T chains, statistics step last)
    static constexpr int NG_Y = 2 * DB;                // MFMAs of stage C
    static constexpr int EPG = DH == 128 ? 1 : 2;      // elements of P per gap of X
    // AGPR map
    static constexpr int R_DQ = 0, R_QF = 16 * DB, R_GF = R_QF + 4 * KS, R_TK = R_GF + 4 * KS;
    static constexpr int R_END = R_TK + 4 * NG_Y;      // 160 at dh 128
};

struct DqArgs {
    const bf16_t* dout;
    const bf16_t* qkv;
    const float* lse;
    const float* nd32;  // -delta, rows padded to Sp
    const float* slopes;
    bf16_t* dqkv;
    int B, S, Sp, H;
    float c, scale;
    int nqb, np;        // 128-query blocks per (b,h); pairs of them
    int nwork;          // B*H*np pairs in all
};

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_pipe_kernel(DqArgs g) {
    using Cfg = DqCfg<DH>;
    constexpr int NB = Cfg::NB, UB = Cfg::UB, TB = Cfg::TB, SLOT = Cfg::SLOT, G8 = Img<DH>::G8;
    constexpr int KS = Cfg::KS, DB = Cfg::DB, NGX = Cfg::NG_X, NGY = Cfg::NG_Y, VM = Cfg::VM_PER_TILE;
    static_assert(Cfg::R_END <= 256, "AGPR map");
    static_assert(2 * VM <= NGX, "one LDS-DMA piece per odd gap of X");
    constexpr int EPI = NB * SLOT;                       // 4 KiB per wave: transposition buffer of the item epilogue
    __shared__ __attribute__((aligned(1024))) char smem[NB * SLOT + 4 * 4096];
    // the AGPRs named in the asm below, made known to the compiler's resource accounting (it allocates none itself)
    asm volatile("" ::: "a0", "a127", "a159");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h5 = lane >> 5;
    const int S = g.S, Sp = g.Sp, H = g.H;
    const unsigned rs = 3u * H * DH;
    const unsigned rs2 = rs * 2u;
    const LaneAddr<DH> la(lane);
    const int tend = Sp / 64;
    const int tlast_real = (S - 1) / 64;                 // last tile that holds real rows

    // ---- this workgroup's items: pair q = (b*H + h, p) -> query blocks nqb-1-p (long) and p (short)
    const int G = gridDim.x;
    const int gq = (G & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    const int nseq = 2 * ((g.nwork + G - 1) / G);        // sequence numbers [0, nseq)
    auto item_at = [&](int n, int& bh, int& qb) -> bool {
        const int q = (n >> 1) * G + gq;
        if (n >= nseq || q >= g.nwork) return false;
        bh = q / g.np;
        const int p = q - bh * g.np;
        qb = (n & 1) ? p : g.nqb - 1 - p;
        return !(n & 1) || qb < g.nqb - 1 - p;
    };
    auto next_item = [&](int n, int& bh, int& qb) -> int {   // first valid sequence number > n, or nseq
        for (++n; n < nseq; ++n)
            if (item_at(n, bh, qb)) return n;
        return nseq;
    };
    auto item_tiles = [&](int qb) { return min(2 * (qb + 1), tend); };   // key tiles [0, 2 (qb+1)) of the item

    // ---- producer: LDS-DMA of one 64-key tile (K rows and V rows of the packed projection, same row stride).  Wave w
    // brings rows 16w..16w+15 of each; lane l of a KiB-piece lands at byte 16 l of the image
    //   off(row, ch) = G8 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)).
    const int rowin = (lane >> 2) & 7;
    unsigned offk[2][DH / 64];                           // per-lane byte offsets from the tile's first row
    int prow[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        prow[gi] = 16 * w + 8 * gi + rowin;
        const int sx = (lane & 3) ^ ((prow[gi] >> 2) & 3);
#pragma unroll
        for (int hf = 0; hf < DH / 64; ++hf)
            offk[gi][hf] = (unsigned)prow[gi] * rs2 + (unsigned)((4 * (2 * hf + (lane >> 5)) + sx) * 16);
    }
    int p_n, p_bh = 0, p_qb = 0, p_tile = 0, p_ntiles = 0;   // producer cursor
    int p_slot = 0;
    const char* p_k = nullptr;                           // K rows of the producer's (b, h); V rows follow H*DH elements on
    auto producer_set_item = [&]() {
        const int b = p_bh / H, h = p_bh - b * H;
        p_k = (const char*)(g.qkv + (long)b * S * rs + (long)h * DH + (long)H * DH);
        p_tile = 0;
        p_ntiles = item_tiles(p_qb);
    };
    const char* p_k0 = nullptr;
    int p_back[2] = {0, 0};
    auto produce_begin = [&]() {
        const int tq = min(p_tile, tlast_real);          // padding tiles re-read real rows (all their keys are masked)
        p_k0 = p_k + (long)tq * 64 * rs2;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) p_back[gi] = max(tq * 64 + prow[gi] - (S - 1), 0);   // rows past the end: row S-1
    };
    auto produce_piece = [&](auto kt) {
        constexpr int k = decltype(kt)::value;
        constexpr int which = k & 1, hf = (k >> 1) % (DH / 64), gi = (k >> 1) / (DH / 64);
        char* dst = smem + p_slot * SLOT + 2 * G8 * w + G8 * gi + 1024 * hf + which * TB;
        const char* src = p_k0 + (offk[gi][hf] - (unsigned)p_back[gi] * rs2) + which * (H * DH * 2);
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_void_t*)dst, 16, 0, 0);
    };
    auto produce_end = [&]() {
        p_slot = (p_slot + 1) & (NB - 1);
        // advance; past the last item the last tile is fetched again (harmless, keeps the counts uniform)
        if (p_tile + 1 < p_ntiles) {
            ++p_tile;
        } else if (p_n < nseq) {
            int bh, qb;
            const int n = next_item(p_n, bh, qb);
            p_n = n;
            if (n < nseq) {
                p_bh = bh;
                p_qb = qb;
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
    bf16x8 rk[KS], rv[KS];           // row fragments of K / V (A operands of S^T / dP^T)
    // statistics step.  S^T[j][i] += ka[j][.] . bq[.][i]:  ka = (t_j split in three, 1, 1, 1, 0, 0) with
    // t_j = slope*log2e/c * (j mod 32), bq = (1, 1, 1, x_i split in three, 0, 0) with x_i = -(slope*log2e/c * (i - i0)
    // + lse_i*log2e/c), i0 the wave's first query; dP^T[j][i] += ones[j][.] . bg[.][i], bg = (-delta_i split in three).
    // Only the lanes of the first half carry them (k = 0..7); the second half (k = 8..15) is zero.
    bf16x8 ka, ones, bq, bg;
    auto split3v = [&](float x, bf16_t& hi, bf16_t& mid, bf16_t& lo) {
        hi = (bf16_t)x;
        const float r1 = x - (float)hi;
        mid = (bf16_t)r1;
        lo = (bf16_t)(r1 - (float)mid);
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (bf16_t)((h5 == 0 && k < 3) ? 1.0f : 0.0f);

    // row fragment fg of the unit at byte `ub`: fg even: K rows (S^T chain), odd: V rows (dP^T chain); k-step fg >> 1
    auto read_row = [&](auto gt, unsigned ub) {
        constexpr int fg = decltype(gt)::value, ks = fg >> 1;
        if constexpr (ks < KS) {
            const unsigned a = ub + ((ks & 1) ? la.row1 : la.row0);
            if constexpr (fg & 1)
                rv[ks] = lds_b128<TB + 512 * (ks >> 1)>(a);
            else
                rk[ks] = lds_b128<512 * (ks >> 1)>(a);
        }
    };
    // transposed fragment f of stage C of the unit at byte `ub`: K^T -> a[R_TK + 4 f ..]; (s2, d) = (f / DB, f % DB)
    auto read_tr = [&](auto ft, unsigned ub) {
        constexpr int f = decltype(ft)::value;
        constexpr int d = f % DB, s2 = f / DB;
        constexpr int o0 = G8 * (2 * s2) + 512 * d, o1 = o0 + G8;
        lds_tr128_agpr<Cfg::R_TK + 4 * f, o0, o1>(ub + la.tr0, ub + la.tr1);
    };
    // stage A: S^T and dP^T chains, alternating (gap index fg: even = S^T, odd = dP^T)
    auto mfma_a_step = [&](auto gt, f32x16& sc, f32x16& dp) {
        constexpr int fg = decltype(gt)::value, ks = fg >> 1;
        if constexpr (fg == 0)
            mfma_s0<Cfg::R_QF>(sc, rk[0]);
        else if constexpr (fg == 1)
            mfma_s0<Cfg::R_GF>(dp, rv[0]);
        else if constexpr (ks == KS) {
            if constexpr (fg & 1)
                mfma_sv(dp, ones, bg);
            else
                mfma_sv(sc, ka, bq);
        } else if constexpr (fg & 1)
            mfma_s<Cfg::R_GF + 4 * ks>(dp, rv[ks]);
        else
            mfma_s<Cfg::R_QF + 4 * ks>(sc, rk[ks]);
    };
    auto load_q = [&](int bh, int qb) {
        // this lane's query row of Q and dO straight into AGPRs; queries past the end read row S-1 (not stored)
        const int b = bh / H, h = bh - b * H;
        const long ir = min(qb * 128 + 32 * w + l31, S - 1);
        const bf16_t* qp = g.qkv + ((long)b * S + ir) * rs + (long)h * DH + 8 * h5;
        const bf16_t* gp = g.dout + (((long)b * S + ir) * H + h) * DH + 8 * h5;
        sfor<KS>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            global_b128_to_agpr<Cfg::R_QF + 4 * ks>(qp + ks * 16);
            global_b128_to_agpr<Cfg::R_GF + 4 * ks>(gp + ks * 16);
        });
    };

    // ---- start: first item, its Q / dO rows, three tiles in flight
    int c_n, c_bh = 0, c_qb = 0;
    c_n = next_item(-1, c_bh, c_qb);
    if (c_n >= nseq) return;                             // (uniform) nothing to do for this workgroup
    p_n = c_n;
    p_bh = c_bh;
    p_qb = c_qb;
    producer_set_item();
    load_q(c_bh, c_qb);
    produce();
    produce();
    produce();
    sfor<16 * DB>([&](auto rt) { agpr_zero<Cfg::R_DQ + decltype(rt)::value>(); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VM) : "memory");      // Q, dO rows and the first tile
    PIPE_FENCE();
    __builtin_amdgcn_s_barrier();
    PIPE_FENCE();

    int slot_c = 0;                                      // slot of the tile being consumed
    f32x16 scA, dpA, scB, dpB;
    while (true) {
        // ================= one item: query block c_qb of (b, h) = c_bh
        const int b = c_bh / H, h = c_bh - b * H;
        const int i0w = c_qb * 128 + 32 * w;             // the wave's first query
        const int i = i0w + l31;
        const float csb = g.slopes[h] * LOG2E;           // bias per key step in log2 units
        const float inv_c = 1.f / g.c;
        const int ntiles = item_tiles(c_qb);
        const int umask = 4 * c_qb;                      // units >= umask can touch the causal diagonal
        float cu = -csb * (float)i0w;                    // bias of the unit's first key against the wave's first query
        {
            const int ic = min(i, S - 1);
            const float x = -(csb * (float)l31 + g.lse[((long)b * H + h) * S + ic] * LOG2E) * inv_c;
            const float nd = g.nd32[((long)b * H + h) * Sp + ic];
            const float tj = csb * inv_c * (float)l31;
            bf16_t a0, a1, a2, b0, b1, b2, c0, c1, c2;
            split3v(tj, a0, a1, a2);
            split3v(x, b0, b1, b2);
            split3v(nd, c0, c1, c2);
            const bf16_t one = (bf16_t)1.0f, z = (bf16_t)0.0f;
            if (h5 == 0) {
                ka = bf16x8{a0, a1, a2, one, one, one, z, z};
                bq = bf16x8{one, one, one, b0, b1, b2, z, z};
                bg = bf16x8{c0, c1, c2, z, z, z, z, z};
            } else {
                ka = bq = bg = bf16x8{z, z, z, z, z, z, z, z};
            }
        }
        int n_bh = 0, n_qb = 0;
        const int n_n = next_item(c_n, n_bh, n_qb);
        {
            // not overlapped: A(0), then the row fragments of unit 1 and the K^T fragments of unit 0
            const unsigned ub = lds0 + slot_c * SLOT;
            sfor<2 * KS>([&](auto gt) { read_row(gt, ub); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIPE_FENCE();
            sfor<NGX>([&](auto gt) { mfma_a_step(gt, scA, dpA); });
            PIPE_FENCE();
            sfor<2 * KS>([&](auto gt) { read_row(gt, ub + UB); });
            sfor<NGY>([&](auto ft) { read_tr(ft, ub); });
            PIPE_FENCE();
        }

        // One unit (tile t, half HB).  (sc, dp) hold S^T / dP^T of unit u (from the previous X), (scn, dpn) receive u + 1.
        auto unit = [&](int t, auto hb_tag, f32x16& sc, f32x16& dp, f32x16& scn, f32x16& dpn) {
            constexpr int HB = decltype(hb_tag)::value;
            const int u = 2 * t + HB;
            const int slot_n = (slot_c + 1) & (NB - 1);
            const unsigned unext = lds0 + slot_n * SLOT + HB * UB;                        // unit u + 2
            const unsigned u1 = HB ? lds0 + slot_n * SLOT : lds0 + slot_c * SLOT + UB;    // unit u + 1
            // rows of u+1 (read during X(u-1)) must have landed; the newest reads (K^T of unit u) may fly
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(lgkm_cap(2 * NGY)) : "memory");
            PIPE_FENCE();
            if (HB && t + 1 == ntiles && n_n < nseq) load_q(n_bh, n_qb);     // Q / dO rows are dead: A(u+1) is never used
            if (u >= umask) {                             // wave-uniform, at most 4 units per item
                const int di = i - 32 * u - 4 * h5;       // key row part > di  <=>  key after query
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA of A(u) must have landed
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((e & 3) + 8 * (e >> 2) > di) sc[e] = -INFINITY;
            }
            // ---------------- X(u)
            PIPE_FENCE();
            unsigned dsf[8];                              // bf16 pairs: dS^T of unit u
            float pe[16], de[16];
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
                read_row(gt, unext);
                // P^T of this gap's elements, dS^T of the previous gap's: the multiply is inline asm, which the
                // compiler's hazard recogniser does not look into, and a VALU that reads the result of the v_exp_f32
                // right before it gets the OLD register contents on some lanes (transcendental forwarding hazard)
                if constexpr (fg >= 1 && fg <= 16 / Cfg::EPG) {
#pragma unroll
                    for (int q = 0; q < Cfg::EPG; ++q) {
                        const int e = (fg - 1) * Cfg::EPG + q;
                        de[e] = mul_np(pe[e], dp[e]);
                        if (e & 1) dsf[e >> 1] = pack2(de[e - 1], de[e]);
                    }
                }
                if constexpr (fg < 16 / Cfg::EPG) {
#pragma unroll
                    for (int q = 0; q < Cfg::EPG; ++q) {
                        const int e = fg * Cfg::EPG + q;
                        pe[e] = fast_exp2(fma_np(sc[e], g.c, cu));
                    }
                }
                PIPE_FENCE();
            });
            // ---------------- Y(u): dQ MFMAs with the K^T fragments of unit u+1 between them
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(lgkm_cap(2 * KS)) : "memory");   // K^T fragments of unit u
            PIPE_FENCE();
            bf16x8 db[2] = {words_to_frag(dsf[0], dsf[1], dsf[2], dsf[3]),
                                  words_to_frag(dsf[4], dsf[5], dsf[6], dsf[7])};
            pin_frags(db[0], db[1]);                      // VALU write of dS -> MFMA read
            PIPE_FENCE();
            sfor<NGY>([&](auto gt) {
                constexpr int f = decltype(gt)::value;
                constexpr int d = f % DB, s2 = f / DB;
                mfma_acc<Cfg::R_DQ + 16 * d, Cfg::R_TK + 4 * f>(db[s2]);
                PIPE_FENCE();
                read_tr(gt, u1);
                PIPE_FENCE();
            });
            cu += 32.f * csb;
        };

        for (int t = 0; t < ntiles; ++t) {
            unit(t, std::integral_constant<int, 0>{}, scA, dpA, scB, dpB);
            unit(t, std::integral_constant<int, 1>{}, scB, dpB, scA, dpA);
            slot_c = (slot_c + 1) & (NB - 1);
        }

        // ---- item done.  The last MFMAs must have written dQ before it is read back (nothing tells the compiler).
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // The next item's resident rows (requested a unit ago, younger than every tile request in flight) are waited
        // for HERE, before the epilogue's stores are issued: after them a vmcnt(0) would also wait for the stores'
        // acknowledgements.  The counted tile waits of the next item stay valid with stores outstanding: loads
        // retire in order among themselves, stores only make the count conservative.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_FENCE();
        {
            // dQ^T has the query on the lane and dh down the registers: each wave turns its 32 queries x 64 columns
            // at a time through 4 KiB of LDS (16-byte chunks XOR-swizzled with the row) and stores 128-byte row segments
            const unsigned eb = lds0 + EPI + 4096 * w;
            const int erow = lane >> 3, ech = lane & 7;
            bf16_t* const orow = g.dqkv + ((long)b * S + i0w) * rs + (long)h * DH;
            sfor<DH / 64>([&](auto pt) {
                constexpr int half = decltype(pt)::value;
                sfor<2>([&](auto dt) {
                    constexpr int dbk = 2 * half + decltype(dt)::value;
                    sfor<4>([&](auto gt) {
                        constexpr int gg = decltype(gt)::value;
                        constexpr int r0 = Cfg::R_DQ + 16 * dbk + 4 * gg;
                        bf16x4 v;
                        v[0] = (bf16_t)(agpr_get<r0 + 0>() * g.scale);
                        v[1] = (bf16_t)(agpr_get<r0 + 1>() * g.scale);
                        v[2] = (bf16_t)(agpr_get<r0 + 2>() * g.scale);
                        v[3] = (bf16_t)(agpr_get<r0 + 3>() * g.scale);
                        const int c16 = 4 * (dbk & 1) + gg;
                        lds_write_b64(eb + l31 * 128 + ((c16 ^ (l31 & 7)) * 16) + 8 * h5, v);
                    });
                });
                // the wave reads back what it wrote itself: LDS operations of one wave complete in order
                bf16x8 ov[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 8 * r + erow;
                    ov[r] = lds_b128<0>(eb + row * 128 + ((ech ^ (row & 7)) * 16));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PIPE_FENCE();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 8 * r + erow;
                    if (i0w + row < S) *reinterpret_cast<bf16x8*>(orow + (long)row * rs + 64 * half + 8 * ech) = ov[r];
                }
            });
        }
        PIPE_FENCE();
        if (n_n >= nseq) break;
        c_n = n_n;
        c_bh = n_bh;
        c_qb = n_qb;
        sfor<16 * DB>([&](auto rt) { agpr_zero<Cfg::R_DQ + decltype(rt)::value>(); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIPE_FENCE();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-fetched tail tiles
}

}  // namespace

// dQ with the statistics of attn_bwd_pipe.hip's pre-pass (nd32 = -delta, rows padded to Sp)
int mgx_attn_bwd_dq_pipe(const void* dout, const void* qkv, const float* lse, const float* nd32, const float* slopes,
                         void* dqkv, int B, int S, int H, int dh, hipStream_t st) {
    const int Sp = (S + 63) / 64 * 64;
    const float scale = 1.0f / sqrtf((float)dh);
    const float c = LOG2E * scale;
    DqArgs a{(const bf16_t*)dout, (const bf16_t*)qkv, lse, nd32, slopes, (bf16_t*)dqkv, B, S, Sp, H, c, scale, 0, 0, 0};
    a.nqb = (S + 127) / 128;
    a.np = (a.nqb + 1) / 2;
    a.nwork = B * H * a.np;
    const int n_cu = mgx_cu_count();
    MGX_REQUIRE(n_cu > 0, MGX_ELAUNCH, "attention backward: cannot query the device");
    // Persistent workgroups, one resident per CU (registers), TWO per CU in the grid: the second half is handed out by
    // the dispatcher as CUs come free.  On an idle chip that costs nothing measurable (639 against 641 us per backward);
    // when another kernel holds some CUs -- RCCL's reduction kernels during the data-parallel backward -- the work
    // lists are half as long, so the workgroups that had to wait add half as much.
    constexpr int kGridPerCu = 2;
    const int grid = a.nwork < n_cu * kGridPerCu ? a.nwork : n_cu * kGridPerCu;
    if (dh == 64)
        hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<64>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attn_bwd_dq_pipe_kernel<128>), dim3(grid), dim3(256), 0, st, a);
    MGX_CHECK_LAUNCH("attn_bwd_dq_pipe");
    return MGX_OK;
}
