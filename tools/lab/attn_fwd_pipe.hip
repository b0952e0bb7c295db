// LAB, not part of libmgx (round 3): the forward on the backward kernels' one-wave-per-SIMD pipeline with 64 QUERIES
// PER WAVE and the keys walked backwards from the diagonal.  Correct (passes every bf16 attention test when wired
// into mgx_attn_fwd_mfma), but NOT faster than attn_fwd_mfma_kernel: B16 S2048 H8 dh128 237-242 us against 203-215 us,
// dh64 166 against 130 us, B8 S4096 dh64 278 against 223 us.  Ablations at 240 us (one part compiled out): no exp 219,
// no LDS-DMA 211, no softmax VALU 208, no mask / maxima / rescale 200, MFMAs + fragment reads + DMA alone 176.  SQ
// counters: the wave issues instructions in 48 % of its cycles (VALU 37 %: 284 VALU instructions per 64x32 unit against
// 34 MFMAs), waits at s_waitcnt / barrier in 29 % (8 % of that is the tiles a wave skips above its diagonal), MFMA
// busy 32 %.  Total instruction-issue cycles per SIMD (SQ_ACTIVE_INST_ANY) are the same as in the shipped kernel
// (56 M against 54 M per launch = ~215 k cycles per SIMD = ~113 us): with one wave per SIMD nothing fills the waits,
// with two (the shipped kernel) the partner wave does.  To build it: copy into musicgeneration_amd/csrc/ and return
// mgx_attn_fwd_pipe(...) from mgx_attn_fwd_mfma (mode 0).  It replaces the 32-query lab port of round 2.
// bf16 MFMA flash-attention forward for gfx950 on the hand-placed pipeline of the backward kernels
// (attn_bwd_pipe.hip, attn_bwd_dq_pipe.hip; building blocks in attn_pipe_common.h), 64 QUERIES PER WAVE.
// SimpleAttention.multihead_attn (model_baseline/alibi_unified.py:40-54): causal softmax of
// q.k^T / sqrt(dh) + slope_h (j - i), times v; nothing S x S reaches HBM, the [H,S,S] mask buffer
// (alibi_unified.py:26-31) does not exist.
//
// One wave per SIMD (4 waves); a wave owns TWO 32-query blocks a / b (queries i0w.. and i0w + 32..): their rows of Q
// and the 2 x 32 x dh O accumulators stay in hand-allocated AGPRs, and every K row fragment and every V^T fragment
// read from LDS feeds two MFMAs (the 32-query kernel, attn_mfma.hip, needs a fresh 1-KiB fragment per MFMA).
// Persistent workgroups walk (batch, head, 256-query block) items, late (long) and early (short) query blocks paired;
// K / V tiles of 64 keys stream through a 4-slot LDS-DMA ring across item boundaries.  Keys are walked from the
// diagonal BACKWARDS (with ALiBi the largest scores sit next to the diagonal: the running maximum is right after
// the first unit and the deferred rescale of O - 128 AGPR read / multiply / write triples here - stays the exception).
// A wave skips the item's tiles that lie wholly above its own diagonal (it only keeps the ring going).
// Per unit u of 32 keys and wave:
//     A(u)  S^T = K.Q^T + bias for a and b     2 (dh/16 + 1) MFMAs: the bias enters as one extra k-step whose operands
//                                               are constants of the item (key part x ones + ones x query part)
//     B(u)  P^T = exp2(c S^T + cu - m)         online softmax with the deferred rescale of attn_mfma.hip
//     C(u)  O^T += V^T.P^T for a and b         4 dh/32 MFMAs (P^T straight from the accumulators)
// X(u) = A(u+1) with the exponentials of unit u and the K row fragments of unit u+2 between the MFMAs; Y(u) = C(u)
// with the V^T fragments of unit u+1 and the mask + row maxima of unit u+1 between them; the (rare, wave-uniform)
// rescale of O sits between Y(u) and X(u+1).
#include "../../musicgeneration_amd/csrc/attn_pipe_common.h"

namespace {

constexpr int fwd_lgkm_cap(int n) { return n < 15 ? n : 15; }     // the lgkmcnt field is 4 bits

template <int DH>
struct FwdCfg {
    static constexpr int NB = 4;                       // LDS ring slots (one 64-key tile each)
    static constexpr int UB = 32 * DH * 2;             // bytes of one 32 x dh bf16 image (a unit)
    static constexpr int TB = 2 * UB;                  // a 64-row tile of K or V
    static constexpr int SLOT = 2 * TB;
    static constexpr int VM_PER_TILE = 4 * (DH / 64);  // LDS-DMA instructions per wave and tile
    static constexpr int PPU = VM_PER_TILE / 2;        // ... issued per unit, at the odd gaps of X
    static constexpr int KS = DH / 16, DB = DH / 32;
    static constexpr int NG_X = 2 * (KS + 1);          // MFMAs of stage A: (k-step, query block) pairs, the bias steps last
    static constexpr int NF = 2 * DB;                  // V^T fragments of a unit
    static constexpr int NG_Y = 2 * NF;                // MFMAs of stage C: (fragment, query block) pairs
    static constexpr int EPG = 16 / KS;                // elements of P per gap of X (32 elements over the 2 KS first gaps)
    // AGPR map
    static constexpr int R_O = 0;                      // O of block a, then of block b: 16 DB each
    static constexpr int R_QF = 32 * DB;               // Q rows of block a, then of block b: 4 KS each
    static constexpr int R_TV = R_QF + 8 * KS;
    static constexpr int R_END = R_TV + 4 * NF;        // 224 at dh 128, 112 at dh 64
};

struct FwdArgs {
    const bf16_t* qkv;
    const float* slopes;
    bf16_t* out;
    float* lse;
    int B, S, Sp, H;
    float c;            // scale * log2(e)
    int nqb, np;        // 256-query blocks per (b,h); pairs of them
    int nwork;          // B*H*np pairs in all
};

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_fwd_pipe_kernel(FwdArgs g) {
    using Cfg = FwdCfg<DH>;
    constexpr int NB = Cfg::NB, UB = Cfg::UB, TB = Cfg::TB, SLOT = Cfg::SLOT, G8 = Img<DH>::G8;
    constexpr int KS = Cfg::KS, DB = Cfg::DB, NGX = Cfg::NG_X, NF = Cfg::NF, NGY = Cfg::NG_Y, VM = Cfg::VM_PER_TILE,
                  PPU = Cfg::PPU, EPG = Cfg::EPG;
    static_assert(Cfg::R_END <= 256, "AGPR map");
    static_assert(2 * PPU - 1 < NGX, "one LDS-DMA piece per odd gap of X");
    static_assert(NGY >= 6, "mask + row maxima of the next unit sit at gaps 3 and 5 of Y");
    static_assert(EPG * 2 * KS == 32, "exponentials per gap");
    constexpr int EPI = NB * SLOT;                       // 4 KiB per wave: transposition buffer of the item epilogue
    __shared__ __attribute__((aligned(1024))) char smem[NB * SLOT + 4 * 4096];
    // the AGPRs named in the asm below, made known to the compiler's resource accounting (it allocates none itself)
    asm volatile("" ::: "a0", "a63", "a127", "a223");
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
    auto item_tiles = [&](int qb) { return min(4 * (qb + 1), tend); };   // key tiles [0, 4 (qb+1)) of the item

    // ---- producer: LDS-DMA of one 64-key tile (K rows and V rows of the packed projection, same row stride), the
    // item's tiles in DESCENDING order.  Wave w brings rows 16w..16w+15 of each; lane l of a KiB-piece lands at byte
    // 16 l of the image  off(row, ch) = G8 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)).
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
    int p_n, p_bh = 0, p_qb = 0, p_tile = 0;             // producer cursor: item sequence number, tile inside it
    int p_slot = 0;
    const char* p_k = nullptr;                           // K rows of the producer's (b, h); V rows follow H*DH elements on
    auto producer_set_item = [&]() {
        const int b = p_bh / H, h = p_bh - b * H;
        p_k = (const char*)(g.qkv + (long)b * S * rs + (long)h * DH + (long)H * DH);
        p_tile = item_tiles(p_qb) - 1;
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
        if (p_tile > 0) {
            --p_tile;
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
    bf16x8 rk[KS];                   // row fragments of K (A operand of S^T), shared by the two query blocks
    // bias step.  S^T[j][i] += ka[j][.] . bq[.][i]:  ka = (t_j split in three, 1, 1, 1, 0, 0) with
    // t_j = slope*log2e/c * (j mod 32), bq = (1, 1, 1, x_i split in three, 0, 0) with x_i = -slope*log2e/c * (i - i0),
    // i0 the block's first query.  Only the lanes of the first half carry them (k = 0..7); the second half is zero.
    // Both blocks have the same x_i (their 32-query offset goes into the unit bias cu).
    bf16x8 ka, bq;
    auto split3v = [&](float x, bf16_t& hi, bf16_t& mid, bf16_t& lo) {
        hi = (bf16_t)x;
        const float r1 = x - (float)hi;
        mid = (bf16_t)r1;
        lo = (bf16_t)(r1 - (float)mid);
    };

    // K row fragment ks of the unit at byte `ub`
    auto read_row = [&](auto kt, unsigned ub) {
        constexpr int ks = decltype(kt)::value;
        rk[ks] = lds_b128<512 * (ks >> 1)>(ub + ((ks & 1) ? la.row1 : la.row0));
    };
    // transposed fragment f of stage C of the unit at byte `ub`: V^T -> a[R_TV + 4 f ..]; (s2, d) = (f / DB, f % DB)
    auto read_tr = [&](auto ft, unsigned ub) {
        constexpr int f = decltype(ft)::value;
        constexpr int d = f % DB, s2 = f / DB;
        constexpr int o0 = TB + G8 * (2 * s2) + 512 * d, o1 = o0 + G8;
        lds_tr128_agpr<Cfg::R_TV + 4 * f, o0, o1>(ub + la.tr0, ub + la.tr1);
    };
    // stage A, gap fg: k-step fg >> 1 of query block fg & 1; the two bias steps last
    auto mfma_a_step = [&](auto gt, f32x16& sa, f32x16& sb) {
        constexpr int fg = decltype(gt)::value, ks = fg >> 1, qb = fg & 1;
        f32x16& sc = qb ? sb : sa;
        if constexpr (ks == 0)
            mfma_s0<Cfg::R_QF + 4 * KS * qb>(sc, rk[0]);
        else if constexpr (ks == KS)
            mfma_sv(sc, ka, bq);
        else
            mfma_s<Cfg::R_QF + 4 * KS * qb + 4 * ks>(sc, rk[ks]);
    };
    auto load_q = [&](int bh, int qb) {
        // this lane's two query rows straight into AGPRs; queries past the end read row S-1 (not stored)
        const int b = bh / H, h = bh - b * H;
        const bf16_t* qp0 = g.qkv + (long)b * S * rs + (long)h * DH + 8 * h5;
        const long ia = min(qb * 256 + 64 * w + l31, S - 1), ib = min(qb * 256 + 64 * w + 32 + l31, S - 1);
        sfor<KS>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            global_b128_to_agpr<Cfg::R_QF + 4 * ks>(qp0 + ia * rs + ks * 16);
            global_b128_to_agpr<Cfg::R_QF + 4 * KS + 4 * ks>(qp0 + ib * rs + ks * 16);
        });
    };
    // a tile this wave has nothing to compute on: keep the ring and the barrier count going
    auto skip_tile = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        PIPE_FENCE();
        __builtin_amdgcn_s_barrier();
        PIPE_FENCE();
        produce();
        PIPE_FENCE();
    };

    // ---- start: first item, its Q rows, three tiles in flight
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
    sfor<32 * DB>([&](auto rt) { agpr_zero<Cfg::R_O + decltype(rt)::value>(); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * VM) : "memory");      // Q rows and the first tile
    PIPE_FENCE();
    __builtin_amdgcn_s_barrier();
    PIPE_FENCE();

    int slot_c = 0;                                      // slot of the tile being consumed
    f32x16 scAa, scAb, scBa, scBb;
    while (true) {
        // ================= one item: query block c_qb of (b, h) = c_bh
        const int b = c_bh / H, h = c_bh - b * H;
        const int i0w = c_qb * 256 + 64 * w;             // the wave's first query (block a); block b starts 32 on
        const int ia = i0w + l31;
        const float csb = g.slopes[h] * LOG2E;           // bias per key step in log2 units
        const float inv_c = 1.f / g.c;
        const int ntiles = item_tiles(c_qb);
        // the wave's own diagonal tile is tile i0w / 64 of the item; the tiles above it hold no key <= its queries
        const int nskip = max(0, ntiles - 1 - (i0w >> 6));
        const int nun = 2 * (ntiles - nskip);            // units this wave computes, walked from key unit ku0 down to 0
        const int ku0 = nun - 1;
        float cu = csb * (float)(32 * ku0 - i0w);        // bias of the unit's first key against block a's first query
        const float cb = -32.f * csb;                    // ... block b's first query lies 32 further on
        float ma = -1e30f, mb = -1e30f, la_ = 0.f, lb_ = 0.f;   // running maxima (log2 units, bias included) and sums
        float cuma = 0.f, cumb = 0.f;                    // cu - m of the unit whose P is formed next
        {
            const float x = -(csb * (float)l31) * inv_c;
            const float tj = csb * inv_c * (float)l31;
            bf16_t a0, a1, a2, b0, b1, b2;
            split3v(tj, a0, a1, a2);
            split3v(x, b0, b1, b2);
            const bf16_t one = (bf16_t)1.0f, z = (bf16_t)0.0f;
            if (h5 == 0) {
                ka = bf16x8{a0, a1, a2, one, one, one, z, z};
                bq = bf16x8{one, one, one, b0, b1, b2, z, z};
            } else {
                ka = bq = bf16x8{z, z, z, z, z, z, z, z};
            }
        }
        int n_bh = 0, n_qb = 0;
        const int n_n = next_item(c_n, n_bh, n_qb);

        // causal mask of key unit ku on the raw scores of the block whose first query is iq0, then this lane's maximum
        // over its 16 keys
        auto mask_max = [&](f32x16& sc, int ku, int iq0) -> float {
            if (32 * ku + 31 > iq0) {                    // wave-uniform, the units at and above the block's diagonal
                const int di = iq0 + l31 - 32 * ku - 4 * h5;     // key row part > di  <=>  key after query
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((e & 3) + 8 * (e >> 2) > di) sc[e] = -INFINITY;
            }
            float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = fmaxf(mx, fmaxf(sc[e], sc[e + 1]));
            return mx;
        };
        // the row maxima of the next unit are known: raise the running maxima if some row grew by more than 2^8 (then
        // rescale l and O: the C MFMAs of the previous unit must have written O first), set cum for its P
        auto commit_max = [&](float mxa, float mxb, float cu_u) {
            mxa = half_swap_max(mxa);
            mxb = half_swap_max(mxb);
            // log2 units, unit bias included; a wholly masked unit (-inf) leaves the maximum where it is
            const float va = fmaxf(fmaf(mxa, g.c, cu_u), -1e30f), vb = fmaxf(fmaf(mxb, g.c, cu_u + cb), -1e30f);
            if (!__all(va - ma <= RESCALE_LOG2 && vb - mb <= RESCALE_LOG2)) {
                const float na = fmaxf(ma, va), nb = fmaxf(mb, vb);
                const float alpha_a = fast_exp2(ma - na), alpha_b = fast_exp2(mb - nb);
                ma = na;
                mb = nb;
                la_ *= alpha_a;
                lb_ *= alpha_b;
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                sfor<16 * DB>([&](auto rt) {
                    constexpr int R = Cfg::R_O + decltype(rt)::value;
                    agpr_put<R>(agpr_get<R>() * alpha_a);
                    agpr_put<R + 16 * DB>(agpr_get<R + 16 * DB>() * alpha_b);
                });
                asm volatile("s_nop 7" ::: "memory");    // VALU write of an AGPR -> MFMA read as C
            }
            cuma = cu_u - ma;
            cumb = cu_u + cb - mb;
        };

        for (int t = 0; t < nskip; ++t) {
            skip_tile();
            slot_c = (slot_c + 1) & (NB - 1);
        }
        {
            // not overlapped: A(0) on the upper half of the wave's first tile, then the K rows of unit 1 (its lower
            // half), the V^T fragments of unit 0, the maxima of unit 0
            const unsigned ub = lds0 + slot_c * SLOT;
            sfor<KS>([&](auto kt) { read_row(kt, ub + UB); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PIPE_FENCE();
            sfor<NGX>([&](auto gt) { mfma_a_step(gt, scAa, scAb); });
            PIPE_FENCE();
            sfor<KS>([&](auto kt) { read_row(kt, ub); });
            sfor<NF>([&](auto ft) { read_tr(ft, ub + UB); });
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA of A(0) must have landed
            PIPE_FENCE();
            const float mxa = mask_max(scAa, ku0, i0w);
            const float mxb = mask_max(scAb, ku0, i0w + 32);
            commit_max(mxa, mxb, cu);
            PIPE_FENCE();
        }

        // One unit: walk index u = 2 t + HB, key unit ku0 - u = half 1 - HB of the tile in slot_c.  (sa, sb) hold the
        // masked S^T of unit u (from the previous X), (sna, snb) receive unit u + 1.
        auto unit = [&](int t, auto hb_tag, f32x16& sa, f32x16& sb, f32x16& sna, f32x16& snb) {
            constexpr int HB = decltype(hb_tag)::value;
            const int u = 2 * t + HB;
            const int slot_n = (slot_c + 1) & (NB - 1);
            const unsigned unext = lds0 + slot_n * SLOT + (1 - HB) * UB;                  // unit u + 2
            const unsigned u1 = HB ? lds0 + slot_n * SLOT + UB : lds0 + slot_c * SLOT;    // unit u + 1
            const bool last = HB && 2 * t + 2 == nun;    // A(u+1) of the wave's last unit is never used
            // K rows of u+1 (read during X(u-1)) must have landed; the newest reads (V^T of unit u) may fly
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(fwd_lgkm_cap(2 * NF)) : "memory");
            PIPE_FENCE();
            if (last && n_n < nseq) load_q(n_bh, n_qb);  // the Q rows are dead
            // ---------------- X(u)
            PIPE_FENCE();
            unsigned pfa[8], pfb[8];                      // bf16 pairs: P^T of unit u, blocks a and b
            float pe[32];
            sfor<NGX>([&](auto gt) {
                constexpr int fg = decltype(gt)::value;
                mfma_a_step(gt, sna, snb);
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
                if constexpr ((fg & 1) && (fg >> 1) < PPU) {
                    produce_piece(std::integral_constant<int, HB * PPU + (fg >> 1)>{});
                    if constexpr (HB == 1 && (fg >> 1) == PPU - 1) produce_end();
                }
                // K row fragment ks has fed both of its MFMAs after the odd gap
                if constexpr ((fg & 1) && (fg >> 1) < KS) read_row(std::integral_constant<int, (fg >> 1)>{}, unext);
                // P^T of this gap's elements; sum and pack of the previous gap's (a gap apart: the add is inline asm,
                // which the hazard recogniser does not look into -- transcendental forwarding hazard).  Elements 0..15
                // are block a's, 16..31 block b's.
                if constexpr (fg >= 1 && fg <= 32 / EPG) {
#pragma unroll
                    for (int q = 0; q < EPG; ++q) {
                        const int e = (fg - 1) * EPG + q;
                        if (e < 16) {
                            la_ = add_np(la_, pe[e]);
                            if (e & 1) pfa[e >> 1] = pack2(pe[e - 1], pe[e]);
                        } else {
                            lb_ = add_np(lb_, pe[e]);
                            if (e & 1) pfb[(e - 16) >> 1] = pack2(pe[e - 1], pe[e]);
                        }
                    }
                }
                if constexpr (fg < 32 / EPG) {
#pragma unroll
                    for (int q = 0; q < EPG; ++q) {
                        const int e = fg * EPG + q;
                        pe[e] = e < 16 ? fast_exp2(fma_np(sa[e], g.c, cuma)) : fast_exp2(fma_np(sb[e - 16], g.c, cumb));
                    }
                }
                PIPE_FENCE();
            });
            // ---------------- Y(u): O MFMAs with the V^T fragments of unit u+1 between them
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(fwd_lgkm_cap(KS)) : "memory");   // V^T fragments of unit u
            PIPE_FENCE();
            bf16x8 pba[2] = {words_to_frag(pfa[0], pfa[1], pfa[2], pfa[3]), words_to_frag(pfa[4], pfa[5], pfa[6], pfa[7])};
            bf16x8 pbb[2] = {words_to_frag(pfb[0], pfb[1], pfb[2], pfb[3]), words_to_frag(pfb[4], pfb[5], pfb[6], pfb[7])};
            pin_frags(pba[0], pba[1]);                    // VALU write of P -> MFMA read
            pin_frags(pbb[0], pbb[1]);
            PIPE_FENCE();
            float mxa = 0.f, mxb = 0.f;
            sfor<NGY>([&](auto gt) {
                constexpr int gy = decltype(gt)::value, f = gy >> 1, qb = gy & 1;
                constexpr int d = f % DB, s2 = f / DB;
                if constexpr (qb)
                    mfma_acc<Cfg::R_O + 16 * DB + 16 * d, Cfg::R_TV + 4 * f>(pbb[s2]);
                else
                    mfma_acc<Cfg::R_O + 16 * d, Cfg::R_TV + 4 * f>(pba[s2]);
                PIPE_FENCE();
                if constexpr (qb) read_tr(std::integral_constant<int, f>{}, u1);     // fragment f has fed both MFMAs
                // three MFMAs after the last one of A(u+1): its results have landed
                if constexpr (gy == 3) {
                    if (!last) mxa = mask_max(sna, ku0 - u - 1, i0w);
                }
                if constexpr (gy == 5) {
                    if (!last) mxb = mask_max(snb, ku0 - u - 1, i0w + 32);
                }
                PIPE_FENCE();
            });
            cu -= 32.f * csb;
            if (!last) commit_max(mxa, mxb, cu);
            PIPE_FENCE();
        };

        for (int t = 0; t < ntiles - nskip; ++t) {
            unit(t, std::integral_constant<int, 0>{}, scAa, scAb, scBa, scBb);
            unit(t, std::integral_constant<int, 1>{}, scBa, scBb, scAa, scAb);
            slot_c = (slot_c + 1) & (NB - 1);
        }

        // ---- item done.  The last MFMAs must have written O before it is read back (nothing tells the compiler).
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // the next item's Q rows were requested at the top of the last unit, before that unit's PPU LDS-DMA pieces: loads
        // retire in order, so at most PPU outstanding means they have landed.  Waited for before the epilogue's stores
        // are issued (see attn_bwd_pipe.hip).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPU) : "memory");
        PIPE_FENCE();
        {
            // O^T has the query on the lane and dh down the registers: each wave turns 32 queries x 64 columns at a
            // time through 4 KiB of LDS (16-byte chunks XOR-swizzled with the row) and stores 128-byte row segments
            const unsigned eb = lds0 + EPI + 4096 * w;
            const int erow = lane >> 3, ech = lane & 7;
            const long ors = (long)H * DH;
            sfor<2>([&](auto qt) {
                constexpr int qb = decltype(qt)::value;
                const int i0 = i0w + 32 * qb, i = i0 + l31;
                const float lt = half_swap_sum(qb ? lb_ : la_);
                const float inv = 1.f / lt;
                if (h5 == 0 && i < S) g.lse[((long)b * H + h) * S + i] = ((qb ? mb : ma) + log2f(lt)) * LN2;
                bf16_t* const orow = g.out + (((long)b * S + i0) * H + h) * DH;
                sfor<DH / 64>([&](auto pt) {
                    constexpr int half = decltype(pt)::value;
                    sfor<2>([&](auto dt) {
                        constexpr int dbk = 2 * half + decltype(dt)::value;
                        sfor<4>([&](auto gt) {
                            constexpr int gg = decltype(gt)::value;
                            constexpr int r0 = Cfg::R_O + 16 * DB * qb + 16 * dbk + 4 * gg;
                            bf16x4 v;
                            v[0] = (bf16_t)(agpr_get<r0 + 0>() * inv);
                            v[1] = (bf16_t)(agpr_get<r0 + 1>() * inv);
                            v[2] = (bf16_t)(agpr_get<r0 + 2>() * inv);
                            v[3] = (bf16_t)(agpr_get<r0 + 3>() * inv);
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
                        if (i0 + row < S) *reinterpret_cast<bf16x8*>(orow + (long)row * ors + 64 * half + 8 * ech) = ov[r];
                    }
                });
            });
        }
        PIPE_FENCE();
        if (n_n >= nseq) break;
        c_n = n_n;
        c_bh = n_bh;
        c_qb = n_qb;
        sfor<32 * DB>([&](auto rt) { agpr_zero<Cfg::R_O + decltype(rt)::value>(); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIPE_FENCE();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the re-fetched tail tiles
}

}  // namespace

bool mgx_attn_fwd_pipe_supported(int S, int dh) { return (dh == 64 || dh == 128) && S >= 1; }

int mgx_attn_fwd_pipe(const void* qkv, const float* slopes, void* out, float* lse, int B, int S, int H, int dh,
                      hipStream_t st) {
    const int Sp = (S + 63) / 64 * 64;
    const float c = LOG2E / sqrtf((float)dh);
    FwdArgs a{(const bf16_t*)qkv, slopes, (bf16_t*)out, lse, B, S, Sp, H, c, 0, 0, 0};
    a.nqb = (S + 255) / 256;
    a.np = (a.nqb + 1) / 2;
    a.nwork = B * H * a.np;
    const int n_cu = mgx_cu_count();
    MGX_REQUIRE(n_cu > 0, MGX_ELAUNCH, "attention forward: cannot query the device");
    const int grid = a.nwork < n_cu ? a.nwork : n_cu;     // one persistent workgroup per CU
    if (dh == 64)
        hipLaunchKernelGGL((attn_fwd_pipe_kernel<64>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attn_fwd_pipe_kernel<128>), dim3(grid), dim3(256), 0, st, a);
    MGX_CHECK_LAUNCH("attn_fwd_pipe");
    return MGX_OK;
}
