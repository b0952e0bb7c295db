// bf16 MFMA flash-attention backward for gfx950, software-pipelined inside the wave (causal + ALiBi).
// Autograd of SimpleAttention.multihead_attn (model_baseline/alibi_unified.py:40-54).
//
// dK/dV kernel.  One wave per SIMD (4 waves, 32 keys each; K and V fragments and the 32 x dh dK / dV accumulators
// stay in registers); queries arrive in UNITS of 32 rows (Q and dO, 32 x dh each) through a ring of NB LDS slots
// filled by LDS-DMA.  Per unit u and wave:
//     A(u)  S = Q.K^T, dP = dO.V^T                          2*dh/16 MFMAs
//     B(u)  P = exp2(c S + bias - lse), dS = P*(dP - delta)  VALU (packed fp32), bf16 packs
//     C(u)  dV^T += dO^T.P, dK^T += Q^T.dS                  4*dh/32 MFMAs   (P, dS straight from the accumulators)
// and the loop body issues  X(u) = A(u+1) interleaved with B(u)  then  Y(u) = C(u) interleaved with the LDS reads
// of the next stages, so the MFMA pipe always has independent work while the VALU, the LDS and the DMA engine run
// beside it.  All LDS reads are inline asm with explicit s_waitcnt (hipcc would otherwise drain vmcnt before every
// read that follows an LDS-DMA).
#include <type_traits>
#include <utility>
#include "attn_mfma_common.h"

// No packed-fp32 VALU in this file: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do not run beside the MFMA pipe
// (tools/mfma_overlap_lab.cpp: two of them per MFMA cost as much as ten plain VALU instructions), and hipcc forms
// them from any pair of scalar operations.
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)

namespace {

typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ unsigned lds_addr32(const char* p) { return (unsigned)(unsigned long)(lds_char_t*)p; }

template <int OFF>
__device__ __forceinline__ bf16x8 lds_b128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_f128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

// ---- AGPRs are allocated BY HAND.  hipcc keeps every MFMA accumulator of a 1-wave-per-SIMD kernel in AGPRs and copies
// S / dP to VGPRs and back around each VALU use, and values with an "a" inline-asm constraint are shuttled through
// VGPRs as well; so everything that lives in the AGPR half of the register file (dK / dV accumulators, the K / V
// fragments, the transposed Q / dO fragments) is named literally in the asm text (register numbers are template
// constants printed through "n" operands) and never exists as a compiler value.  The compiler owns the VGPR half
// (a kernel of this shape needs < 256 of them, so it has no reason to spill into AGPRs; the build checks the ISA).
// Its hazard recogniser does not see these MFMAs: the pipeline keeps every VALU producer / consumer of an MFMA
// operand a whole stage away, and explicit s_nop's guard the places where that is not so.
template <int R0, int OFF0, int OFF1>
__device__ __forceinline__ void lds_tr128_agpr(unsigned a0, unsigned a1) {
    static_assert(OFF0 >= 0 && OFF1 < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b64_tr_b16 a[%2:%3], %0 offset:%6\n\tds_read_b64_tr_b16 a[%4:%5], %1 offset:%7"
                 :
                 : "v"(a0), "v"(a1), "n"(R0), "n"(R0 + 1), "n"(R0 + 2), "n"(R0 + 3), "n"(OFF0), "n"(OFF1));
}
// S / dP: D (VGPRs) = A (row fragment, VGPRs) x B (K or V fragment, a[RB..RB+3]) [+ D]
template <int RB>
__device__ __forceinline__ void mfma_s0(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], 0" : "=&v"(d) : "v"(a), "n"(RB), "n"(RB + 3));
}
template <int RB>
__device__ __forceinline__ void mfma_s(f32x16& d, const bf16x8& a) {
#ifndef PIPE_ABL_NOMFMA
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "n"(RB), "n"(RB + 3));
#endif
#ifdef PIPE_ABL_MFMA2
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "n"(RB), "n"(RB + 3));
#endif
}
// dV / dK: a[RD..RD+15] += A (transposed fragment, a[RA..RA+3]) x B (P or dS, VGPRs)
template <int RD, int RA>
__device__ __forceinline__ void mfma_acc(const bf16x8& b) {
#ifndef PIPE_ABL_NOMFMA
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]"
                 :
                 : "v"(b), "n"(RD), "n"(RD + 15), "n"(RA), "n"(RA + 3));
#endif
#ifdef PIPE_ABL_MFMA2
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]"
                 :
                 : "v"(b), "n"(RD), "n"(RD + 15), "n"(RA), "n"(RA + 3));
#endif
}
template <int R>
__device__ __forceinline__ void agpr_zero() {
    asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(R));
}
template <int R>
__device__ __forceinline__ float agpr_get() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}
template <int R0>
__device__ __forceinline__ void global_b128_to_agpr(const void* p) {
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off" ::"v"(p), "n"(R0), "n"(R0 + 3) : "memory");
}

template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(std::make_integer_sequence<int, N>{}, f);
}
#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    union {
        bf16x2 v;
        unsigned u;
    } r;
    r.v = __builtin_convertvector(f2{a, b}, bf16x2);
    return r.u;
}

// nl[b,h,i] = -(slope_h * (i mod 32) + lse) * log2e      (log2 units, relative to the start of the 32-row unit),
//             -inf for S <= i < Sp
// nd[b,h,i] = -sum_d dout*out                               0 for S <= i < Sp
template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const bf16_t* __restrict__ dout,
                                                              const bf16_t* __restrict__ out,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ slopes, float* __restrict__ nl,
                                                              float* __restrict__ nd, long n_chunks, int S, int Sp,
                                                              int H) {
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
        if (i < S) {
            nl[dst] = -(slopes[h] * (float)(i & 31) + lse[(b * H + h) * S + i]) * LOG2E;
            nd[dst] = -s;
        } else {
            nl[dst] = -INFINITY;
            nd[dst] = 0.f;
        }
    }
}

template <int DH>
struct DkvCfg {
    static constexpr int NB = 7;                       // LDS ring slots (one 32-query unit each)
    static constexpr int UB = 32 * DH * 2;             // bytes of one 32 x dh bf16 image
    static constexpr int SLOT = 2 * UB + 1024;         // Q image, dO image, 4 per-wave copies of (nl[32], nd[32])
    static constexpr int VM_PER_UNIT = 2 * (DH / 64) + 1;   // LDS-DMA instructions per wave and unit
    static constexpr int KS = DH / 16, DB = DH / 32;
    static constexpr int NG_X = 2 * KS;                // MFMAs of stage A
    static constexpr int NG_Y = 4 * DB;                // MFMAs of stage C: NG_Y / 2 for dV, then NG_Y / 2 for dK
    static constexpr int HF = NG_Y / 2;
    // AGPR map
    static constexpr int R_DK = 0, R_DV = 16 * DB, R_KF = 32 * DB, R_VF = R_KF + 4 * KS, R_TR = R_VF + 4 * KS;
    static constexpr int R_END = R_TR + 4 * NG_Y;      // 256 at dh 128, 128 at dh 64
    // LDS reads issued in gap g of Y (first half only): two ds_read_b64_tr_b16 + 8 / HF statistics reads
    static constexpr int SPG = 8 / HF;                 // 1 at dh 128, 2 at dh 64
    static constexpr int reads_in_gap(int g) { return g < HF ? 2 + SPG : 0; }
    // reads issued after those of gap f - HF and before MFMA f: what lgkmcnt may still show
    static constexpr int y_wait(int f) {
        int n = 0;
        for (int g = f - HF + 1; g < f; ++g) n += reads_in_gap(g);
        return n < 12 ? n : 12;
    }
};

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_pipe_kernel(
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ qkv, const float* __restrict__ nl,
    const float* __restrict__ nd, const float* __restrict__ slopes, bf16_t* __restrict__ dqkv, int S, int Sp, int H,
    float c, float scale) {
    using Cfg = DkvCfg<DH>;
    constexpr int NB = Cfg::NB, UB = Cfg::UB, SLOT = Cfg::SLOT, G8 = Img<DH>::G8;
    constexpr int KS = Cfg::KS, DB = Cfg::DB, HF = Cfg::HF;
    static_assert(Cfg::R_END <= 256, "AGPR map");
    __shared__ __attribute__((aligned(1024))) char smem[NB * SLOT];
    // the AGPRs named in the asm below, made known to the compiler's resource accounting (it allocates none itself)
    asm volatile("" ::: "a0", "a127", "a255");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h5 = lane >> 5;
    const int jb0 = blockIdx.z * 128, jw0 = jb0 + 32 * w;        // early key blocks (most queries) first
    const int b = blockIdx.y, h = blockIdx.x;
    const unsigned rs = 3u * H * DH;
    const bf16_t* base = qkv + (long)b * S * rs + (long)h * DH;
    const bf16_t* kp = base + (long)H * DH;
    const bf16_t* vp = base + 2L * H * DH;
    const bf16_t* gp = dout + (long)b * S * H * DH + (long)h * DH;
    const float* nl_bh = nl + ((long)b * H + h) * Sp;
    const float* nd_bh = nd + ((long)b * H + h) * Sp;
    const LaneAddr<DH> la(lane);

    const int j = jw0 + l31;
    const bool jok = j < S;
    {
        // K / V fragments of this lane's key straight into AGPRs; keys past the end read row S-1 (finite data, and
        // their columns of dK / dV are never stored)
        const long jr = (long)min(j, S - 1) * rs + 8 * h5;
        sfor<KS>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            global_b128_to_agpr<Cfg::R_KF + 4 * ks>(kp + jr + ks * 16);
            global_b128_to_agpr<Cfg::R_VF + 4 * ks>(vp + jr + ks * 16);
        });
        sfor<32 * DB>([&](auto rt) { agpr_zero<decltype(rt)::value>(); });
    }
    const float csb = slopes[h] * LOG2E;                 // bias per key step in log2 units

    // units [u0, uendp): an even count (u0 is a multiple of 4, Sp of 64); rows S..Sp-1 carry nl = -inf, so the
    // padding units contribute exact zeros
    const int u0 = jb0 / 32;
    const int uendp = Sp / 32;

    // ---- LDS-DMA staging of one unit: wave w brings rows 8w..8w+7 of Q and dO (DH/64 KiB-pieces each) and its
    // own copy of the 64 statistics.  Lane l of a piece lands at byte 16 l: 512-B sub-tile l >> 5, row (l >> 2) & 7,
    // slot l & 3 of the image  off(row, ch) = G8 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)).
    const int rowin = (lane >> 2) & 7;
    const int srow = 8 * w + rowin;                      // row of the unit this lane fetches
    const int sx = (lane & 3) ^ ((srow >> 2) & 3);
    unsigned colb[DH / 64];
#pragma unroll
    for (int hf = 0; hf < DH / 64; ++hf) colb[hf] = (unsigned)((4 * (2 * hf + (lane >> 5)) + sx) * 16);
    auto stage = [&](int v, int slot) {
        const int i0 = min(v, uendp - 1) * 32;
        const int rr = min(i0 + srow, S - 1);            // rows past the end re-read row S-1 (their P is exactly 0)
        const unsigned oq = (unsigned)rr * (rs * 2u);
        const unsigned og = (unsigned)rr * (unsigned)(H * DH * 2);
        char* dst = smem + slot * SLOT + G8 * w;
#pragma unroll
        for (int hf = 0; hf < DH / 64; ++hf) {
            __builtin_amdgcn_global_load_lds((const void*)((const char*)base + (oq + colb[hf])),
                                             (lds_void_t*)(dst + 1024 * hf), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)((const char*)gp + (og + colb[hf])),
                                             (lds_void_t*)(dst + UB + 1024 * hf), 16, 0, 0);
        }
        const float* sp = (h5 ? nd_bh : nl_bh) + i0 + l31;
        __builtin_amdgcn_global_load_lds((const void*)sp, (lds_void_t*)(smem + slot * SLOT + 2 * UB + 256 * w), 4, 0, 0);
    };

    const unsigned lds0 = lds_addr32(smem);
    const unsigned stat_lane = (unsigned)(2 * UB + 256 * w + 16 * h5);
    // values carried between stages (VGPRs)
    bf16x8 rq[KS], rg[KS];        // row fragments (A operand of S / dP) of the unit whose stage A comes next
    f32x4 cnl[4], cnd[4];         // row statistics of the unit whose stage B comes next

    // row fragment g of a unit (g even: Q rows for S, g odd: dO rows for dP; k-step g >> 1)
    auto read_row = [&](auto gt, unsigned a0, unsigned a1) {
        constexpr int g = decltype(gt)::value, ks = g >> 1;
        if constexpr (g & 1)
            rg[ks] = lds_b128<UB + 512 * (ks >> 1)>((ks & 1) ? a1 : a0);
        else
            rq[ks] = lds_b128<512 * (ks >> 1)>((ks & 1) ? a1 : a0);
    };
    auto read_stat = [&](auto rt, unsigned as) {
        constexpr int r = decltype(rt)::value;
        if constexpr (r < 4)
            cnl[r] = lds_f128<32 * r>(as);
        else
            cnd[r - 4] = lds_f128<128 + 32 * (r - 4)>(as);
    };
    // fragment f of stage C -> a[R_TR + 4 f ..]: f < HF: dO^T for dV, f >= HF: Q^T for dK; (s2, d) = ((f % HF) / DB, f % DB)
    auto read_tr = [&](auto ft, unsigned t0, unsigned t1) {
        constexpr int f = decltype(ft)::value;
        constexpr int which = f / HF, d = f % DB, s2 = (f % HF) / DB;
        constexpr int o0 = (which ? 0 : UB) + G8 * (2 * s2) + 512 * d, o1 = o0 + G8;
        lds_tr128_agpr<Cfg::R_TR + 4 * f, o0, o1>(t0, t1);
    };
    // stage A of one unit: S and dP chains, alternating (gap index g: even = S, odd = dP)
    auto mfma_a_step = [&](auto gt, f32x16& sc, f32x16& dp) {
        constexpr int gI = decltype(gt)::value, ks = gI >> 1;
        if constexpr (gI == 0)
            mfma_s0<Cfg::R_KF>(sc, rq[0]);
        else if constexpr (gI == 1)
            mfma_s0<Cfg::R_VF>(dp, rg[0]);
        else if constexpr (gI & 1)
            mfma_s<Cfg::R_VF + 4 * ks>(dp, rg[ks]);
        else
            mfma_s<Cfg::R_KF + 4 * ks>(sc, rq[ks]);
    };

    // ---- prologue: units u0 .. u0+NB-2 into slots 0 .. NB-2
#pragma unroll
    for (int s = 0; s < NB - 1; ++s) stage(u0 + s, s);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::VM_PER_UNIT * (NB - 3)) : "memory");   // K, V, units u0 and u0+1
    PIPE_FENCE();
    __builtin_amdgcn_s_barrier();
    PIPE_FENCE();
    f32x16 scA, dpA, scB, dpB;
    {
        // rows of unit u0 -> A(u0); then statistics of u0 and rows of u0+1
        const unsigned a0 = lds0 + la.row0, a1 = lds0 + la.row1, as = lds0 + stat_lane;
        sfor<Cfg::NG_X>([&](auto gt) { read_row(gt, a0, a1); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PIPE_FENCE();
        sfor<Cfg::NG_X>([&](auto gt) { mfma_a_step(gt, scA, dpA); });
        PIPE_FENCE();
        const unsigned b0 = a0 + SLOT, b1 = a1 + SLOT;
        sfor<8>([&](auto rt) { read_stat(rt, as); });
        sfor<Cfg::NG_X>([&](auto gt) { read_row(gt, b0, b1); });
        PIPE_FENCE();
    }

    int slot_c = 0;                                      // slot of the unit in stage B / C
    int slot_p = NB - 1;                                 // slot of the unit before it (free once every wave is past it)
    float cjc = csb * (float)(j - 32 * u0);              // bias of key j against the first row of the unit (log2 units)
    const int umask = u0 + 4;                            // units [u0, umask) can touch the causal diagonal

    // One unit.  (sc, dp) hold S / dP of unit u (from the previous X), (scn, dpn) receive unit u + 1.
    auto unit = [&](int u, f32x16& sc, f32x16& dp, f32x16& scn, f32x16& dpn) {
        constexpr int NGX = Cfg::NG_X;
        constexpr int EPG = 16 / NGX;                     // elements of P per gap of X (1 at dh 128, 2 at dh 64)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // rows of u+1, statistics of u
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::VM_PER_UNIT * (NB - 4)) : "memory");   // my pieces of unit u+2
        PIPE_FENCE();
#ifndef PIPE_ABL_NOBAR
        __builtin_amdgcn_s_barrier();                             // unit u+2 complete; slot of unit u-1 free in every wave
#endif
        PIPE_FENCE();
#ifndef PIPE_ABL_NODMA
        stage(u + NB - 1, slot_p);
#endif
        const int slot_1 = slot_c + 1 >= NB ? slot_c + 1 - NB : slot_c + 1;
        const int slot_2 = slot_1 + 1 >= NB ? slot_1 + 1 - NB : slot_1 + 1;
        const unsigned a0 = lds0 + slot_2 * SLOT + la.row0, a1 = lds0 + slot_2 * SLOT + la.row1;   // rows of u+2
        const unsigned as = lds0 + slot_1 * SLOT + stat_lane;                                      // statistics of u+1
        const unsigned tc0 = lds0 + slot_c * SLOT + la.tr0, tc1 = lds0 + slot_c * SLOT + la.tr1;   // unit u, transposed
        if (u < umask) {                                  // wave-uniform, at most 4 units per workgroup
            const int di = j - 32 * u - 4 * h5;           // row part < di  <=>  query before key
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMA of A(u) must have landed
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((e & 3) + 8 * (e >> 2) < di) sc[e] = -INFINITY;
        }
        // ---------------- X(u): A(u+1) MFMAs; between them P of unit u (plain fp32 VALU: v_pk_* does not run beside
        // the MFMA pipe), the dO^T fragments of unit u, and the row fragments of unit u+2 into the registers that
        // the MFMA before the gap has just consumed
        PIPE_FENCE();
        float pe[16];
        unsigned pf[8], dsf[8];                           // bf16 pairs: P and dS of unit u
        sfor<NGX>([&](auto gt) {
            constexpr int gI = decltype(gt)::value;
            mfma_a_step(gt, scn, dpn);
            PIPE_FENCE();
#ifndef PIPE_ABL_NOLDS
            if constexpr (gI < HF) read_tr(gt, tc0, tc1);
            read_row(gt, a0, a1);
#endif
#ifdef PIPE_ABL_NOVALU
            if (gI == 0)
                for (int k = 0; k < 8; ++k) pf[k] = dsf[k] = __float_as_uint(sc[k]);
#else
#pragma unroll
            for (int q = 0; q < EPG; ++q) {
                constexpr int e0 = gI * EPG;
                const int e = e0 + q;
                pe[e] = fast_exp2(fmaf(sc[e], c, cnl[e >> 2][e & 3] + cjc));
                if (e & 1) pf[e >> 1] = pack2(pe[e - 1], pe[e]);
            }
#endif
            PIPE_FENCE();
        });
        // ---------------- Y(u): C(u) MFMAs (dV first, then dK); between the dV MFMAs dS of unit u, the Q^T fragments
        // of unit u and the statistics of unit u+1
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NGX - HF) : "memory");   // dO^T fragments (the last row reads may fly)
        PIPE_FENCE();
        union {
            unsigned u[4];
            bf16x8 v;
        } pb[2], db[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int k = 0; k < 4; ++k) pb[s2].u[k] = pf[4 * s2 + k];
        PIPE_FENCE();
        sfor<Cfg::NG_Y>([&](auto gt) {
            constexpr int f = decltype(gt)::value;
            constexpr int which = f / HF, d = f % DB, s2 = (f % HF) / DB;
            if constexpr (f >= HF) {
                if constexpr (f == HF) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int k = 0; k < 4; ++k) db[s].u[k] = dsf[4 * s + k];
                    asm volatile("s_nop 1" ::: "memory");         // VALU write of dS -> MFMA read
                }
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(Cfg::y_wait(f)) : "memory");
                PIPE_FENCE();
            }
            if constexpr (which)
                mfma_acc<Cfg::R_DK + 16 * d, Cfg::R_TR + 4 * f>(db[s2].v);
            else
                mfma_acc<Cfg::R_DV + 16 * d, Cfg::R_TR + 4 * f>(pb[s2].v);
            PIPE_FENCE();
            if constexpr (f < HF) {
#ifndef PIPE_ABL_NOVALU
                // dS pairs: 8 / HF per gap
#pragma unroll
                for (int q = 0; q < 8 / HF; ++q) {
                    constexpr int pr0 = f * (8 / HF);
                    const int pr = pr0 + q, e = 2 * pr;
                    const float d0 = pe[e] * (dp[e] + cnd[e >> 2][e & 3]);
                    const float d1 = pe[e + 1] * (dp[e + 1] + cnd[e >> 2][(e & 3) + 1]);
                    dsf[pr] = pack2(d0, d1);
                }
#endif
#ifndef PIPE_ABL_NOLDS
                read_tr(std::integral_constant<int, f + HF>{}, tc0, tc1);
                sfor<Cfg::SPG>([&](auto st) {
                    // cnl is dead since X; cnd[k] is last used by the dS pairs of gap (2k+1) * HF / 8
                    constexpr int r = f * Cfg::SPG + decltype(st)::value;
                    read_stat(std::integral_constant<int, r>{}, as);
                });
#endif
            }
            PIPE_FENCE();
        });
        slot_p = slot_c;
        slot_c = slot_1;
        cjc -= 32.f * csb;
    };

    for (int u = u0; u < uendp; u += 2) {
        unit(u, scA, dpA, scB, dpB);
        unit(u + 1, scB, dpB, scA, dpA);
    }
    // drain the clamped re-loads and reads of the tail; the last MFMAs must have written dK / dV before they are
    // read back (16 passes; nothing tells the compiler about the dependency)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    PIPE_FENCE();

    if (jok) {
        bf16_t* op = dqkv + ((long)b * S + j) * rs + (long)h * DH;
        sfor<DB>([&](auto dt) {
            constexpr int d = decltype(dt)::value;
            sfor<4>([&](auto gt) {
                constexpr int g = decltype(gt)::value;
                bf16x4 a, v;
                a[0] = (bf16_t)(agpr_get<Cfg::R_DK + 16 * d + 4 * g + 0>() * scale);
                a[1] = (bf16_t)(agpr_get<Cfg::R_DK + 16 * d + 4 * g + 1>() * scale);
                a[2] = (bf16_t)(agpr_get<Cfg::R_DK + 16 * d + 4 * g + 2>() * scale);
                a[3] = (bf16_t)(agpr_get<Cfg::R_DK + 16 * d + 4 * g + 3>() * scale);
                v[0] = (bf16_t)agpr_get<Cfg::R_DV + 16 * d + 4 * g + 0>();
                v[1] = (bf16_t)agpr_get<Cfg::R_DV + 16 * d + 4 * g + 1>();
                v[2] = (bf16_t)agpr_get<Cfg::R_DV + 16 * d + 4 * g + 2>();
                v[3] = (bf16_t)agpr_get<Cfg::R_DV + 16 * d + 4 * g + 3>();
                *reinterpret_cast<bf16x4*>(op + (long)H * DH + 32 * d + 8 * g + 4 * h5) = a;
                *reinterpret_cast<bf16x4*>(op + 2L * H * DH + 32 * d + 8 * g + 4 * h5) = v;
            });
        });
    }
}

}  // namespace

#pragma clang attribute pop

long mgx_attn_bwd_pipe_stats_floats(int B, int S, int H) { return 2L * B * H * ((S + 63) / 64 * 64); }

// statistics pre-pass + dK/dV; `stats` holds mgx_attn_bwd_pipe_stats_floats(B, S, H) floats: nl then nd
int mgx_attn_bwd_dkv_pipe(const void* dout, const void* qkv, const void* out, const float* lse, const float* slopes,
                          void* dqkv, float* stats, int B, int S, int H, int dh, hipStream_t st) {
    const int Sp = (S + 63) / 64 * 64;
    float* nl = stats;
    float* nd = stats + (long)B * H * Sp;
    const float scale = 1.0f / sqrtf((float)dh);
    const float c = LOG2E * scale;
    const long n_chunks = (long)B * Sp * H * (dh / 8);
    dim3 gs((unsigned)((n_chunks + 255) / 256));
    dim3 gk(H, B, (S + 127) / 128);
    if (dh == 64) {
        hipLaunchKernelGGL((attn_bwd_stats_kernel<64>), gs, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                           lse, slopes, nl, nd, n_chunks, S, Sp, H);
        MGX_CHECK_LAUNCH("attn_bwd_stats");
        hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<64>), gk, dim3(256), 0, st, (const bf16_t*)dout,
                           (const bf16_t*)qkv, nl, nd, slopes, (bf16_t*)dqkv, S, Sp, H, c, scale);
    } else {
        hipLaunchKernelGGL((attn_bwd_stats_kernel<128>), gs, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out,
                           lse, slopes, nl, nd, n_chunks, S, Sp, H);
        MGX_CHECK_LAUNCH("attn_bwd_stats");
        hipLaunchKernelGGL((attn_bwd_dkv_pipe_kernel<128>), gk, dim3(256), 0, st, (const bf16_t*)dout,
                           (const bf16_t*)qkv, nl, nd, slopes, (bf16_t*)dqkv, S, Sp, H, c, scale);
    }
    MGX_CHECK_LAUNCH("attn_bwd_dkv_pipe");
    return MGX_OK;
}
