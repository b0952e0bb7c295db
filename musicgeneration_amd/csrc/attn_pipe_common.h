// Building blocks of the software-pipelined attention kernels (attn_bwd_pipe.hip, attn_bwd_dq_pipe.hip): inline-asm
// LDS reads and MFMAs with hand-allocated AGPRs, compile-time loops.
#pragma once
#include <type_traits>
#include <utility>
#include "attn_mfma_common.h"

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

__device__ __forceinline__ void lds_write_b64(unsigned addr, const bf16x4& v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// ---- AGPRs are allocated BY HAND.  hipcc keeps every MFMA accumulator of a 1-wave-per-SIMD kernel in AGPRs and copies
// S / dP to VGPRs and back around each VALU use, and values with an "a" inline-asm constraint are shuttled through
// VGPRs as well; so everything that lives in the AGPR half of the register file (dK / dV accumulators, the K / V
// fragments, the transposed Q / dO fragments) is named literally in the asm text (register numbers are template
// constants printed through "n" operands) and never exists as a compiler value.  The compiler owns the VGPR half
// (a kernel of this shape needs < 256 of them, so it has no reason to spill into AGPRs; check the ISA after a change:
// no v_accvgpr_* outside the item prologue / epilogue).
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
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%2:%3], %0" : "+v"(d) : "v"(a), "n"(RB), "n"(RB + 3));
}
// the statistics step: B (ones in k = 0..2) in VGPRs
__device__ __forceinline__ void mfma_sv(f32x16& d, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// dV / dK: a[RD..RD+15] += A (transposed fragment, a[RA..RA+3]) x B (P or dS, VGPRs)
template <int RD, int RA>
__device__ __forceinline__ void mfma_acc(const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]"
                 :
                 : "v"(b), "n"(RD), "n"(RD + 15), "n"(RA), "n"(RA + 3));
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
template <int R>
__device__ __forceinline__ void agpr_put(float v) {
    asm volatile("v_accvgpr_write_b32 a[%0], %1" ::"n"(R), "v"(v));
}
template <int R0>
__device__ __forceinline__ void global_b128_to_agpr(const void* p) {
    asm volatile("global_load_dwordx4 a[%1:%2], %0, off" ::"v"(p), "n"(R0), "n"(R0 + 3) : "memory");
}
// fp32 VALU the compiler may not pair into v_pk_fma_f32 / v_pk_mul_f32 (those do not run beside the MFMA pipe)
__device__ __forceinline__ float fma_np(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float add_np(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
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
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf16x2));
}
// four packed pairs -> one MFMA operand.  (No union type punning here: read through the other member, hipcc's
// alias analysis let the MFMA asm see the operand before the words had been written.)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ bf16x8 words_to_frag(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    return __builtin_bit_cast(bf16x8, u32x4{w0, w1, w2, w3});
}

// Both halves of a freshly packed MFMA operand pinned in registers HERE, followed by the wait states a VALU write
// needs before an MFMA reads the register.  Without the "+v" operands hipcc sinks the v_cvt_pk of the second half to
// right before its first use -- an inline-asm MFMA the hazard recogniser does not see, so it read the old contents.
__device__ __forceinline__ void pin_frags(bf16x8& f0, bf16x8& f1) {
    asm volatile("s_nop 1" : "+v"(f0), "+v"(f1)::"memory");
}

}  // namespace
