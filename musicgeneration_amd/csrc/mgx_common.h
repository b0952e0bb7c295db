// Shared device/host helpers for libmgx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mgx.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

void mgx_set_error(const char* fmt, ...);
// compute units of the CURRENT device (looked up once per device, api.hip); <= 0 on failure
int mgx_cu_count();

#define MGX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            mgx_set_error(__VA_ARGS__);         \
            return (code);                      \
        }                                       \
    } while (0)

#define MGX_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            mgx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return MGX_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout: one 32-bit hash serves two elements (16 random bits each).
// keep(element idx) <=> bits16(hash(seed, idx >> 1), idx & 1) >= thr16, thr16 = round(p * 65536);
// the kept fraction is exactly 1 - thr16/65536 and kernels scale kept values by its inverse.
__device__ __forceinline__ unsigned mgx_hash32(unsigned long long seed, unsigned long long pair_idx) {
    unsigned x = (unsigned)pair_idx ^ (unsigned)seed;
    x += (unsigned)(pair_idx >> 32) * 0x9E3779B9u + (unsigned)(seed >> 32);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned mgx_drop_thr16(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ float mgx_keep_scale(unsigned thr16) { return 65536.0f / (float)(65536u - thr16); }
// element idx must be even for the _pair form; returns keep flags for idx and idx+1
__device__ __forceinline__ void mgx_keep_pair(unsigned long long seed, unsigned long long idx, unsigned thr16,
                                              bool& k0, bool& k1) {
    const unsigned h = mgx_hash32(seed, idx >> 1);
    k0 = (h & 0xFFFFu) >= thr16;
    k1 = (h >> 16) >= thr16;
}

// dG workspace of the skew relative attention backward (attn_rel_mfma.hip writes it, gemm_tn.hip MODE 1 reads it): 2-KiB
// blocks (32 queries x 32 distances, bf16).  32-query unit qu holds the distance blocks 0 .. qu | 7 - the causal ones plus
// the zero blocks that complete its 256-distance tile - so the rows form a staircase; mgx_rel_dg_row(qu) is the index of
// unit qu's first block and mgx_rel_dg_row(nrb) the block count of one (batch, head): 8704 instead of 16384 at S = 4096.
__host__ __device__ __forceinline__ long mgx_rel_dg_row(int qu) {
    const long a = qu >> 3, r = qu & 7;
    return 8 * (a + 1) * (4 * a + r);
}
