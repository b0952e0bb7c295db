// C-ABI entry points that dispatch between kernel families, plus error state.
#include "mgx_common.h"

static thread_local char g_err[512] = "";

void mgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Per-device cache of the CU count: immutable once written (every writer stores the same value), so concurrent first
// calls are benign.
int mgx_cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    if (dev >= 0 && dev < 64) cache[dev] = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}

extern "C" int mgx_version(void) { return 100; }
extern "C" const char* mgx_last_error_string(void) { return g_err; }

// attn_generic.hip
int mgx_check_attn_shape(const char* fn, int B, int S, int H, int dh, int dtype);
int mgx_attn_fwd_generic(int mode, const void* qkv, const float* slopes, const void* E, void* out, float* lse, int B,
                         int S, int H, int dh, int max_seq, int dtype, hipStream_t st);
int mgx_attn_bwd_generic(int mode, const void* dout, const void* qkv, const void* out, const float* lse,
                         const float* slopes, const void* E, void* dqkv, float* dE, float* delta, int B, int S, int H,
                         int dh, int max_seq, int dtype, hipStream_t st);
// attn_mfma.hip
bool mgx_attn_mfma_supported(int mode, int dh, int dtype, int S, int H, int max_seq, bool backward);
int mgx_attn_fwd_mfma(int mode, const void* qkv, const float* slopes, const void* E, void* out, float* lse, int B,
                      int S, int H, int dh, int max_seq, hipStream_t st);
int mgx_attn_bwd_mfma(int mode, const void* dout, const void* qkv, const void* out, const float* lse,
                      const float* slopes, const void* E, void* dqkv, float* dE, float* delta, int B, int S, int H,
                      int dh, int max_seq, hipStream_t st);

// dtype | MGX_GENERIC_KERNELS: per-call request for the generic (non-MFMA) kernels (tests, diagnostics)
static inline bool take_generic_flag(int& dtype) {
    const bool g = (dtype & MGX_GENERIC_KERNELS) != 0;
    dtype &= ~MGX_GENERIC_KERNELS;
    return g;
}

extern "C" int mgx_alibi_attn_fwd(const void* qkv, const float* slopes, void* out, float* lse, int B, int S, int H,
                                  int dh, int dtype, void* stream) {
    const bool g_force_generic = take_generic_flag(dtype);
    MGX_REQUIRE(qkv && slopes && out && lse, MGX_EINVAL, "mgx_alibi_attn_fwd: null pointer");
    int rc = mgx_check_attn_shape("mgx_alibi_attn_fwd", B, S, H, dh, dtype);
    if (rc) return rc;
    if (!g_force_generic && mgx_attn_mfma_supported(0, dh, dtype, S, H, 0, false))
        return mgx_attn_fwd_mfma(0, qkv, slopes, nullptr, out, lse, B, S, H, dh, 0, (hipStream_t)stream);
    return mgx_attn_fwd_generic(0, qkv, slopes, nullptr, out, lse, B, S, H, dh, 0, dtype, (hipStream_t)stream);
}

long mgx_attn_bwd_pipe_stats_floats(int B, int S, int H);     // attn_bwd_pipe.hip
extern "C" long mgx_alibi_attn_bwd_workspace_floats(int B, int S, int H) {
    return mgx_attn_bwd_pipe_stats_floats(B, S, H);
}

extern "C" int mgx_alibi_attn_bwd(const void* dout, const void* qkv, const void* out, const float* lse,
                                  const float* slopes, void* dqkv, float* delta, int B, int S, int H, int dh,
                                  int dtype, void* stream) {
    const bool g_force_generic = take_generic_flag(dtype);
    MGX_REQUIRE(dout && qkv && out && lse && slopes && dqkv && delta, MGX_EINVAL, "mgx_alibi_attn_bwd: null pointer");
    int rc = mgx_check_attn_shape("mgx_alibi_attn_bwd", B, S, H, dh, dtype);
    if (rc) return rc;
    if (!g_force_generic && mgx_attn_mfma_supported(0, dh, dtype, S, H, 0, true))
        return mgx_attn_bwd_mfma(0, dout, qkv, out, lse, slopes, nullptr, dqkv, nullptr, delta, B, S, H, dh, 0,
                                 (hipStream_t)stream);
    return mgx_attn_bwd_generic(0, dout, qkv, out, lse, slopes, nullptr, dqkv, nullptr, delta, B, S, H, dh, 0, dtype,
                                (hipStream_t)stream);
}

extern "C" int mgx_rel_attn_fwd(const void* qkv, const void* E, void* out, float* lse, int B, int S, int H, int dh,
                                int max_seq, int dtype, void* stream) {
    const bool g_force_generic = take_generic_flag(dtype);
    MGX_REQUIRE(qkv && E && out && lse, MGX_EINVAL, "mgx_rel_attn_fwd: null pointer");
    int rc = mgx_check_attn_shape("mgx_rel_attn_fwd", B, S, H, dh, dtype);
    if (rc) return rc;
    MGX_REQUIRE(S <= max_seq, MGX_EINVAL, "mgx_rel_attn_fwd: S=%d exceeds max_seq=%d", S, max_seq);
    if (!g_force_generic && mgx_attn_mfma_supported(1, dh, dtype, S, H, max_seq, false))
        return mgx_attn_fwd_mfma(1, qkv, nullptr, E, out, lse, B, S, H, dh, max_seq, (hipStream_t)stream);
    return mgx_attn_fwd_generic(1, qkv, nullptr, E, out, lse, B, S, H, dh, max_seq, dtype, (hipStream_t)stream);
}

long mgx_rel_bwd_mfma_workspace_floats(int B, int S, int H, int dh);     // attn_rel_mfma.hip
extern "C" long mgx_rel_attn_bwd_workspace_floats(int B, int S, int H, int dh, int max_seq, int dtype) {
    const bool generic = take_generic_flag(dtype);
    if (!generic && mgx_attn_mfma_supported(1, dh, dtype, S, H, max_seq, true)) return mgx_rel_bwd_mfma_workspace_floats(B, S, H, dh);
    return (long)B * H * S;
}

extern "C" int mgx_rel_attn_bwd(const void* dout, const void* qkv, const void* out, const float* lse, const void* E,
                                void* dqkv, float* dE, float* workspace, int B, int S, int H, int dh, int max_seq,
                                int dtype, void* stream) {
    float* const delta = workspace;
    const bool g_force_generic = take_generic_flag(dtype);
    MGX_REQUIRE(dout && qkv && out && lse && E && dqkv && dE && delta, MGX_EINVAL, "mgx_rel_attn_bwd: null pointer");
    int rc = mgx_check_attn_shape("mgx_rel_attn_bwd", B, S, H, dh, dtype);
    if (rc) return rc;
    MGX_REQUIRE(S <= max_seq, MGX_EINVAL, "mgx_rel_attn_bwd: S=%d exceeds max_seq=%d", S, max_seq);
    MGX_REQUIRE(max_seq <= 65535, MGX_EUNSUPPORTED, "mgx_rel_attn_bwd: max_seq too large");
    if (!g_force_generic && mgx_attn_mfma_supported(1, dh, dtype, S, H, max_seq, true))
        return mgx_attn_bwd_mfma(1, dout, qkv, out, lse, nullptr, E, dqkv, dE, delta, B, S, H, dh, max_seq,
                                 (hipStream_t)stream);
    return mgx_attn_bwd_generic(1, dout, qkv, out, lse, nullptr, E, dqkv, dE, delta, B, S, H, dh, max_seq, dtype,
                                (hipStream_t)stream);
}
