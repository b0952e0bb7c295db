// Head and tail of one sampling step (inference_unified.py:14-20, inference_separated.py:20-27), bf16, for one token
// stream (unified) or several (separated: note | velocity | duration), greedy or stochastic:
//   mgx_sample_gather_embed   seq[:, (k,) pos .. pos+R-1] -> embedding rows (concatenated over streams) -> bf16 [B,R,d]
//                             (seq.index_select + F.embedding (+ cat) + .to(dtype): alibi_unified.py:130-131,
//                             alibi_separated.py:135-139)
//   mgx_sample_head           row R-1 of the trunk output -> Norm (base_layers.py:36-41) -> the stacked output heads
//                             (alibi_unified.py:136-137, alibi_separated.py:151-154) -> per sequence and head the
//                             running argmax (inference_unified.py:18) or a Gumbel-max draw (:16-17)
//   mgx_sample_commit         seq[:, (k,) pos+1] = drawn class, pos += 1 (inference_unified.py:19-20; the reference
//                             draws token i from the logits of row i+1)
// mgx_decode_gather_embed / _head_argmax / _commit are the one-stream greedy forms of the same kernels.
// Three launches instead of the ~14 tiny library kernels (index arithmetic, gather, casts, LayerNorm over all
// rows, a 192 x 618 GEMM, slice, argmax, scatter) that made ~50 us of every 1 ms step.  The position lives in
// device memory, so the step replays as a HIP graph.  Integer results are exact; logits are the bf16-rounded
// fp32-accumulated values a bf16 F.linear produces; ties go to the lowest class index as torch.argmax does.
#include "mgx_common.h"

namespace {

// several token streams (alibi_separated.py:135-139) or one (alibi_unified.py:130): tables and widths by value
struct EmbedCat {
    const float* table[4];
    int V[4], d[4];
    int n;
};

// one workgroup per (b, r) row; seq [B, n, L]: stream k of sequence b is row b*n + k
__global__ __launch_bounds__(256) void gather_embed_kernel(const void* __restrict__ seq, int seq_i64, long L,
                                                            const int* __restrict__ pos, EmbedCat e, int D, int R,
                                                            bf16_t* __restrict__ out) {
    const int b = blockIdx.x / R, r = blockIdx.x % R;
    const long col = min((long)(*pos + r), L - 1);
    bf16_t* dst = out + (long)blockIdx.x * D;
    for (int k = 0; k < e.n; ++k) {
        const long at = ((long)b * e.n + k) * L + col;
        long tok = seq_i64 ? reinterpret_cast<const long*>(seq)[at] : (long)reinterpret_cast<const int*>(seq)[at];
        tok = min(max(tok, 0L), (long)e.V[k] - 1);        // F.embedding would raise; a sampling buffer never holds these
        const float* src = e.table[k] + tok * e.d[k];
        for (int c = threadIdx.x * 4; c < e.d[k]; c += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
            const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(dst + c) = o;
        }
        dst += e.d[k];
    }
}

// training-time embedding rows: out[t, :] = bf16(table[tokens[t], :]) -- F.embedding + .to(bfloat16) in one pass
// (alibi_unified.py:130-131).  One wave per token row, 16-byte loads, 8-byte stores.
__global__ __launch_bounds__(256) void embed_rows_kernel(const long* __restrict__ tokens, const float* __restrict__ table,
                                                          int V, int d, long T, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < T; t += (long)gridDim.x * 4) {
        long tok = tokens[t];
        tok = min(max(tok, 0L), (long)V - 1);
        const float* src = table + tok * d;
        bf16_t* dst = out + t * d;
        for (int c = lane * 4; c < d; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
            const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(dst + c) = o;
        }
    }
}

// several token streams x [B, n, S] -> concat_k bf16(table_k[x[b, k, s], :]) as rows of out [B*S, sum d_k]
// (alibi_separated.py:135-139: three F.embedding gathers + torch.cat + the autocast cast).  One wave per row.
__global__ __launch_bounds__(256) void embed_cat_kernel(const long* __restrict__ tokens, EmbedCat e, int B, int S, int D,
                                                         bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long T = (long)B * S;
    for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < T; t += (long)gridDim.x * 4) {
        const long b = t / S, sidx = t - b * S;
        bf16_t* dst = out + t * D;
        for (int k = 0; k < e.n; ++k) {
            long tok = tokens[(b * e.n + k) * S + sidx];
            tok = min(max(tok, 0L), (long)e.V[k] - 1);
            const float* src = e.table[k] + tok * e.d[k];
            for (int c = lane * 4; c < e.d[k]; c += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
                const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(dst + c) = o;
            }
            dst += e.d[k];
        }
    }
}

__device__ __forceinline__ unsigned long long argmax_key(float v, int idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving map of finite floats to unsigned
    return ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)idx);   // ties: the lower index wins
}

// Output heads stacked in one weight: head k owns classes [off[k], off[k+1]) of W / bias (alibi_separated.py:151-154:
// note | velocity | duration; n = 1 for the unified model).
struct SampleHeads {
    int n;
    int off[5];
};

// Standard Gumbel noise from a counter: (seed, step, element) -> 64-bit mix (splitmix64 finaliser) -> u in (0,1) with
// 32 random bits -> -log(-log u).  argmax_c(logit_c + G_c) is a draw from softmax(logits) (the Gumbel-max identity):
// torch.multinomial(softmax(n[:, i+1]), 1) of inference_unified.py:16-17 without the normalisation, the prefix sum,
// or host-side generator state, so the step still replays as a graph.  The stream differs from torch's (it must: the
// reference's generator is host state), the distribution is the same.
__device__ __forceinline__ float gumbel_noise(unsigned long long seed, unsigned step, unsigned long long elem) {
    unsigned long long x = seed ^ ((unsigned long long)(step + 1u) * 0x9E3779B97F4A7C15ull) ^ (elem * 0xD1B54A32D192ED03ull);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = ((float)(unsigned)(x >> 40) + 0.5f) * (1.0f / 16777216.0f);      // 24 bits: exact in fp32, never 0 or 1
    const unsigned lo = (unsigned)x & 0xFFu;                                          // 8 more bits for the tails
    const float u2 = u + ((float)lo - 127.5f) * (1.0f / 4294967296.0f);
    return -__logf(-__logf(fminf(fmaxf(u2, 1.0e-10f), 0.99999994f)));
}

// grid (VC, B), 4 waves.  Every workgroup normalises row R-1 of sequence b itself (d <= 2048 values: cheaper than
// a launch boundary), then its waves walk the classes of chunk blockIdx.x: four classes per wave per pass, a
// weight row read as two 16-byte loads per lane per 1024 columns.  keys[b, k] (one per head) must be 0 on entry
// (the commit kernel leaves them so).  logits (optional) [B,V] fp32.  stochastic: the keys order logit + Gumbel noise
// keyed by (seed, *pos, b * V + class) instead of the logit.
template <int NV>   // d <= 1024 * NV: every thread normalises 4 * NV columns
__global__ __launch_bounds__(256) void head_argmax_kernel(const bf16_t* __restrict__ h, int R, int d,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           float eps, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, int V, int vper,
                                                           SampleHeads hd, int stochastic, unsigned long long seed,
                                                           const int* __restrict__ pos,
                                                           unsigned long long* __restrict__ keys,
                                                           float* __restrict__ logits) {
    __shared__ __attribute__((aligned(16))) bf16_t y[256 * 4 * NV];
    __shared__ float red[8];
    __shared__ unsigned long long best[4];
    if (threadIdx.x < 4) best[threadIdx.x] = 0;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bf16_t* x = h + ((long)b * R + (R - 1)) * d;
    // LayerNorm as ln_fwd_kernel: fp32 mean, biased variance of the deviations, result rounded to bf16
    float v[4 * NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 256 + tid) * 4;
        if (col < d) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(x + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[4 * c + e] = (float)t[e];
                s += v[4 * c + e];
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[w] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 256 + tid) * 4;
        if (col < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[4 * c + e] - mu;
                q = fmaf(t, t, q);
            }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + w] = q;
    __syncthreads();
    const float rs = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)d + eps);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int col = (c * 256 + tid) * 4;
        if (col < d) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + col);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(be + col);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((v[4 * c + e] - mu) * rs * gg[e] + bb[e]);
            *reinterpret_cast<bf16x4*>(y + col) = o;
        }
    }
    __syncthreads();
    // classes of this chunk
    const int v0 = blockIdx.x * vper, v1 = min(v0 + vper, V);
    const unsigned step = stochastic ? (unsigned)*pos : 0u;
    // four classes per pass (their weight rows' loads all in flight before the first FMA: one class at a time was a
    // chain of ~1 us memory round trips, 28 us for the launch)
    for (int c0 = v0 + 4 * w; c0 < v1; c0 += 16) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int col = lane * 8; col < d; col += 512) {
            bf16x8 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const bf16x8*>(W + (long)min(c0 + u, v1 - 1) * d + col);
            const bf16x8 yy = *reinterpret_cast<const bf16x8*>(y + col);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[u] = fmaf((float)a[u][e], (float)yy[e], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cls = c0 + u;
            const float t = wave_sum(acc[u]);
            if (cls < v1) {
                const float lg = (float)(bf16_t)(t + bias[cls]);
                if (lane == 0) {
                    if (logits) logits[(long)b * V + cls] = lg;
                    const int hk = (cls >= hd.off[1]) + (cls >= hd.off[2]) + (cls >= hd.off[3]);
                    const float val = stochastic ? lg + gumbel_noise(seed, step, (unsigned long long)b * V + cls) : lg;
                    atomicMax(&best[hk], argmax_key(val, cls - hd.off[hk]));
                }
            }
        }
    }
    __syncthreads();
    if (tid < hd.n && best[tid]) atomicMax(keys + (long)b * hd.n + tid, best[tid]);
}

// one workgroup: seq[row, pos+1] = class of keys[row] for the B*n rows (sequence b, stream k) of seq [B, n, L];
// keys[row] = 0; pos += 1 (after every thread has read it)
__global__ __launch_bounds__(256) void commit_kernel(void* __restrict__ seq, int seq_i64, long L, int rows,
                                                      int* __restrict__ pos, unsigned long long* __restrict__ keys) {
    const int p = *pos;
    __syncthreads();
    const long at = min((long)p + 1, L - 1);
    for (int b = threadIdx.x; b < rows; b += 256) {
        const unsigned long long k = keys[b];
        const long tok = (long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        if (seq_i64) reinterpret_cast<long*>(seq)[(long)b * L + at] = tok;
        else reinterpret_cast<int*>(seq)[(long)b * L + at] = (int)tok;
        keys[b] = 0;
    }
    if (threadIdx.x == 0) *pos = p + 1;
}

}  // namespace

extern "C" int mgx_sample_gather_embed(const void* seq, int seq_is_int64, long L, const int* pos_dev, int n,
                                       const float* const* tables, const int* V, const int* d, int B, int R, void* out,
                                       void* stream) {
    MGX_REQUIRE(seq && pos_dev && tables && V && d && out, MGX_EINVAL, "mgx_sample_gather_embed: null pointer");
    MGX_REQUIRE(n >= 1 && n <= 4 && B > 0 && R > 0 && L >= R, MGX_EINVAL, "mgx_sample_gather_embed: n=%d B=%d R=%d L=%ld", n, B, R, L);
    EmbedCat e{};
    e.n = n;
    int D = 0;
    for (int k = 0; k < n; ++k) {
        MGX_REQUIRE(tables[k] && V[k] > 0 && d[k] > 0 && d[k] % 4 == 0, MGX_EUNSUPPORTED,
                    "mgx_sample_gather_embed: table %d: V=%d d=%d (d must be a multiple of 4)", k, V[k], d[k]);
        e.table[k] = tables[k];
        e.V[k] = V[k];
        e.d[k] = d[k];
        D += d[k];
    }
    hipLaunchKernelGGL(gather_embed_kernel, dim3(B * R), dim3(256), 0, (hipStream_t)stream, seq, seq_is_int64, L, pos_dev,
                       e, D, R, (bf16_t*)out);
    MGX_CHECK_LAUNCH("gather_embed");
    return MGX_OK;
}

extern "C" int mgx_decode_gather_embed(const void* seq, int seq_is_int64, long L, const int* pos_dev,
                                       const float* table, int V, int d, int B, int R, void* out, void* stream) {
    return mgx_sample_gather_embed(seq, seq_is_int64, L, pos_dev, 1, &table, &V, &d, B, R, out, stream);
}

extern "C" int mgx_sample_head(const void* h, int B, int R, int d, const float* gamma, const float* beta, float eps,
                               const void* W, const float* bias, int n, const int* V, int stochastic,
                               unsigned long long seed, const int* pos_dev, unsigned long long* keys, float* logits,
                               void* stream) {
    MGX_REQUIRE(h && gamma && beta && W && bias && keys && V, MGX_EINVAL, "mgx_sample_head: null pointer");
    MGX_REQUIRE(n >= 1 && n <= 4 && B > 0 && R > 0, MGX_EINVAL, "mgx_sample_head: n=%d B=%d R=%d", n, B, R);
    MGX_REQUIRE(!stochastic || pos_dev, MGX_EINVAL, "mgx_sample_head: a stochastic draw needs the device-side step counter");
    MGX_REQUIRE(d >= 8 && d % 8 == 0 && d <= 2048, MGX_EUNSUPPORTED, "mgx_sample_head: d=%d (multiple of 8, at most 2048)", d);
    SampleHeads hd{};
    hd.n = n;
    int Vt = 0;
    for (int k = 0; k < n; ++k) {
        MGX_REQUIRE(V[k] > 0, MGX_EINVAL, "mgx_sample_head: head %d has %d classes", k, V[k]);
        hd.off[k] = Vt;
        Vt += V[k];
    }
    for (int k = n; k < 5; ++k) hd.off[k] = Vt;          // heads that do not exist own no class
    const int VC = Vt >= 256 ? 16 : (Vt >= 64 ? 4 : 1);   // class chunks: B x VC workgroups
    const int vper = (Vt + VC - 1) / VC;
    const dim3 grid(VC, B);
    hipStream_t st = (hipStream_t)stream;
    if (d <= 1024)
        hipLaunchKernelGGL((head_argmax_kernel<1>), grid, dim3(256), 0, st, (const bf16_t*)h, R, d, gamma, beta, eps,
                           (const bf16_t*)W, bias, Vt, vper, hd, stochastic, seed, pos_dev, keys, logits);
    else
        hipLaunchKernelGGL((head_argmax_kernel<2>), grid, dim3(256), 0, st, (const bf16_t*)h, R, d, gamma, beta, eps,
                           (const bf16_t*)W, bias, Vt, vper, hd, stochastic, seed, pos_dev, keys, logits);
    MGX_CHECK_LAUNCH("sample_head");
    return MGX_OK;
}

extern "C" int mgx_decode_head_argmax(const void* h, int B, int R, int d, const float* gamma, const float* beta, float eps,
                                      const void* W, const float* bias, int V, unsigned long long* keys, float* logits,
                                      void* stream) {
    return mgx_sample_head(h, B, R, d, gamma, beta, eps, W, bias, 1, &V, 0, 0ull, nullptr, keys, logits, stream);
}

extern "C" int mgx_sample_commit(void* seq, int seq_is_int64, long L, int B, int n, int* pos_dev, unsigned long long* keys,
                                 void* stream) {
    MGX_REQUIRE(seq && pos_dev && keys, MGX_EINVAL, "mgx_sample_commit: null pointer");
    MGX_REQUIRE(B > 0 && n >= 1 && n <= 4 && L >= 2, MGX_EINVAL, "mgx_sample_commit: B=%d n=%d L=%ld", B, n, L);
    hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seq, seq_is_int64, L, B * n, pos_dev, keys);
    MGX_CHECK_LAUNCH("commit");
    return MGX_OK;
}

extern "C" int mgx_decode_commit(void* seq, int seq_is_int64, long L, int B, int* pos_dev, unsigned long long* keys,
                                 void* stream) {
    return mgx_sample_commit(seq, seq_is_int64, L, B, 1, pos_dev, keys, stream);
}

extern "C" int mgx_embed_rows_bf16(const long long* tokens, const float* table, int V, int d, long T, void* out,
                                   void* stream) {
    MGX_REQUIRE(tokens && table && out, MGX_EINVAL, "mgx_embed_rows_bf16: null pointer");
    MGX_REQUIRE(T > 0 && V > 0 && d > 0 && d % 4 == 0, MGX_EUNSUPPORTED, "mgx_embed_rows_bf16: T=%ld V=%d d=%d (d must be a multiple of 4)", T, V, d);
    long blocks = (T + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const long*)tokens,
                       table, V, d, T, (bf16_t*)out);
    MGX_CHECK_LAUNCH("embed_rows");
    return MGX_OK;
}

extern "C" int mgx_embed_cat_bf16(const long long* tokens, int n, int B, int S, const float* const* tables, const int* V,
                                  const int* d, void* out, void* stream) {
    MGX_REQUIRE(tokens && tables && V && d && out, MGX_EINVAL, "mgx_embed_cat_bf16: null pointer");
    MGX_REQUIRE(n >= 1 && n <= 4 && B > 0 && S > 0, MGX_EINVAL, "mgx_embed_cat_bf16: n=%d B=%d S=%d", n, B, S);
    EmbedCat e{};
    e.n = n;
    int D = 0;
    for (int k = 0; k < n; ++k) {
        MGX_REQUIRE(tables[k] && V[k] > 0 && d[k] > 0 && d[k] % 4 == 0, MGX_EUNSUPPORTED,
                    "mgx_embed_cat_bf16: table %d: V=%d d=%d (d must be a multiple of 4)", k, V[k], d[k]);
        e.table[k] = tables[k];
        e.V[k] = V[k];
        e.d[k] = d[k];
        D += d[k];
    }
    const long T = (long)B * S;
    long blocks = (T + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(embed_cat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(tokens), e, B, S, D, (bf16_t*)out);
    MGX_CHECK_LAUNCH("embed_cat");
    return MGX_OK;
}
