/*
 * mgx.h — C ABI of libmgx.so, the MI355X (gfx950) kernels behind the
 * MusicGeneration hot path.
 *
 * The reference (Aalanli/MusicGeneration) is pure Python on PyTorch: it has no
 * FFI of its own.  The drop-in boundary is therefore the Python module API
 * (musicgeneration_amd/ mirrors model_baseline/alibi_unified.py etc.), and this
 * header is the layer directly under it: one entry point per fused operator,
 * each citing the reference lines whose ATen op sequence it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's
 *     data_ptr()); nothing is allocated, freed or retained by the library
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     the call returns without synchronising
 *   - `dtype` selects the storage type of activations: MGX_F32 or MGX_BF16;
 *     statistics (lse, mean, rstd, losses) are always fp32
 *   - return value: 0 on success, a negative MGX_E* code otherwise (never throws);
 *     mgx_last_error_string() describes the last failure on the calling thread
 *   - re-entrant and thread-safe.  The library's only state beyond the per-thread error string: a per-device cache
 *     of the compute-unit count (immutable once looked up) and, for mgx_colsum_bf16_static alone, that entry's
 *     device-resident scratch array with its per-device address table (see its comment: overlapping calls race)
 *   - tensors are dense row-major with the shapes given per function
 */
#ifndef MGX_H
#define MGX_H

#ifdef __cplusplus
extern "C" {
#endif

enum { MGX_F32 = 0, MGX_BF16 = 1 };
/* OR-ed into the `dtype` argument of the attention entry points: run the generic wave-per-row kernels even where the
 * MFMA kernels apply (tests and diagnostics; a per-call request, the library keeps no mutable state). */
enum { MGX_GENERIC_KERNELS = 0x100 };
enum { MGX_ACT_GELU = 0, MGX_ACT_RELU = 1 };
enum { MGX_OK = 0, MGX_EINVAL = -1, MGX_EUNSUPPORTED = -2, MGX_ELAUNCH = -3 };

int mgx_version(void);
const char* mgx_last_error_string(void);

/* ---- ALiBi causal attention -------------------------------------------------
 * Replaces SimpleAttention.multihead_attn + split_heads/combine_heads
 * (model_baseline/alibi_unified.py:40-54,59-65; base_layers.py:69-79) and the
 * materialised [H,S,S] mask of construct_alibi (alibi_unified.py:26-31).
 *   qkv    [B,S,3,H,dh]  output of c_attn, read in place (q | k | v thirds of a row)
 *   slopes [H] fp32      get_slopes(H) (alibi_unified.py:13-23)
 *   out    [B,S,H,dh]    == combine_heads(softmax(q k^T/sqrt(dh) + slope_h*j + causal) v)
 *   lse    [B,H,S] fp32  natural-log row normaliser of the shifted scores
 *                        q.k/sqrt(dh) + slope_h*(j-i), kept for the backward
 * The bias is evaluated as slope*(j-i): per row it differs from the reference's
 * slope*j by a constant, which softmax cancels.
 * bf16 with dh in {64,128} runs the MFMA flash kernel; every other case runs
 * the generic wave-per-row kernel (fp32 math).
 */
int mgx_alibi_attn_fwd(const void* qkv, const float* slopes, void* out, float* lse,
                       int B, int S, int H, int dh, int dtype, void* stream);

/* Backward of the above.  dqkv [B,S,3,H,dh] receives dq | dk | dv (fully
 * overwritten).  workspace: caller-provided fp32 scratch of
 * mgx_alibi_attn_bwd_workspace_floats(B, S, H) elements (row statistics: rowsum(dout*out) and the
 * shifted normaliser, rows padded to a multiple of 64). */
long mgx_alibi_attn_bwd_workspace_floats(int B, int S, int H);
int mgx_alibi_attn_bwd(const void* dout, const void* qkv, const void* out, const float* lse,
                       const float* slopes, void* dqkv, float* workspace,
                       int B, int S, int H, int dh, int dtype, void* stream);

/* ---- skew relative-position causal attention --------------------------------
 * Replaces RelativeAttention.relative_attn + multihead_attn
 * (old_models/relative_transformer.py:18-44): softmax((q k^T + rel)*dh^-1/2 + causal) v
 * with rel[i,j] = q_i . E[h, max_seq-1-(i-j)]  (the pad/reshape skew, never materialised).
 *   E [H,max_seq,dh] (same dtype as qkv); requires S <= max_seq.
 */
int mgx_rel_attn_fwd(const void* qkv, const void* E, void* out, float* lse,
                     int B, int S, int H, int dh, int max_seq, int dtype, void* stream);

/* Autograd of the above.  dE [H,max_seq,dh] fp32, fully overwritten (rows < max_seq-S get zero).
 * workspace: mgx_rel_attn_bwd_workspace_floats(...) floats for the same shape and dtype flags: rowsum(dout*out)
 * and, on the bf16 MFMA path, the dG blocks the dQ kernel stores (2 KiB per 32x32 block of the causal triangle, the rows of
 * blocks padded to their 256-distance tile: about B*H*ceil(S/32)^2/2 x 2 KiB) and the per-(b,h) fp32
 * slabs of the dE GEMM over them. */
long mgx_rel_attn_bwd_workspace_floats(int B, int S, int H, int dh, int max_seq, int dtype);
int mgx_rel_attn_bwd(const void* dout, const void* qkv, const void* out, const float* lse,
                     const void* E, void* dqkv, float* dE, float* workspace,
                     int B, int S, int H, int dh, int max_seq, int dtype, void* stream);

/* ---- (dropout +) residual + LayerNorm -----------------------------------------
 * Replaces `x = norm(drop(x1) + x)` (alibi_unified.py:88-89, 93-94) and, with
 * res == NULL and p_drop == 0, the final Norm (base_layers.py:36-41).
 *   a [T,d] branch output, res [T,d] or NULL, gamma/beta [d] fp32
 *   z [T,d] (dtype) pre-norm sum kept for the backward (may be NULL for inference)
 *   y [T,d], mean/rstd [T] fp32 (may be NULL when z is NULL)
 * Dropout keeps an element when hash(seed, element index) >= p_drop and scales by 1/(1-p).
 */
int mgx_ln_residual_fwd(const void* a, const void* res, const float* gamma, const float* beta,
                        void* y, void* z, float* mean, float* rstd,
                        long T, int d, float eps, float p_drop, unsigned long long seed,
                        int dtype, void* stream);

/* dy2 may be NULL; otherwise the incoming gradient is dy + dy2 (fuses the
 * residual-branch add of the backward).  dz [T,d] is the gradient wrt the
 * pre-norm sum (= gradient of `res`); da [T,d] (may alias dz when p_drop==0, or
 * be NULL) is dz with the dropout mask re-applied.  dgamma/dbeta [d] fp32 are
 * overwritten.  dsum [d] fp32 (may be NULL) receives the column sums of da (of dz when there is
 * no dropout): the bias gradient of the projection that produced the branch (c_proj.b / linear2.bias,
 * alibi_unified.py:88,92), so no separate reduction pass over da is needed.
 * partial [nblk_max*3*d] fp32 scratch, nblk_max = mgx_ln_bwd_partial_rows(). */
int mgx_ln_bwd_partial_rows(void);
int mgx_ln_residual_bwd(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                        const float* gamma, void* dz, void* da, float* dgamma, float* dbeta, float* dsum,
                        float* partial, long T, int d, float p_drop, unsigned long long seed, int dtype, void* stream);

/* ---- bias + activation (+ dropout) ------------------------------------------------
 * Replaces `drop(activation(linear1(x)))`'s elementwise tail (alibi_unified.py:90-91):
 *   u [T,n] GEMM output WITHOUT bias; bias [n] fp32
 *   pre [T,n] = u + bias (kept for the backward; may alias u), h [T,n] = drop(act(pre))
 *   bias == NULL and pre == NULL: u already is the pre-activation (a GEMM epilogue added the bias); only h is written
 * act: MGX_ACT_GELU is the exact erf form (F.gelu default), MGX_ACT_RELU for the relative model.
 */
int mgx_bias_act_fwd(const void* u, const float* bias, void* pre, void* h,
                     long T, int n, int act, float p_drop, unsigned long long seed, int dtype, void* stream);
/* dpre [T,n] = dh * mask/(1-p) * act'(pre);  dbias [n] fp32 overwritten;
 * partial scratch as for LayerNorm ([mgx_ln_bwd_partial_rows()*n] fp32). */
int mgx_bias_act_bwd(const void* dh, const void* pre, void* dpre, float* dbias, float* partial,
                     long T, int n, int act, float p_drop, unsigned long long seed, int dtype, void* stream);

/* ---- softmax cross-entropy over the MIDI vocabulary --------------------------------
 * Replaces F.cross_entropy(logits.transpose(-1,-2), y, weight, ignore_index)
 * (alibi_unified.py:161; alibi_separated.py:193-195).
 *   logits [T,ld] with the V class scores of this head at columns [col0, col0+V)
 *   targets [T] int64; weight [V] fp32 or NULL; ignore_index < 0 disables it
 *   row_loss [T] fp32 = w[y]*nll (0 for ignored rows); row_lse [T] fp32
 *   sums fp32 [mgx_xent_sums_len()]: sums[0] = sum w[y]*nll, sums[1] = sum w[y] over
 *   non-ignored rows (overwritten; the tail is scratch for the deterministic
 *   two-stage reduction); the mean loss is sums[0]/sums[1]
 */
int mgx_xent_sums_len(void);
int mgx_softmax_xent_fwd(const void* logits, const long long* targets, const float* weight,
                         float* row_loss, float* row_lse, float* sums,
                         long T, int V, int ld, int col0, long long ignore_index, int dtype, void* stream);
/* dlogits [T,ld] columns [col0,col0+V) = gscale * w[y]/sums[1] * (softmax - onehot);
 * gscale points at one fp32 on the device (upstream grad * loss weight). */
int mgx_softmax_xent_bwd(const void* logits, const long long* targets, const float* weight,
                         const float* row_lse, const float* sums, const float* gscale, void* dlogits,
                         long T, int V, int ld, int col0, long long ignore_index, int dtype, void* stream);

/* ---- KV-cached decode attention ------------------------------------------------------
 * Replaces the full re-forward per generated token of inference_unified.py:12-21
 * (O(L^2) attention per step) by attention of R new query rows against a cache.
 *   q [B,R,H,dh]; kcache/vcache [B,H,Lmax,dh]; row r of q sits at absolute position
 *   pos0 + r and attends keys 0..pos0+r (the cache must already hold them);
 *   out [B,R,H,dh], R <= 4.  Wavefront-reduction softmax; HBM-bound.  bf16 with dh in {64,128}
 *   runs the streaming kernel (dot2 + DPP reductions, prefetched chunks); other cases the generic one.
 */
int mgx_decode_attn(const void* q, const void* kcache, const void* vcache, const float* slopes,
                    void* out, int B, int R, int H, int dh, int Lmax, int pos0, int dtype, void* stream);
/* Scatter R new k/v rows of qkv [B,R,3,H,dh] into the caches at positions pos0.. */
int mgx_kv_append(const void* qkv, void* kcache, void* vcache,
                  int B, int R, int H, int dh, int Lmax, int pos0, int dtype, void* stream);

/* Same two operations with the position read from device memory at execution time
 * (*pos0_dev, one int32), so that a sampling step can be recorded once in a HIP graph and
 * replayed while the position advances (inference_unified.py:14's loop variable).  A position
 * with pos0 + R > Lmax cannot be rejected on the host: the kernels clamp it to Lmax - R. */
int mgx_decode_attn_at(const void* q, const void* kcache, const void* vcache, const float* slopes,
                       void* out, int B, int R, int H, int dh, int Lmax, const int* pos0_dev, int dtype, void* stream);
int mgx_kv_append_at(const void* qkv, void* kcache, void* vcache,
                     int B, int R, int H, int dh, int Lmax, const int* pos0_dev, int dtype, void* stream);

/* One call per layer and sampling step: append the R new K/V rows of qkv [B,R,3,H,dh] to the caches at
 * pos0.. AND attend the R query rows of the same qkv against keys 0..pos0+r (mgx_kv_append +
 * mgx_decode_attn without the intermediate q copy).  pos0_dev != NULL: the position is read from
 * device memory (HIP-graph replay), pos0 is ignored.  bf16 with dh in {64,128} only
 * (MGX_EUNSUPPORTED otherwise: use the two calls above). */
int mgx_decode_step_attn(const void* qkv, void* kcache, void* vcache, const float* slopes, void* out,
                         int B, int R, int H, int dh, int Lmax, int pos0, const int* pos0_dev, int dtype, void* stream);

/* ---- fused AdamW ---------------------------------------------------------------------
 * Replaces torch.optim.AdamW.step (build_everything_unified.py:42) for one flat
 * chunk: decoupled weight decay, bias-corrected moments; also refreshes the bf16
 * shadow copy of the weights when shadow != NULL.  grad may be fp32 or bf16.
 */
int mgx_adamw(float* param, const void* grad, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
              long n, float lr, float beta1, float beta2, float eps, float weight_decay,
              int step, float grad_scale, int grad_dtype, void* stream);

/* Multi-tensor form of mgx_adamw (fp32 gradients, no shadow): `count` tensors, HOST arrays of device
 * pointers and element counts; one launch per 32 tensors.  Same arithmetic, element by element. */
int mgx_adamw_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const long* sizes, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* stream);

/* ---- head and tail of one greedy sampling step (unified model, bf16) ---------------------------
 * Replace, around the decoder blocks of one step of inference_unified.py:14-20:
 *   mgx_decode_gather_embed : seq[:, pos .. pos+R-1] (int32 or int64 rows of length L) -> embedding rows
 *                             (fp32 table [V,d]) -> out bf16 [B,R,d]   (alibi_unified.py:130-131)
 *   mgx_decode_head_argmax  : row R-1 of h [B,R,d] bf16 -> Norm (base_layers.py:36-41) -> proj_class W [V,d] bf16,
 *                             bias fp32 (alibi_unified.py:136-137) -> per-sequence running argmax in keys[B]
 *                             (uint64, must be 0 on entry; ties -> lowest class, as torch.argmax); logits
 *                             (optional, may be NULL) [B,V] fp32 receives the bf16-rounded logits
 *   mgx_decode_commit       : seq[:, pos+1] = argmax class; keys = 0; pos += 1   (inference_unified.py:18-20:
 *                             token i is drawn from the logits of row i+1)
 * pos_dev: one int32 in device memory (position of row 0), so a step replays as a HIP graph. */
int mgx_decode_gather_embed(const void* seq, int seq_is_int64, long L, const int* pos_dev, const float* table,
                            int V, int d, int B, int R, void* out, void* stream);
int mgx_decode_head_argmax(const void* h, int B, int R, int d, const float* gamma, const float* beta, float eps,
                           const void* W, const float* bias, int V, unsigned long long* keys, float* logits,
                           void* stream);
int mgx_decode_commit(void* seq, int seq_is_int64, long L, int B, int* pos_dev, unsigned long long* keys,
                      void* stream);

/* The same three steps for n token streams (n <= 4; n = 3: the separated model's note | velocity | duration rows,
 * inference_separated.py:18-28; n = 1: the unified model) and for the stochastic branch (inference_unified.py:16-17:
 * `multinomial(softmax(n[:, i+1]), 1)`):
 *   seq [B,n,L]; tables[k] fp32 [V[k], d[k]] (HOST arrays of device pointers / ints), out bf16 [B,R,sum d[k]]
 *                (alibi_separated.py:135-139: three gathers + cat);
 *   W [sum V[k], d] bf16 = the output heads stacked (alibi_separated.py:151-154), bias [sum V[k]] fp32, keys [B,n];
 *   stochastic != 0: head k's class is argmax_c(logit_c + G_c), G_c standard Gumbel noise from a counter-based hash of
 *                (seed, *pos_dev, b, c): a draw from softmax(logits) (Gumbel-max), reproducible per seed, no host
 *                generator state, so the step still replays as a HIP graph; not torch.multinomial's stream.
 *   commit: seq[b,k,pos+1] = class of keys[b,k]; keys = 0; pos += 1. */
int mgx_sample_gather_embed(const void* seq, int seq_is_int64, long L, const int* pos_dev, int n,
                            const float* const* tables, const int* V, const int* d, int B, int R, void* out,
                            void* stream);
int mgx_sample_head(const void* h, int B, int R, int d, const float* gamma, const float* beta, float eps,
                    const void* W, const float* bias, int n, const int* V, int stochastic, unsigned long long seed,
                    const int* pos_dev, unsigned long long* keys, float* logits, void* stream);
int mgx_sample_commit(void* seq, int seq_is_int64, long L, int B, int n, int* pos_dev, unsigned long long* keys,
                      void* stream);

/* Token embedding for training: out[T,d] bf16 = table[tokens[t], :] (fp32 [V,d]) rounded to nearest even:
 * `self.embed(x)` + the autocast cast in one pass (alibi_unified.py:130-131).  tokens int64 [T], clamped to [0, V). */
int mgx_embed_rows_bf16(const long long* tokens, const float* table, int V, int d, long T, void* out, void* stream);

/* The same for several token streams at once: tokens [B,n,S] int64 (n <= 4), tables[k] fp32 [V[k], d[k]] (HOST arrays)
 * -> out [B*S, d[0]+..+d[n-1]] bf16 = concat_k table_k[tokens[b,k,s], :]: the three gathers, the concatenation and the
 * autocast cast of alibi_separated.py:135-139 in one pass.  d[k] multiples of 4. */
int mgx_embed_cat_bf16(const long long* tokens, int n, int B, int S, const float* const* tables, const int* V,
                       const int* d, void* out, void* stream);

/* Column sums of a bf16 matrix x [T,n] -> out [n] fp32: the bias gradient of a projection whose output gradient no
 * other kernel of the block reduces (c_attn: dqkv.sum(0), autograd of alibi_unified.py:59 / base_layers.py:54-57).
 * workspace: mgx_colsum_workspace_floats(n) floats.  Fixed summation order (deterministic). */
long mgx_colsum_workspace_floats(int n);
int mgx_colsum_bf16(const void* x, float* out, float* workspace, long T, int n, void* stream);

/* The same column sums with NO workspace argument and no allocation of any kind (no hipMalloc / hipFree, no
 * allocating library routine): it has no counterpart in the reference and exists for callers that cannot allocate —
 * code under HIP-graph capture, tight serving loops, harnesses that forbid hipMalloc.  The stage-1 partial sums live in
 * a device array inside the library: 128 x 8192 floats = 4 MiB on every GPU the library runs on, held until the
 * process ends.  n <= mgx_colsum_static_max_n() (8192; wider matrices are rejected with MGX_EUNSUPPORTED: use
 * mgx_colsum_bf16 with a workspace).  Results are bit-identical to mgx_colsum_bf16.  All calls on one GPU share that
 * GPU's array: calls that overlap in time on one GPU (two streams, two threads) race; calls on one stream do not.
 * The first call on a device looks up the array's address (hipGetSymbolAddress): make it outside graph capture. */
int mgx_colsum_static_max_n(void);
int mgx_colsum_bf16_static(const void* x, float* out, long T, int n, void* stream);

/* bf16 shadows of fp32 master weights (the `.to(x.dtype)` of every autocast projection, and the transposed
 * operand the GEMM libraries prefer): master [rows, cols] fp32 -> dst [rows, cols] bf16 and/or dst_t
 * [cols, rows] bf16 in one pass over the master.  `count` matrices per call (HOST arrays of device pointers;
 * dst[i] or dst_t[i] may be NULL, not both); rows and cols multiples of 64.  Round to nearest even, as
 * Tensor.to(torch.bfloat16). */
int mgx_cast_shadow_bf16(int count, const float* const* src, void* const* dst, void* const* dst_t,
                         const int* rows, const int* cols, void* stream);

/* ---- bf16 MFMA GEMMs for the projections ----------------------------------------------------
 * Replace Conv1d.forward (base_layers.py:54-57) and nn.Linear (alibi_unified.py:90,92) and their
 * autograd GEMMs.  Tile 256x256x64, fp32 accumulate.
 * NT: C[M,N] = A[M,K] . Bt[N,K]^T (+ bias[N] fp32) (+ res[M,N] bf16); all bf16, K % 64 == 0, N % 8 == 0,
 *     lda/ldb/ldc multiples of 8.  act = -1: nothing more.  act = MGX_ACT_*: with pre = the bf16-rounded
 *     A.Bt^T + bias, C2[M,N] = dropout(act(pre)) (alibi_unified.py:90-91; dropout keyed by (seed, m*N+n), the same
 *     mask mgx_bias_act_bwd regenerates) and C = the derivative factor act'(pre) * mask the backward multiplies by
 *     (mgx_gemm_nt_dfac_bf16) as ONE BYTE per element: C is then an [M,N] uint8 matrix with a row stride of ldc BYTES,
 *     value = (code - 27) * step, step = 0x1.47aep-8 with its low 8 mantissa bits cleared (~0.005: GELU' in
 *     [-0.129, 1.129] to +-0.0025) or 2^-7 for ReLU (0 / 1 exact); code 27 = a dropped element = exactly 0.  The
 *     pre-activation itself is not stored.  res must be NULL then.
 */
int mgx_gemm_nt_bf16(const void* A, const void* Bt, void* C, void* C2, const float* bias, const void* res,
                     int M, int N, int K, int lda, int ldb, int ldc, int act, float p_drop,
                     unsigned long long seed, void* stream);
/* NT with the backward of `drop(activation(.))` (alibi_unified.py:90-91) as its epilogue: the dgrad GEMM of
 * linear2 (autograd of alibi_unified.py:92) times the derivative factor the forward left, in one launch:
 *   C[M,N] = (A[M,K] . Bt[N,K]^T) * fac[M,N] / (1 - p_drop)      fac [M,N] uint8 codes (row stride ldc bytes) = C of
 *   mgx_gemm_nt_bf16 with the same act >= 0 and p_drop;
 *   colpart [mgx_gemm_nt_dfac_colpart_rows(M,N), N] fp32 (may be NULL) = partial column sums of the rounded C, two
 *   rows per row tile (one per wave row; the row tile is 256 rows, or 128 where 256-row tiles would leave a quarter
 *   of the chip idle - which is why the row count is a function and not a formula): their sum over rows is the bias
 *   gradient of linear1; exactly mgx_gemm_nt_dfac_colpart_floats(M,N) = rows * N floats are written, every one of
 *   them (no zero-fill needed), fixed order.
 *   dbias [N] fp32 (may be NULL; needs colpart): that sum, formed by a second small launch in a fixed order. */
int mgx_gemm_nt_dfac_colpart_rows(int M, int N);
long mgx_gemm_nt_dfac_colpart_floats(int M, int N);
int mgx_gemm_nt_dfac_bf16(const void* A, const void* Bt, const void* fac, void* C, float* colpart, float* dbias, int M,
                          int N, int K, int lda, int ldb, int ldc, int act, float p_drop, void* stream);
/* TN (weight gradients): C[M,N] = A[K,M]^T . B[K,N], bf16 inputs, split over K into fp32 slabs in
 * `workspace` (mgx_gemm_tn_workspace_floats() floats) and reduced in a fixed order (deterministic);
 * C is fp32 or bf16 (out_dtype).  M, N, lda, ldb multiples of 8. */
long mgx_gemm_tn_workspace_floats(int M, int N, int K, int* splits_out);
int mgx_gemm_tn_bf16(const void* A, const void* B, void* C, float* workspace, int M, int N, int K,
                     int lda, int ldb, int out_dtype, void* stream);
/* Two such weight gradients over the same K (token rows) in ONE launch: C0 = A0^T . B0 and C1 = A1^T . B1 (the
 * gradients of linear1 and linear2, alibi_unified.py:90,92, or of c_attn and c_proj, :58,66).  Together the two have
 * enough tiles for half as many K splits, i.e. half the fp32 slab traffic per gradient; results are bit-identical to
 * two mgx_gemm_tn_bf16 calls only where the split count happens to be the same (the summation order over K differs
 * otherwise; each order is fixed).  workspace: mgx_gemm_tn_pair_workspace_floats() floats. */
long mgx_gemm_tn_pair_workspace_floats(int M0, int N0, int M1, int N1, int K);
int mgx_gemm_tn_pair_bf16(const void* A0, const void* B0, void* C0, int M0, int N0, int lda0, int ldb0,
                          const void* A1, const void* B1, void* C1, int M1, int N1, int lda1, int ldb1, int K,
                          float* workspace, int out_dtype, void* stream);

/* ---- generic GEMM: the fp32 parity mode and every shape off the tuned kernels' grid ---------------------------
 * C[M,N] = opA(A)[M,K] . opB(B)[K,N] (+ bias[N] fp32) (+ res[M,N]); transA: A is stored [K,M]; transB: B is stored
 * [N,K] (nn.Linear's weight, alibi_unified.py:90,92,137; Conv1d's [in,out] weight, base_layers.py:54-57, is transB = 0).
 * Replaces torch.addmm / mm / F.linear wherever mgx_gemm_nt/tn/skinny do not apply: fp32 storage (north_star's 1e-5
 * mode), bf16 widths that are not multiples of 64, unpadded vocabularies.  Any sizes, any row strides (lda/ldb/ldc/ldr
 * in elements), a/b/c/res each MGX_F32 or MGX_BF16; products and sums in fp32 (v_mfma_f32_32x32x2_f32), K walked in
 * order in one launch (deterministic).  res may alias C.  A correctness path, not a tuned one. */
int mgx_gemm_generic(const void* A, const void* B, void* C, const float* bias, const void* res, int M, int N, int K,
                     int lda, int ldb, int ldc, int ldr, int transA, int transB, int a_dtype, int b_dtype, int c_dtype,
                     int res_dtype, void* stream);

/* ---- decode-time projections (M <= 256 token rows) --------------------------------------------------
 * out[M,N] (bf16) = epilogue(A[M,K] . Wt[N,K]^T + bias[N]); A, Wt, res, out bf16 row-major; bias / gamma / beta fp32.
 * Replaces, for the three token rows per sequence of a sampling step (inference_unified.py:15), Conv1d.forward /
 * nn.Linear (base_layers.py:54-57, alibi_unified.py:90,92) together with what follows them in the block:
 *   mode 0: nothing more (c_attn)               mode 1 / 2: GELU (exact erf) / ReLU (linear1, alibi_unified.py:90)
 *   mode 3: LayerNorm(. + res) * gamma + beta   (c_proj + norm1, linear2 + norm2: alibi_unified.py:88-89,92-94)
 * Split over N/64 column tiles x 48- or 64-row tiles x K slices up to 1024 deep.  Where one slice covers K
 * (modes 0-2 at K <= 1024) the epilogue runs in the same launch; otherwise fp32 partial slabs go to `workspace`
 * (mgx_skinny_gemm_workspace_floats floats; always required non-NULL) and a second launch reduces them and
 * applies the epilogue.
 * N % 64 == 0, K % 64 == 0; mode 3 needs N <= 2048. */
long mgx_skinny_gemm_workspace_floats(int M, int N, int K);
int mgx_skinny_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* res, const float* gamma,
                         const float* beta, float eps, void* out, float* workspace, int M, int N, int K, int mode,
                         void* stream);

/* ---- rotary position embedding on the packed projection ------------------------------------------
 * Replaces rotate_every_two / apply_rotary_pos_emb (old_models/rotary_transformer.py:9-21): q and k of
 * qkv [B,S,3,H,dh] are rotated IN PLACE pair-wise by the angles of cos_tab / sin_tab [S, dh/2] fp32
 * (FixedPositionalEmbedding, rotary_transformer.py:24-42: sin | cos halves of `emb`).  inverse != 0 applies the
 * transposed rotation: the backward, applied to dq | dk.  v is untouched.  The attention itself then runs
 * mgx_alibi_attn_* with zero slopes (plain causal). */
int mgx_rope_qk(void* qkv, const float* cos_tab, const float* sin_tab, int B, int S, int H, int dh, int inverse,
                int dtype, void* stream);

/* ---- unified MIDI token codec (the step either side of the model) ---------------------------------
 * Decode replaces UnifiedReconstruct.reconstruct_stream (data/transforms.py:339-365) with unflatten_notes_
 * (transforms.py:94-128) and the running time of stream_to_raw_stream_ (data/parse_midi_numba.py:196-204):
 *   tokens [B,L] int32 (sampled rows, BOS 0 / EOS-pad 1 handled as the reference does)
 *   dur_ticks [duration_bins] int32 = (bin_arr[d] * ticks) truncated, computed once on the host in float64
 *   events [B,L,3] int32: rows 0..counts[b]-1 = (note 0..87 or -1 for a time shift, velocity, duration in ticks)
 *   abs_time [B,L] int32: absolute tick at which event k starts (sum of the time shifts before it)
 *   counts [B] int32: number of events; -1 when the row has no note-class token (the reference raises IndexError)
 * Encode replaces flatten_notes (transforms.py:11-38), the window of transforms.py:172-176 (its random start is
 * the `starts` argument) and UnifiedEncoding.pad (transforms.py:149-152):
 *   notes / velocities / durations [B,Nmax] int32 (note 0 = time shift; durations already binned), lengths [B],
 *   starts [B] -> tokens [B,seq_len] int32.
 * Integer work, bit-exact. */
int mgx_unified_decode_events(const int* tokens, int B, int L, const int* dur_ticks, int note_bins, int velocity_bins,
                              int duration_bins, int* events, int* abs_time, int* counts, void* stream);
/* Separated stream (SeparatedReconstruct.reconstruct_stream, data/transforms.py:290-318): tokens [B,3,L] int32
 * (note / velocity / duration rows); outputs as mgx_unified_decode_events (counts is never negative here). */
int mgx_separated_decode_events(const int* tokens, int B, int L, const int* dur_ticks, int note_bins, int velocity_bins,
                                int duration_bins, int* events, int* abs_time, int* counts, void* stream);
int mgx_unified_encode_tokens(const int* notes, const int* velocities, const int* durations, const int* lengths,
                              const int* starts, int B, int Nmax, int seq_len, int note_bins, int velocity_bins,
                              int duration_bins, int* tokens, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
