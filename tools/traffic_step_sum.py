"""Summarise tools/traffic_step.sh: per-launch FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) of the roofline
kernels inside one training step of the bench workload, and the HBM byte estimate of MI355X_MICROARCH.md (HBM
section): FETCH_SIZE counts 128-B requests of wide coalesced streaming reads at 64 B on gfx950, so the read side is
doubled; WRITE_SIZE is exact for 16-B stores.  Keys are bench.py's event names."""
import csv, glob, json, sys
root = sys.argv[1]
def per_launch(tag, ctr, pat, also=None):
    """mean counter value per launch of kernels matching pat; `also` = helper kernels whose traffic is charged to
    the same launches (split-K reduce, dq and delta kernels of one attention backward)"""
    f = sorted(glob.glob(f"{root}/tr2_{tag}/**/*counter_collection.csv", recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    vals = [float(r["Counter_Value"]) for r in rows if pat in r["Kernel_Name"]]
    extra = sum(float(r["Counter_Value"]) for r in rows if also and any(a in r["Kernel_Name"] for a in also))
    return (sum(vals) + extra) / len(vals), len(vals)
T, d, ff, L, Vp = 32768, 1024, 4096, 6, 640
def nt(M, N, K, extra_mn=0):          # bf16 operands and output; extra_mn = further M x N bf16 streams (residual, pre-activation)
    return 2 * (M * K + N * K + M * N * (1 + extra_mn))
fwd = nt(T, 3 * d, d) + nt(T, d, d, 1) + nt(T, ff, d, 1) + nt(T, d, ff, 1)      # linear1 writes two outputs (activated, derivative factor)
bwd = nt(T, ff, d, 1) + nt(T, d, ff) + nt(T, d, d) + nt(T, d, 3 * d)      # first one: dgrad x derivative factor (reads fac)
nt_algo = (L * (fwd + bwd) + nt(T, Vp, d) + nt(T, d, Vp)) / (8 * L + 2)
tn = lambda m, n: 2 * T * (m + n) + 4 * m * n                               # bf16 in, fp32 dW out
tn_algo = (L * (tn(d, ff) + tn(ff, d) + tn(d, d) + tn(d, 3 * d)) + tn(d, Vp) + tn(Vp, d)) / (2 * L + 2)   # the block's four gradients leave as two paired launches
out = {}
for key, name, tagf, tagw, pat, algo, also in (
        ("gemm_nt", "gemm_nt_kernel, the step's 50 launches (24 forward, 24 dgrad, head forward + dgrad)", "step_fetch",
         "step_write", "gemm_nt_kernel", nt_algo, None),
        ("gemm_tn", "gemm_tn_kernel + splitk_reduce, the step's 14 launches (12 paired block weight gradients, head, embedding)",
         "step_fetch", "step_write", "gemm_tn_kernel", tn_algo, ("splitk_reduce",)),
        ("alibi_attn_fwd", "attn_fwd_mfma_kernel<128,8> B16 S2048 H8", "step_fetch", "step_write", "attn_fwd_mfma_kernel",
         16 * 2048 * 3072 * 2 + 16 * 2048 * 1024 * 2 + 16 * 8 * 2048 * 4, None),
        ("alibi_attn_bwd", "attn_bwd stats + dkv_pipe + dq_pipe <128> B16 S2048 H8 (per backward = per dkv launch)", "step_fetch",
         "step_write", "attn_bwd_dkv_pipe_kernel", 2 * 16 * 2048 * 3072 * 2 + 2 * 16 * 2048 * 1024 * 2 + 16 * 8 * 2048 * 4,
         ("attn_bwd_dq_pipe_kernel", "attn_bwd_stats_kernel")),
        ("decode_attn", "decode_attn_bf16_kernel B64 H8 dh128 R3 ctx2112", "dec_fetch", "dec_write", "decode_attn",
         2 * 64 * 8 * 2112 * 128 * 2 + 2 * 64 * 3 * 1024 * 2 + 64 * 3 * 3072 * 2, None)):
    fk, n = per_launch(tagf, "FETCH_SIZE", pat, also)
    wk, _ = per_launch(tagw, "WRITE_SIZE", pat, also)
    out[key] = {"kernel": name, "launches": n, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                "hbm_bytes_corrected": 2 * fk * 1024 + wk * 1024, "hbm_bytes_uncorrected": (fk + wk) * 1024,
                "algorithmic_bytes": algo}
print(json.dumps(out, indent=1))
