"""Summarise tools/traffic.sh: per-launch FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) of the roofline
kernels and the HBM byte estimate of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests of wide
coalesced streaming reads at 64 B on gfx950, so the read side is doubled; WRITE_SIZE is exact for 16-B stores."""
import csv, glob, json, sys, collections
root = sys.argv[1]
def per_launch(tag, ctr, pat, also=None):
    """mean counter value per launch of kernels matching pat; `also` = helper kernels whose traffic is charged to
    the same launches (split-K reduce, delta pre-pass)"""
    f = glob.glob(f"{root}/tr_{tag}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    vals = [float(r["Counter_Value"]) for r in rows if pat in r["Kernel_Name"]]
    extra = sum(float(r["Counter_Value"]) for r in rows if also and any(a in r["Kernel_Name"] for a in also))
    return (sum(vals) + extra) / len(vals), len(vals)
out = {}
T, d, ff = 32768, 1024, 4096
tn_algo = sum(2 * T * (m + n) + 4 * m * n for m, n in ((d, ff), (ff, d), (d, d), (d, 3 * d))) / 4   # bf16 in, fp32 dW out
for name, tagf, tagw, pat, algo, also in (
        ("attn_fwd_mfma_kernel<128,8> B16 S2048 H8", "attn_fetch", "attn_write", "attn_fwd_mfma_kernel",
         16 * 2048 * 3072 * 2 + 16 * 2048 * 1024 * 2 + 16 * 8 * 2048 * 4, None),
        ("attn_bwd dkv+dq+delta <128> B16 S2048 H8 (per backward = per dkv launch)", "attnb_fetch", "attnb_write",
         "attn_bwd_dkv_pipe_kernel", 2 * 16 * 2048 * 3072 * 2 + 2 * 16 * 2048 * 1024 * 2 + 16 * 8 * 2048 * 4,
         ("attn_bwd_dq_pipe_kernel", "attn_bwd_stats_kernel")),
        ("gemm_tn_kernel + splitk_reduce, mean of dw2/dw1/dwp/dwqkv at T=32768 d=1024 ff=4096", "tn_fetch", "tn_write",
         "gemm_tn_kernel", tn_algo, ("splitk_reduce",)),
        ("gemm_nt_kernel<gelu,dropout> mlp1 T=32768 d=1024 ff=4096", "tn_fetch", "tn_write", "gemm_nt_kernel",
         2 * T * d + 2 * ff * d + 2 * 2 * T * ff, None),
        ("decode_attn_bf16_kernel B64 H8 dh128 R3 ctx2112", "dec_fetch", "dec_write", "decode_attn",
         2 * 64 * 8 * 2112 * 128 * 2 + 2 * 64 * 3 * 1024 * 2 + 64 * 3 * 3072 * 2, None)):
    fk, n = per_launch(tagf, "FETCH_SIZE", pat, also)
    wk, _ = per_launch(tagw, "WRITE_SIZE", pat, also)
    out[name] = {"launches": n, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                 "hbm_bytes_corrected": 2 * fk * 1024 + wk * 1024, "hbm_bytes_uncorrected": (fk + wk) * 1024,
                 "algorithmic_bytes": algo}
print(json.dumps(out, indent=1))
