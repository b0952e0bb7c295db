"""Summarise tools/traffic.sh: per-launch FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them) of the roofline
kernels and the HBM byte estimate of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 128-B requests of wide
coalesced streaming reads at 64 B on gfx950, so the read side is doubled; WRITE_SIZE is exact for 16-B stores."""
import csv, glob, json, sys, collections
root = sys.argv[1]
def per_launch(tag, ctr, pat):
    f = glob.glob(f"{root}/tr_{tag}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    return sum(vals) / len(vals), len(vals)
out = {}
for name, tagf, tagw, pat, algo in (
        ("attn_fwd_mfma_kernel<128,8> B16 S2048 H8", "attn_fetch", "attn_write", "attn_fwd_mfma_kernel",
         16 * 2048 * 3072 * 2 + 16 * 2048 * 1024 * 2 + 16 * 8 * 2048 * 4),
        ("decode_attn_bf16_kernel B64 H8 dh128 R3 ctx2112", "dec_fetch", "dec_write", "decode_attn",
         2 * 64 * 8 * 2112 * 128 * 2 + 2 * 64 * 3 * 1024 * 2 + 64 * 3 * 3072 * 2)):
    fk, n = per_launch(tagf, "FETCH_SIZE", pat)
    wk, _ = per_launch(tagw, "WRITE_SIZE", pat)
    out[name] = {"launches": n, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                 "hbm_bytes_corrected": 2 * fk * 1024 + wk * 1024, "hbm_bytes_uncorrected": (fk + wk) * 1024,
                 "algorithmic_bytes": algo}
print(json.dumps(out, indent=1))
