"""Debug helper: compare the MFMA attention backward against the generic kernels; prints the rows that differ."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from musicgeneration_amd import ops
from musicgeneration_amd.model_baseline.alibi_unified import get_slopes
B, S, H, dh = (int(x) for x in sys.argv[1:5])
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device='cuda').bfloat16()
dout = torch.randn(B, S, H * dh, device='cuda').bfloat16()
slopes = torch.tensor(get_slopes(H), device='cuda')
out, lse = ops.alibi_attn_fwd(qkv, slopes, H)
a = ops.alibi_attn_bwd(dout, qkv, out, lse, slopes, H).float().view(B, S, 3, H, dh)
b = ops.alibi_attn_bwd(dout, qkv, out, lse, slopes, H, generic=True).float().view(B, S, 3, H, dh)
for t, name in enumerate("qkv"):
    d = (a[:, :, t] - b[:, :, t])
    bad = (~torch.isfinite(d)) | (d.abs() > 0.05 * b[:, :, t].abs().max())
    rows = bad.any(-1).nonzero()
    print(f"d{name}: max abs diff {d[torch.isfinite(d)].abs().max().item():.4f} ref max {b[:, :, t].abs().max().item():.3f} bad (b, row, head) count {len(rows)}")
    if len(rows):
        rr = sorted(set((int(r[0]), int(r[1]), int(r[2])) for r in rows))
        print("   first:", rr[:12], " last:", rr[-6:])
        r = rr[0]
        print("   got", a[r[0], r[1], t, r[2], :8].tolist(), "\n   ref", b[r[0], r[1], t, r[2], :8].tolist())
# where did the bad dq rows come from?  nearest reference row / column block for a few of them
aq, bq_ = a[0, :, 0, 0], b[0, :, 0, 0]            # [S, dh] of head 0
for r in (0, 1, 8, 9):
    if r < S:
        d = ((bq_ - aq[r][None]) ** 2).sum(-1)
        print(f"row {r}: nearest ref row {int(d.argmin())} dist {float(d.min()):.4f}; own dist {float(d[r]):.4f}; per-64 halves own dist "
              f"{float(((bq_[r,:64]-aq[r,:64])**2).sum()):.4f} {float(((bq_[r,64:]-aq[r,64:])**2).sum()):.4f}")
err = ((aq - bq_) ** 2).sum(-1).sqrt() / (bq_ ** 2).sum(-1).sqrt().clamp_min(1e-3)
print("rows with rel err > 0.05:", [int(r) for r in (err > 0.05).nonzero().flatten()][:80])
ratio = (aq.abs().sum(-1) / bq_.abs().sum(-1).clamp_min(1e-6))
print("|got|/|ref| per row:", [round(float(x), 2) for x in ratio[:40]])
