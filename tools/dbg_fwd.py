"""Debug helper: compare the MFMA attention forward against the generic kernel; prints where they differ."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from musicgeneration_amd import ops
from musicgeneration_amd.model_baseline.alibi_unified import get_slopes
B, S, H, dh = (int(x) for x in sys.argv[1:5])
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device='cuda').bfloat16()
slopes = torch.tensor(get_slopes(H), device='cuda')
out, lse = ops.alibi_attn_fwd(qkv, slopes, H)
ref, lref = ops.alibi_attn_fwd(qkv, slopes, H, generic=True)
a = out.float().view(B, S, H, dh); b = ref.float().view(B, S, H, dh)
d = a - b
bad = (~torch.isfinite(d)) | (d.abs() > 0.03)
print("bad elements", int(bad.sum()), "of", bad.numel(), " lse max diff", float((lse - lref).abs().max()), "lse finite", bool(torch.isfinite(lse).all()))
rows = bad.any(-1).nonzero()
print("bad (b,row,head):", [tuple(int(x) for x in r) for r in rows[:20]], "... count", len(rows))
cols = bad.any(0).any(0).any(0).nonzero().flatten().tolist()
print("bad columns:", cols)
if len(rows):
    r = rows[0]
    print("got", [round(float(x), 3) for x in a[r[0], r[1], r[2], :16]])
    print("ref", [round(float(x), 3) for x in b[r[0], r[1], r[2], :16]])
    print("lse got/ref", float(lse[r[0], r[2], r[1]]), float(lref[r[0], r[2], r[1]]))
