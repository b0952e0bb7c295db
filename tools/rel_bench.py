"""Micro-benchmark of the skew relative attention kernels (run under rocprofv3 for per-kernel times)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
a = sys.argv[1:]
B, S, H, dh = (int(a[i]) if len(a) > i else v for i, v in enumerate((8, 4096, 8, 64)))
iters = int(a[4]) if len(a) > 4 else 5
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device='cuda').bfloat16()
dout = torch.randn(B, S, H * dh, device='cuda').bfloat16()
E = (0.5 * torch.randn(H, S, dh, device='cuda')).bfloat16()
for _ in range(iters):
    out, lse = ops.rel_attn_fwd(qkv, E, H)
    ops.rel_attn_bwd(dout, qkv, out, lse, E, H)
torch.cuda.synchronize()
print("done")
