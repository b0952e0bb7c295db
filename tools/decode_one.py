"""A few launches of the decode attention kernel at one context length (PMC runs): python tools/decode_one.py ctx [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
ctx = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, dh, R, Lmax = 64, 8, 128, 3, 4161
d = H * dh
kc = torch.randn(B, H, Lmax, dh, device="cuda").to(torch.bfloat16)
vc = torch.randn(B, H, Lmax, dh, device="cuda").to(torch.bfloat16)
qkv = torch.randn(B, R, 3 * d, device="cuda").to(torch.bfloat16)
slopes = (2.0 ** (-torch.arange(1, H + 1, dtype=torch.float32))).cuda()
for _ in range(iters):
    ops.decode_step_attn(qkv, kc, vc, slopes, ctx - R)
torch.cuda.synchronize()
