"""Achieved HBM GB/s of mgx_decode_attn at the decode bench shape (B=64, H=8, dh=128, R=3)."""
import sys
import torch
from musicgeneration_amd import ops

B, H, dh, R = 64, 8, 128, 3
Lmax = 4161
d = H * dh
kc = torch.randn(B, H, Lmax, dh, device="cuda").to(torch.bfloat16)
vc = torch.randn(B, H, Lmax, dh, device="cuda").to(torch.bfloat16)
q = torch.randn(B, R, d, device="cuda").to(torch.bfloat16)
slopes = (2.0 ** (-torch.arange(1, H + 1, dtype=torch.float32))).cuda()
for ctx in (256, 1024, 2048, 4096):
    pos0 = ctx - R
    for _ in range(3):
        ops.decode_attn(q, kc, vc, slopes, pos0)
    torch.cuda.synchronize()
    n = 50
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        ops.decode_attn(q, kc, vc, slopes, pos0)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    by = 2.0 * B * H * ctx * dh * 2
    print(f"ctx {ctx:5d}: {us:8.1f} us  {by / us / 1e3:8.1f} GB/s")
