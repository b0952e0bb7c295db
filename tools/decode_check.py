"""Debug aid: per-row error of the bf16 streaming decode kernel against fp32 torch."""
import sys
import torch
from musicgeneration_amd import ops

def run(B, H, dh, R, pos0):
    L = pos0 + R
    d = H * dh
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, L, 3 * d, generator=g).to(torch.bfloat16).cuda()
    slopes = (2.0 ** (-torch.arange(1, H + 1, dtype=torch.float32) * 8 / H)).cuda()
    q, k, v = (t.view(B, L, H, dh).transpose(1, 2).float() for t in qkv.split(d, -1))
    idx = torch.arange(L, device="cuda")
    bias = slopes.view(H, 1, 1) * (idx.view(1, 1, L) - idx.view(1, L, 1))
    sc = q[:, :, pos0:] @ k.transpose(-1, -2) / dh ** 0.5 + bias[:, pos0:]
    sc = sc.masked_fill(idx.view(1, 1, 1, L) > idx[pos0:].view(1, 1, R, 1), float("-inf"))
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(B, R, H, dh)
    kc = torch.zeros(B, H, L + 3, dh, dtype=torch.bfloat16).cuda()
    vc = torch.zeros_like(kc)
    ops.kv_append(qkv.contiguous(), kc, vc, 0) if L <= 4 else [ops.kv_append(qkv[:, i:i + 1].contiguous(), kc, vc, i) for i in range(L)]
    out = ops.decode_attn(qkv[:, pos0:, :d].contiguous(), kc, vc, slopes, pos0).view(B, R, H, dh).float()
    err = (out - ref).abs()
    print(B, H, dh, R, pos0, "max err", float(err.max()), "ref max", float(ref.abs().max()))
    print(" per row", err.amax((0, 2, 3)).tolist(), "per head", err.amax((0, 1, 3)).tolist())
    print(" per d-chunk", err.view(B, R, H, dh // 8, 8).amax((0, 1, 2, 4)).tolist())

for cfg in [(1, 1, 128, 1, 0), (1, 1, 128, 1, 5), (1, 2, 128, 3, 70), (2, 8, 64, 3, 200)]:
    run(*cfg)
