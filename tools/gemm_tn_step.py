"""The four weight-gradient GEMMs of one decoder block at the bench shape (T = 32768 token rows), in the step's mix,
plus the fused MLP up-projection (PMC runs): python tools/gemm_tn_step.py iters"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T, d, ff = 32768, 1024, 4096
torch.manual_seed(0)
x = torch.randn(T, d, device="cuda").bfloat16(); y = torch.randn(T, d, device="cuda").bfloat16()
h = torch.randn(T, ff, device="cuda").bfloat16(); q = torch.randn(T, 3 * d, device="cuda").bfloat16()
w1 = torch.randn(ff, d, device="cuda").bfloat16(); b1 = torch.randn(ff, device="cuda")
for _ in range(iters):
    ops.gemm_tn(y, h)      # dw2   [d, ff]
    ops.gemm_tn(h, x)      # dw1   [ff, d]
    ops.gemm_tn(x, y)      # dwp   [d, d]
    ops.gemm_tn(x, q)      # dwqkv [d, 3d]
    ops.gemm_nt(x, w1, b1, None, "gelu", 0.1, 7)   # mlp1 with bias + GELU + dropout (two outputs)
torch.cuda.synchronize()
