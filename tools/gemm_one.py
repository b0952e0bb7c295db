"""One GEMM shape for PMC runs: python tools/gemm_one.py nt|tn M N K iters"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
kind, M, N, K, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.manual_seed(0)
if kind == "nt":
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    f = lambda: ops.gemm_nt(a, b)
else:
    a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
    f = lambda: ops.gemm_tn(a, b)
for _ in range(iters): f()
torch.cuda.synchronize()
