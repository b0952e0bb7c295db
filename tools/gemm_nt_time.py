"""python tools/gemm_nt_time.py M N K : time the libmgx NT GEMM (A/B runs via MGX_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
for _ in range(3): ops.gemm_nt(a, b)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.gemm_nt(a, b)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 20
print(f"NT {M}x{N}x{K}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF")
