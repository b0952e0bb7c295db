"""Micro-benchmark of the HBM-bound kernels at the training shapes (not part of the product)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
T, d, ff = 32768, 1024, 4096
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
torch.manual_seed(0)
u = torch.randn(T, ff, device="cuda").bfloat16(); b = torch.randn(ff, device="cuda")
g = torch.randn(T, ff, device="cuda").bfloat16()
t = timeit(lambda: ops.bias_act_fwd(u, b, "gelu", 0.1, 7, inplace_pre=False))
print(f"bias_act_fwd  {t*1e3:7.1f} us  {3*T*ff*2/t/1e6:7.0f} GB/s")
t = timeit(lambda: ops.bias_act_bwd(g, u, "gelu", 0.1, 7))
print(f"bias_act_bwd  {t*1e3:7.1f} us  {3*T*ff*2/t/1e6:7.0f} GB/s")
a = torch.randn(T, d, device="cuda").bfloat16(); r = torch.randn(T, d, device="cuda").bfloat16()
gm = torch.ones(d, device="cuda"); bt = torch.zeros(d, device="cuda")
t = timeit(lambda: ops.ln_residual_fwd(a, r, gm, bt, 1e-5, 0.1, 7))
print(f"ln_fwd        {t*1e3:7.1f} us  {4*T*d*2/t/1e6:7.0f} GB/s")
y, z, mean, rstd = ops.ln_residual_fwd(a, r, gm, bt, 1e-5, 0.1, 7)
t = timeit(lambda: ops.ln_residual_bwd(a, r, z, mean, rstd, gm, 0.1, 7, want_dsum=True))
print(f"ln_bwd        {t*1e3:7.1f} us  {5*T*d*2/t/1e6:7.0f} GB/s")
