"""The qkv projection GEMM (M=32768 N=3072 K=1024) on all-zero, all-one and random operands: the same instruction stream,
different switching activity.  python tools/power_probe.py   (the kernels are power-bound on real data)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
T, N, K = 32768, 3072, 1024
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, mk in (("zeros", lambda *s: torch.zeros(*s, device="cuda").bfloat16()),
                 ("ones", lambda *s: torch.ones(*s, device="cuda").bfloat16()),
                 ("randn", lambda *s: torch.randn(*s, device="cuda").bfloat16()),
                 ("zeros", lambda *s: torch.zeros(*s, device="cuda").bfloat16()),
                 ("randn", lambda *s: torch.randn(*s, device="cuda").bfloat16())):
    a, w = mk(T, K), mk(N, K)
    t = timed(lambda: ops.gemm_nt(a, w))
    print(f"{name:6s} {t:7.1f} us  {2.0*T*N*K/t*1e-6:6.0f} TF")
