"""bias_act_fwd in the training step's form (bias already in the GEMM: activation + dropout only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
T, ff = 32768, 4096
u = torch.randn(T, ff, device="cuda").bfloat16(); b = torch.randn(ff, device="cuda")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
t1 = timeit(lambda: ops.bias_act_fwd(u, None, "gelu", 0.1, 7))
t2 = timeit(lambda: ops.bias_act_fwd(u, b, "gelu", 0.1, 7, inplace_pre=False))
print(f"act only {t1*1e3:7.1f} us {2*T*ff*2/t1/1e6:6.0f} GB/s   bias+pre+act {t2*1e3:7.1f} us {3*T*ff*2/t2/1e6:6.0f} GB/s")
