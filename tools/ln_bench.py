"""LayerNorm forward / backward of the bench shape with and without dropout: python tools/ln_bench.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T, d = 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
a = torch.randn(T, d, device="cuda").bfloat16(); r = torch.randn(T, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda"); b = torch.zeros(d, device="cuda")
dy = torch.randn(T, d, device="cuda").bfloat16(); dy2 = torch.randn(T, d, device="cuda").bfloat16()
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for p in (0.0, 0.1, 0.0, 0.1):
    y, z, mean, rstd = ops.ln_residual_fwd(a, r, g, b, 1e-5, p, 5)
    tf = timed(lambda: ops.ln_residual_fwd(a, r, g, b, 1e-5, p, 5))
    tb = timed(lambda: ops.ln_residual_bwd(dy, dy2, z, mean, rstd, g, p, 5, True))
    print(f"p={p}: fwd {tf:6.1f} us ({4 * T * d * 2 / tf * 1e-6:5.2f} TB/s of 4 tensors)   bwd {tb:6.1f} us")
