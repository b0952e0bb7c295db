"""Micro-benchmark of the skew relative attention kernels (not part of the product).
usage: python tools/rel_bench.py [B S H dh iters]   prints forward / backward time and a checksum of every output
(the checksums of two libraries built from different sources must agree to the last digit when only addressing changed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
a = sys.argv[1:]
B, S, H, dh = (int(a[i]) if len(a) > i else v for i, v in enumerate((8, 4096, 8, 64)))
iters = int(a[4]) if len(a) > 4 else 10
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device='cuda').bfloat16()
dout = torch.randn(B, S, H * dh, device='cuda').bfloat16()
E = (0.5 * torch.randn(H, S, dh, device='cuda')).bfloat16()
out, lse = ops.rel_attn_fwd(qkv, E, H)
dqkv, dE = ops.rel_attn_bwd(dout, qkv, out, lse, E, H)
torch.cuda.synchronize()
def cs(t):
    t = t.float()
    return f"{t.double().sum().item():+.10e} {t.double().abs().sum().item():.10e}"
print("out ", cs(out)); print("lse ", cs(lse)); print("dqkv", cs(dqkv)); print("dE  ", cs(dE))
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
fl = 1.0 * B * H * S * S * dh
ms = timeit(lambda: ops.rel_attn_fwd(qkv, E, H))
print(f"fwd  {ms*1e3:8.1f} us  {3*fl/ms/1e9:7.1f} TFLOP/s (3 products at half the square)")
ms = timeit(lambda: ops.rel_attn_bwd(dout, qkv, out, lse, E, H))
print(f"bwd  {ms*1e3:8.1f} us  {6*fl/ms/1e9:7.1f} TFLOP/s (6 products at half the square; with workspace allocation)")
