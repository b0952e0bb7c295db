"""Forward / backward time of the skew relative attention at the bench shape: python tools/rel_time.py [B S H dh]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
a = sys.argv[1:]
B, S, H, dh = (int(a[i]) if len(a) > i else v for i, v in enumerate((8, 4096, 8, 64)))
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device='cuda').bfloat16()
dout = torch.randn(B, S, H * dh, device='cuda').bfloat16()
E = (0.5 * torch.randn(H, S, dh, device='cuda')).bfloat16()
out, lse = ops.rel_attn_fwd(qkv, E, H)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f"fwd {timeit(lambda: ops.rel_attn_fwd(qkv, E, H)):7.1f} us  bwd {timeit(lambda: ops.rel_attn_bwd(dout, qkv, out, lse, E, H)):7.1f} us")
