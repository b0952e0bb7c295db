"""Decode projection shapes through mgx_skinny_gemm_bf16 vs torch.addmm (hipBLASLt): python tools/skinny_time.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192
d, ff = 1024, 4096
torch.manual_seed(0)
def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
res = []
for name, (N, K) in {"qkv": (3 * d, d), "proj": (d, d), "mlp1": (ff, d), "mlp2": (d, ff)}.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); wt = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); ws = torch.empty(ops.skinny_workspace_floats(M, N, K), device="cuda")
    wkn = wt.t().contiguous(); bb = bias.bfloat16()
    g = torch.cuda.CUDAGraph()   # replayed as a graph, as the sampler does
    out = ops.skinny_gemm(a, wt, bias, ws)
    with torch.cuda.graph(g):
        for _ in range(10): out = ops.skinny_gemm(a, wt, bias, ws)
    t1 = timeit(g.replay, 50) / 10
    g2 = torch.cuda.CUDAGraph()
    torch.addmm(bb, a, wkn)
    with torch.cuda.graph(g2):
        for _ in range(10): o2 = torch.addmm(bb, a, wkn)
    t2 = timeit(g2.replay, 50) / 10
    res.append(f"{name} {t1:5.1f}/{t2:5.1f}")
print("us mgx/torch: " + "  ".join(res))
