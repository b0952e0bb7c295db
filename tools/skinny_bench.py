"""Time the decode-step GEMMs (M = 3*64 rows) under the BLAS backends torch offers."""
import torch
M = 192
shapes = [("qkv", 1024, 3072, "nn"), ("proj", 1024, 1024, "nn"), ("mlp1", 1024, 4096, "nt"), ("mlp2", 4096, 1024, "nt")]
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print(lib, "unavailable", ex); continue
    for name, K, N, lay in shapes:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
        w = torch.randn((K, N) if lay == "nn" else (N, K), device="cuda", dtype=torch.bfloat16)
        ws = [w.clone() for _ in range(40)]   # rotate weights so they come from HBM, as in the real step
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 40
            ww = ws[i[0]]
            return torch.addmm(b, x, ww if lay == "nn" else ww.t())
        us = t(f)
        print(f"{lib:10s} {name:5s} {us:7.1f} us  {K*N*2/us/1e3:7.0f} GB/s(weights)")
