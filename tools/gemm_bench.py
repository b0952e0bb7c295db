"""GEMM micro-benchmark: libmgx kernels vs torch (hipBLASLt) at the training shapes."""
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
for name, (M, N, K) in {"qkv": (T, 3 * d, d), "proj": (T, d, d), "mlp1": (T, ff, d), "mlp2": (T, d, ff)}.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); bt = torch.randn(N, K, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.gemm_nt(a, bt)); t2 = timeit(lambda: torch.mm(a, bt.t()))
    print(f"NT {name:5s} M={M} N={N} K={K}: mgx {t1*1e3:7.1f} us {fl/t1/1e9:7.1f} TF | torch {t2*1e3:7.1f} us {fl/t2/1e9:7.1f} TF")
for name, (K, M, N) in {"dWqkv": (T, d, 3 * d), "dWp": (T, d, d), "dW1": (T, ff, d), "dW2": (T, d, ff)}.items():
    a = torch.randn(K, M, device="cuda").bfloat16(); b = torch.randn(K, N, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    t1 = timeit(lambda: ops.gemm_tn(a, b)); t2 = timeit(lambda: torch.mm(a.t(), b))
    print(f"TN {name:5s} M={M} N={N} K={K}: mgx {t1*1e3:7.1f} us {fl/t1/1e9:7.1f} TF | torch {t2*1e3:7.1f} us {fl/t2/1e9:7.1f} TF")
# fused epilogue (mlp1: bias + GELU + dropout, two outputs)
M, N, K = T, ff, d
a = torch.randn(M, K, device="cuda").bfloat16(); bt = torch.randn(N, K, device="cuda").bfloat16(); bias = torch.randn(N, device="cuda")
fl = 2.0 * M * N * K
t1 = timeit(lambda: ops.gemm_nt(a, bt, bias, None, "gelu", 0.1, 7))
def unfused():
    u = torch.mm(a, bt.t()); return ops.bias_act_fwd(u, bias, "gelu", 0.1, 7)
t2 = timeit(unfused)
print(f"NT mlp1+gelu+dropout: mgx fused {t1*1e3:7.1f} us {fl/t1/1e9:7.1f} TF | torch.mm + bias_act_fwd {t2*1e3:7.1f} us")
