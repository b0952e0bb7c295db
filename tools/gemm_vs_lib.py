"""The step's GEMM shapes through libmgx and through torch's library GEMM (hipBLASLt / rocBLAS behind torch.matmul), on
the same box, interleaved so both see the same clocks:  python tools/gemm_vs_lib.py [iters] [T]
A yardstick only: the product never calls the library GEMM."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
torch.manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters           # us


def rnd(*s):
    return (torch.randn(*s, device="cuda") * 0.5).bfloat16()


rows = []
# forward / dgrad: C[T,N] = A[T,K] W[N,K]^T
for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("mlp1", 4096, 1024), ("mlp2", 1024, 4096),
                   ("dgrad qkv", 1024, 3072), ("head", 624, 1024)):
    a, w = rnd(T, K), rnd(N, K)
    t_lib = timed(lambda: F.linear(a, w))
    t_mgx = timed(lambda: ops.gemm_nt(a, w))
    t_lib2 = timed(lambda: F.linear(a, w))
    rows.append((f"nt {name} N={N} K={K}", 2.0 * T * N * K, t_mgx, min(t_lib, t_lib2)))
    del a, w
# weight gradients: dW[M,N] = A[T,M]^T B[T,N] (fp32 out in libmgx; the library's bf16 output is the cheaper job)
for name, M, N in (("dw2", 1024, 4096), ("dw1", 4096, 1024), ("dwp", 1024, 1024), ("dwqkv", 1024, 3072)):
    a, b = rnd(T, M), rnd(T, N)
    t_lib = timed(lambda: torch.matmul(a.t(), b))
    t_mgx = timed(lambda: ops.gemm_tn(a, b))
    t_lib2 = timed(lambda: torch.matmul(a.t(), b))
    rows.append((f"tn {name} M={M} N={N}", 2.0 * T * M * N, t_mgx, min(t_lib, t_lib2)))
    del a, b
print(f"T={T} iters={iters}   (us per launch, TFLOP/s)")
for name, fl, tm, tl in rows:
    print(f"{name:28s} mgx {tm:8.1f} us {fl / tm * 1e-6:7.0f}   lib {tl:8.1f} us {fl / tl * 1e-6:7.0f}   mgx/lib time {tm / tl:5.2f}")
