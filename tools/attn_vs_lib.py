"""The ALiBi attention kernels of libmgx beside torch's library attention (scaled_dot_product_attention, causal, NO bias:
the library's fused kernels take no ALiBi term) at the bench shape, on the same box:
    python tools/attn_vs_lib.py [B S H dh iters]
A yardstick only: the product never calls the library."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
from musicgeneration_amd.model_baseline.alibi_unified import get_slopes

a = sys.argv[1:]
B, S, H, dh = (int(a[i]) if len(a) > i else v for i, v in enumerate((16, 2048, 8, 128)))
iters = int(a[4]) if len(a) > 4 else 10
torch.manual_seed(0)
qkv = torch.randn(B, S, 3 * H * dh, device="cuda").bfloat16()
dout = torch.randn(B, S, H * dh, device="cuda").bfloat16()
slopes = torch.tensor(get_slopes(H), device="cuda")
fl = 2.0 * B * H * S * S * dh                               # causal-algorithmic flops of the forward


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


out, lse = ops.alibi_attn_fwd(qkv, slopes, H)
t_f = timeit(lambda: ops.alibi_attn_fwd(qkv, slopes, H))
t_b = timeit(lambda: ops.alibi_attn_bwd(dout, qkv, out, lse, slopes, H))
print(f"B={B} S={S} H={H} dh={dh}")
print(f"libmgx (ALiBi)   fwd {t_f:8.1f} us {fl / t_f * 1e-6:6.0f} TF   bwd {t_b:8.1f} us {2.5 * fl / t_b * 1e-6:6.0f} TF")

# the library: [B, H, S, dh] operands (contiguous copies made outside the timed region: the library gets its best layout)
q, k, v = (t.reshape(B, S, H, dh).transpose(1, 2).contiguous().requires_grad_(True) for t in qkv.chunk(3, dim=-1))
g = dout.reshape(B, S, H, dh).transpose(1, 2).contiguous()
from torch.nn.attention import SDPBackend, sdpa_kernel
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("mem_efficient", SDPBackend.EFFICIENT_ATTENTION)):
    try:
        with sdpa_kernel(be):
            o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
            t_lf = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))

            def fb():
                o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
                torch.autograd.grad(o, (q, k, v), g)
            t_lfb = timeit(fb)
        print(f"library {name:14s} (no bias) fwd {t_lf:8.1f} us {fl / t_lf * 1e-6:6.0f} TF   "
              f"bwd {t_lfb - t_lf:8.1f} us {2.5 * fl / (t_lfb - t_lf) * 1e-6:6.0f} TF (fwd+bwd minus fwd)")
    except Exception as ex:                                  # a backend this build does not have
        print(f"library {name}: not available ({type(ex).__name__}: {str(ex)[:120]})")
