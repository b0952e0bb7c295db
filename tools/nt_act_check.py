"""bitwise comparison of the activation epilogue between two builds of libmgx: python tools/nt_act_check.py save|cmp FILE"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd import ops
mode, path = sys.argv[1], sys.argv[2]
torch.manual_seed(0)
outs = {}
for (M, N, K) in ((1024, 4096, 1024), (300, 264, 192), (8192, 1024, 256)):
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    # special pre-activations: exact zeros, tiny, huge, inf, nan rows
    x[0] = 0; b2 = b.clone(); b2[:8] = 0
    x[1] = 1e-9; x[2] = 300.0; x[3] = -300.0; x[4] = float("inf"); x[5] = float("nan"); x[6] = 3e-5; x[7] = -2e-7
    for p in (0.0, 0.1):
        for act in ("gelu", "relu"):
            fac, h = ops.gemm_nt(x, w, b2, None, act, p, 99)
            outs[f"{M}.{N}.{K}.{p}.{act}.fac"] = fac.cpu()
            outs[f"{M}.{N}.{K}.{p}.{act}.h"] = h.view(torch.int16).cpu()
if mode == "save":
    torch.save(outs, path)
else:
    ref = torch.load(path)
    bad = 0
    for k, v in outs.items():
        same = torch.equal(v, ref[k])
        if not same:
            d = (v != ref[k])
            bad += 1
            print("DIFF", k, int(d.sum()), "of", d.numel())
    print("identical" if bad == 0 else f"{bad} tensors differ")
