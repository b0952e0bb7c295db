"""Thin torch wrappers over the C ABI (include/mgx.h): tensors in, tensors out.

Each wrapper validates shapes/dtypes/contiguity on the host (a kernel fault can
reset the whole GPU), passes raw device pointers plus the current HIP stream to
libmgx.so and returns torch tensors.  The autograd.Function classes give the
fused operators a backward; `fused_decoder_block` runs a whole post-LN block
(alibi_unified.py:86-95) as one autograd node.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import lib, check

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_ACT = {"gelu": _lib.ACT_GELU, "relu": _lib.ACT_RELU}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"musicgeneration_amd kernels take float32 or bfloat16, got {t.dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: musicgeneration_amd runs on the GPU only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


# bench.py sets this to a dict to collect (start, end) HIP-event pairs around single-kernel launches
KERNEL_EVENTS = None


class _timed:
    """Records a HIP-event pair on the current stream around one kernel launch when KERNEL_EVENTS is a dict."""

    def __init__(self, name, work=0.0):
        self.name = name
        self.work = work          # algorithmic FLOPs (or bytes) of this launch, for bench.py's roofline lines

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if KERNEL_EVENTS is not None:
            self.e.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.s, self.e, self.work))
        return False


def new_seed() -> int:
    """Dropout seed drawn from torch's CPU generator (no device sync)."""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & ((1 << 62) - 1)


# ---------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------
def _attn_dt(t, generic):
    return _dt(t) | (_lib.GENERIC_KERNELS if generic else 0)


def alibi_attn_fwd(qkv: torch.Tensor, slopes: torch.Tensor, n_heads: int, generic: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """qkv [B,S,3*d] -> (out [B,S,d], lse [B,H,S]).  generic=True: the wave-per-row kernels even where MFMA applies."""
    _need(qkv, "qkv")
    _need(slopes, "slopes", torch.float32)
    B, S, d3 = qkv.shape
    d = d3 // 3
    if d3 != 3 * d or d % n_heads or slopes.numel() != n_heads:
        raise ValueError(f"bad attention shapes qkv={tuple(qkv.shape)} heads={n_heads} slopes={slopes.numel()}")
    out = torch.empty(B, S, d, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, n_heads, S, dtype=torch.float32, device=qkv.device)
    with _timed("alibi_attn_fwd", 2.0 * B * n_heads * S * S * (d // n_heads)):
        check(lib.mgx_alibi_attn_fwd(_p(qkv), _p(slopes), _p(out), _p(lse), B, S, n_heads, d // n_heads,
                                     _attn_dt(qkv, generic), _stream()), "mgx_alibi_attn_fwd")
    return out, lse


def alibi_attn_bwd(dout, qkv, out, lse, slopes, n_heads: int, generic: bool = False) -> torch.Tensor:
    _need(dout, "dout", qkv.dtype)
    B, S, d3 = qkv.shape
    d = d3 // 3
    if dout.shape != out.shape or out.shape != (B, S, d):
        raise ValueError("bad attention backward shapes")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(int(lib.mgx_alibi_attn_bwd_workspace_floats(B, S, n_heads)), dtype=torch.float32, device=qkv.device)
    with _timed("alibi_attn_bwd", 5.0 * B * n_heads * S * S * (d // n_heads)):      # 5 products at half the square
        check(lib.mgx_alibi_attn_bwd(_p(dout), _p(qkv), _p(out), _p(lse), _p(slopes), _p(dqkv), _p(delta), B, S, n_heads,
                                     d // n_heads, _attn_dt(qkv, generic), _stream()), "mgx_alibi_attn_bwd")
    return dqkv


def rel_attn_fwd(qkv: torch.Tensor, E: torch.Tensor, n_heads: int, generic: bool = False):
    _need(qkv, "qkv")
    _need(E, "E", qkv.dtype)
    B, S, d3 = qkv.shape
    d = d3 // 3
    H, M, dh = E.shape
    if d3 != 3 * d or H != n_heads or dh * H != d or S > M:
        raise ValueError(f"bad relative attention shapes qkv={tuple(qkv.shape)} E={tuple(E.shape)}")
    out = torch.empty(B, S, d, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=qkv.device)
    with _timed("rel_attn_fwd", 3.0 * B * H * S * S * dh):
        check(lib.mgx_rel_attn_fwd(_p(qkv), _p(E), _p(out), _p(lse), B, S, H, dh, M, _attn_dt(qkv, generic), _stream()),
              "mgx_rel_attn_fwd")
    return out, lse


def rel_attn_bwd(dout, qkv, out, lse, E, n_heads: int, generic: bool = False):
    _need(dout, "dout", qkv.dtype)
    B, S, d3 = qkv.shape
    H, M, dh = E.shape
    dqkv = torch.empty_like(qkv)
    dE = torch.empty(H, M, dh, dtype=torch.float32, device=qkv.device)
    dt = _attn_dt(qkv, generic)
    ws = torch.empty(int(lib.mgx_rel_attn_bwd_workspace_floats(B, S, H, dh, M, dt)), dtype=torch.float32, device=qkv.device)
    with _timed("rel_attn_bwd", 6.0 * B * H * S * S * dh):      # dV, dP, dQ (two terms), dK, dE at half the square
        check(lib.mgx_rel_attn_bwd(_p(dout), _p(qkv), _p(out), _p(lse), _p(E), _p(dqkv), _p(dE), _p(ws), B, S, H, dh, M,
                                   dt, _stream()), "mgx_rel_attn_bwd")
    return dqkv, dE


def rope_qk_(qkv: torch.Tensor, cos_tab: torch.Tensor, sin_tab: torch.Tensor, n_heads: int, inverse: bool = False):
    """Rotate q and k of the packed projection qkv [B,S,3*d] in place (rotary_transformer.py:16-21); tables
    [S, dh/2] fp32.  inverse=True is the backward (applied to dq | dk)."""
    _need(qkv, "qkv")
    _need(cos_tab, "cos_tab", torch.float32)
    _need(sin_tab, "sin_tab", torch.float32)
    B, S, d3 = qkv.shape
    dh = d3 // 3 // n_heads
    if d3 != 3 * n_heads * dh or cos_tab.shape != (S, dh // 2) or sin_tab.shape != (S, dh // 2):
        raise ValueError(f"rope_qk_: qkv {tuple(qkv.shape)} heads {n_heads} tables {tuple(cos_tab.shape)}")
    check(lib.mgx_rope_qk(_p(qkv), _p(cos_tab), _p(sin_tab), B, S, n_heads, dh, 1 if inverse else 0, _dt(qkv),
                          _stream()), "mgx_rope_qk")
    return qkv


# ---------------------------------------------------------------------------
# (dropout +) residual + LayerNorm
# ---------------------------------------------------------------------------
def ln_residual_fwd(a, res, gamma, beta, eps=1e-5, p_drop=0.0, seed=0, save=True):
    """y = LN(dropout(a) + res).  Returns (y, z, mean, rstd); z/mean/rstd are None when save=False."""
    _need(a, "a")
    _need(gamma, "gamma", torch.float32)
    _need(beta, "beta", torch.float32)
    d = a.shape[-1]
    T = a.numel() // d
    if res is not None:
        _need(res, "res", a.dtype)
        if res.shape != a.shape:
            raise ValueError("residual shape mismatch")
    if gamma.numel() != d or beta.numel() != d:
        raise ValueError("gamma/beta size mismatch")
    y = torch.empty_like(a)
    z = torch.empty_like(a) if save else None
    mean = torch.empty(T, dtype=torch.float32, device=a.device) if save else None
    rstd = torch.empty(T, dtype=torch.float32, device=a.device) if save else None
    # algorithmic bytes: a, res read; y, z written (bench.py's HBM lines)
    with _timed("ln_fwd", float(a.numel() * a.element_size() * (2 + (res is not None) + bool(save)))):
        check(lib.mgx_ln_residual_fwd(_p(a), _p(res), _p(gamma), _p(beta), _p(y), _p(z), _p(mean), _p(rstd), T, d, eps,
                                      float(p_drop), seed, _dt(a), _stream()), "mgx_ln_residual_fwd")
    return y, z, mean, rstd


_PARTIAL_ROWS = lib.mgx_ln_bwd_partial_rows()


def ln_residual_bwd(dy, dy2, z, mean, rstd, gamma, p_drop=0.0, seed=0, want_dsum=False):
    """Returns (dz, da, dgamma, dbeta[, dsum]); da is dz when p_drop == 0; dsum = da.sum(0) in fp32
    (the bias gradient of the projection feeding the branch), produced by the same pass."""
    _need(dy, "dy", z.dtype)
    if dy2 is not None:
        _need(dy2, "dy2", z.dtype)
    d = z.shape[-1]
    T = z.numel() // d
    if dy.numel() != z.numel():
        raise ValueError("dy shape mismatch")
    dz = torch.empty_like(z)
    da = torch.empty_like(z) if p_drop > 0 else dz
    dgamma = torch.empty(d, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(d, dtype=torch.float32, device=z.device)
    dsum = torch.empty(d, dtype=torch.float32, device=z.device) if want_dsum else None
    partial = torch.empty(_PARTIAL_ROWS * 3 * d, dtype=torch.float32, device=z.device)
    # algorithmic bytes: dy, (dy2), z read; dz, (da) written
    with _timed("ln_bwd", float(z.numel() * z.element_size() * (3 + (dy2 is not None) + (p_drop > 0)))):
        check(lib.mgx_ln_residual_bwd(_p(dy), _p(dy2), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(da), _p(dgamma),
                                      _p(dbeta), _p(dsum), _p(partial), T, d, float(p_drop), seed, _dt(z), _stream()),
              "mgx_ln_residual_bwd")
    if want_dsum:
        return dz, da, dgamma, dbeta, dsum
    return dz, da, dgamma, dbeta


# ---------------------------------------------------------------------------
# bias + activation (+ dropout)
# ---------------------------------------------------------------------------
def bias_act_fwd(u, bias, act="gelu", p_drop=0.0, seed=0, inplace_pre=True):
    """(pre, h) with pre = u + bias (written over u when inplace_pre) and h = dropout(act(pre)).
    bias=None: u already is the pre-activation (a GEMM with a bias epilogue produced it); it is returned as pre
    untouched and only h is written (one pass less over [T, n])."""
    _need(u, "u")
    n = u.shape[-1]
    T = u.numel() // n
    if bias is None:
        pre = None
    else:
        _need(bias, "bias", torch.float32)
        if bias.numel() != n:
            raise ValueError("bias size mismatch")
        pre = u if inplace_pre else torch.empty_like(u)
    h = torch.empty_like(u)
    check(lib.mgx_bias_act_fwd(_p(u), _p(bias), _p(pre), _p(h), T, n, _ACT[act], float(p_drop), seed, _dt(u), _stream()),
          "mgx_bias_act_fwd")
    return (u if bias is None else pre), h


def bias_act_bwd(dh, pre, act="gelu", p_drop=0.0, seed=0):
    _need(dh, "dh", pre.dtype)
    n = pre.shape[-1]
    T = pre.numel() // n
    dpre = torch.empty_like(pre)
    dbias = torch.empty(n, dtype=torch.float32, device=pre.device)
    partial = torch.empty(_PARTIAL_ROWS * n, dtype=torch.float32, device=pre.device)
    check(lib.mgx_bias_act_bwd(_p(dh), _p(pre), _p(dpre), _p(dbias), _p(partial), T, n, _ACT[act], float(p_drop), seed,
                               _dt(pre), _stream()), "mgx_bias_act_bwd")
    return dpre, dbias


# ---------------------------------------------------------------------------
# bf16 MFMA GEMMs
# ---------------------------------------------------------------------------
def gemm_nt(a, bt, bias=None, res=None, act=None, p_drop=0.0, seed=0):
    """C = a[M,K] @ bt[N,K]^T (+ bias) (+ res); with act: returns (fac, dropout(act(pre))) where pre is the
    bf16-rounded C and fac [M,N] uint8 holds the codes of act'(pre) * mask that gemm_nt_dfac multiplies by in the
    backward (one byte per element: include/mgx.h)."""
    _need(a, "a", torch.bfloat16)
    _need(bt, "bt", torch.bfloat16)
    M, K = a.shape
    N, K2 = bt.shape
    if K != K2 or K % 64:
        raise ValueError(f"gemm_nt: a {tuple(a.shape)} bt {tuple(bt.shape)}: K must match and be a multiple of 64")
    if N % 8:
        raise ValueError("gemm_nt: N must be a multiple of 8 (pad the weight)")
    if bias is not None:
        _need(bias, "bias", torch.float32)
        if bias.numel() != N:
            raise ValueError("gemm_nt: bias size")
    if res is not None:
        _need(res, "res", torch.bfloat16)
        if res.shape != (M, N):
            raise ValueError("gemm_nt: residual shape")
    c = torch.empty(M, N, dtype=torch.bfloat16 if act is None else torch.uint8, device=a.device)
    c2 = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if act is not None else None
    with _timed("gemm_nt", 2.0 * M * N * K):
        check(lib.mgx_gemm_nt_bf16(_p(a), _p(bt), _p(c), _p(c2), _p(bias), _p(res), M, N, K, K, K, N,
                                   -1 if act is None else _ACT[act], float(p_drop), seed, _stream()), "mgx_gemm_nt_bf16")
    return c if act is None else (c, c2)


FAC_ZERO = 27
FAC_STEP = {"gelu": float(__import__("numpy").array([0x3BA3D700], dtype="uint32").view("float32")[0]), "relu": 2.0 ** -7}


def decode_fac(fac, act):
    """fp32 values of the derivative-factor codes gemm_nt(..., act=act) stores: act'(pre) * mask, without the
    1 / (1 - p) of dropout (include/mgx.h).  For tests and diagnostics; the kernels decode in their epilogue."""
    return (fac.to(torch.float32) - FAC_ZERO) * FAC_STEP[act]


def gemm_nt_dfac(dy, bt, fac, act, p_drop=0.0):
    """Backward through `dropout(act(pre))` fused into the dgrad GEMM that produces its incoming gradient:
    dpre[M,N] = (dy[M,K] @ bt[N,K]^T) * fac / (1-p), with fac = the uint8 codes of act'(pre) * mask as
    gemm_nt(..., act=act, p_drop=p_drop) left them, and
    dbias[N] = dpre.sum(0) in fp32 (per-row-tile partial sums, reduced by the library in a fixed order).  Replaces one
    GEMM output + one bias_act_bwd pass."""
    _need(dy, "dy", torch.bfloat16)
    _need(bt, "bt", torch.bfloat16)
    _need(fac, "fac", torch.uint8)
    M, K = dy.shape
    N, K2 = bt.shape
    if K != K2 or K % 64 or N % 8 or fac.shape != (M, N):
        raise ValueError(f"gemm_nt_dfac: dy {tuple(dy.shape)} bt {tuple(bt.shape)} fac {tuple(fac.shape)}")
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dy.device)
    part = torch.empty(int(lib.mgx_gemm_nt_dfac_colpart_floats(M, N)), dtype=torch.float32, device=dy.device)
    dbias = torch.empty(N, dtype=torch.float32, device=dy.device)
    with _timed("gemm_nt", 2.0 * M * N * K):
        check(lib.mgx_gemm_nt_dfac_bf16(_p(dy), _p(bt), _p(fac), _p(c), _p(part), _p(dbias), M, N, K, K, K, N, _ACT[act],
                                        float(p_drop), _stream()), "mgx_gemm_nt_dfac_bf16")
    return c, dbias


def gemm_tn(a, b, out_dtype=torch.float32, tag="gemm_tn", out=None):
    """C[M,N] = a[K,M]^T @ b[K,N] (weight gradient), fp32 accumulation over a split K.  b may be a column slice of a
    wider row-major matrix (row stride a multiple of 8).
    tag: name under which bench.py's HIP-event pairs file this launch (the block's four weight gradients are the
    roofline population; the head and embedding gradients are filed apart).
    out: a contiguous [M,N] tensor to write into (a gradient-bucket slot, see grad_slot) instead of a fresh one."""
    _need(a, "a", torch.bfloat16)
    if not b.is_cuda or b.dtype != torch.bfloat16 or b.dim() != 2 or b.stride(1) != 1 or b.stride(0) % 8 or \
            b.stride(0) < b.shape[1] or b.storage_offset() % 8:
        raise ValueError("gemm_tn: b must be a bf16 GPU matrix with unit column stride and 16-byte aligned rows")
    K, M = a.shape
    K2, N = b.shape
    if K != K2 or M % 8 or N % 8:
        raise ValueError(f"gemm_tn: a {tuple(a.shape)} b {tuple(b.shape)}: K must match, M and N multiples of 8")
    ws_n = lib.mgx_gemm_tn_workspace_floats(M, N, K, None)
    ws = torch.empty(ws_n, dtype=torch.float32, device=a.device)
    if out is not None:
        if out.shape != (M, N) or not out.is_contiguous() or not out.is_cuda or out.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("gemm_tn: out must be a contiguous [M,N] fp32 / bf16 GPU tensor")
        c = out
    else:
        c = torch.empty(M, N, dtype=out_dtype, device=a.device)
    with _timed(tag, 2.0 * M * N * K):
        check(lib.mgx_gemm_tn_bf16(_p(a), _p(b), _p(c), _p(ws), M, N, K, M, b.stride(0), _dt(c), _stream()),
              "mgx_gemm_tn_bf16")
    return c


def _tn_operands(a, b, out, name):
    _need(a, name + " a", torch.bfloat16)
    if not b.is_cuda or b.dtype != torch.bfloat16 or b.dim() != 2 or b.stride(1) != 1 or b.stride(0) % 8 or \
            b.stride(0) < b.shape[1] or b.storage_offset() % 8:
        raise ValueError(f"{name}: b must be a bf16 GPU matrix with unit column stride and 16-byte aligned rows")
    K, M = a.shape
    K2, N = b.shape
    if K != K2 or M % 8 or N % 8:
        raise ValueError(f"{name}: a {tuple(a.shape)} b {tuple(b.shape)}: K must match, M and N multiples of 8")
    if out is not None and (out.shape != (M, N) or not out.is_contiguous() or not out.is_cuda or out.dtype != torch.float32):
        raise ValueError(f"{name}: out must be a contiguous [M,N] fp32 GPU tensor")
    return K, M, N


def gemm_tn_pair(a0, b0, a1, b1, out0=None, out1=None, tag="gemm_tn"):
    """(a0^T @ b0, a1^T @ b1) in one launch (mgx_gemm_tn_pair_bf16): two weight gradients over the same token rows share
    the K splits, which halves the fp32 slab traffic per gradient.  fp32 results; out0 / out1 as in gemm_tn."""
    K, M0, N0 = _tn_operands(a0, b0, out0, "gemm_tn_pair[0]")
    K1, M1, N1 = _tn_operands(a1, b1, out1, "gemm_tn_pair[1]")
    if K != K1:
        raise ValueError("gemm_tn_pair: the two problems must share K")
    ws = torch.empty(int(lib.mgx_gemm_tn_pair_workspace_floats(M0, N0, M1, N1, K)), dtype=torch.float32, device=a0.device)
    c0 = out0 if out0 is not None else torch.empty(M0, N0, dtype=torch.float32, device=a0.device)
    c1 = out1 if out1 is not None else torch.empty(M1, N1, dtype=torch.float32, device=a0.device)
    with _timed(tag, 2.0 * K * (M0 * N0 + M1 * N1)):
        check(lib.mgx_gemm_tn_pair_bf16(_p(a0), _p(b0), _p(c0), M0, N0, M0, b0.stride(0), _p(a1), _p(b1), _p(c1), M1, N1, M1,
                                        b1.stride(0), K, _p(ws), _lib.F32, _stream()), "mgx_gemm_tn_pair_bf16")
    return c0, c1


def _mat(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"gemm: {name}: musicgeneration_amd runs on the GPU only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if t.dim() != 2 or t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"gemm: {name} must be a 2-D fp32 / bf16 tensor")
    return t if t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


def gemm(a, b, trans_a=False, trans_b=False, bias=None, res=None, out_dtype=None, out=None):
    """C = op(a) @ op(b) (+ bias) (+ res) on mgx_gemm_generic: the fp32 parity mode and every shape the tuned bf16
    kernels do not take (any sizes and row strides; fp32 accumulation on the fp32 MFMA; deterministic).
    trans_a: a is stored [K, M]; trans_b: b is stored [N, K].  bias fp32 [N]; res [M, N] may be `out` (in place)."""
    a, b = _mat(a, "a"), _mat(b, "b")
    K, M = a.shape if trans_a else a.shape[::-1]
    N, Kb = b.shape if trans_b else b.shape[::-1]
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ ({K} and {Kb})")
    if out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
    elif out.shape != (M, N) or out.stride(1) != 1 or not out.is_cuda or out.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("gemm: out must be an [M, N] fp32 / bf16 GPU tensor with unit column stride")
    if bias is not None:
        if bias.dtype != torch.float32:
            bias = bias.float()
        _need(bias, "bias", torch.float32)
        if bias.numel() != N:
            raise ValueError("gemm: bias length")
    if res is not None:
        res = res if res is out else _mat(res, "res")
        if res.shape != (M, N):
            raise ValueError("gemm: res shape")
    if M == 0 or N == 0:
        return out
    if K == 0:
        raise ValueError("gemm: empty inner dimension")
    dt = lambda t: _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32
    check(lib.mgx_gemm_generic(a.data_ptr(), b.data_ptr(), out.data_ptr(), None if bias is None else _p(bias),
                               None if res is None else res.data_ptr(), M, N, K, a.stride(0), b.stride(0), out.stride(0),
                               0 if res is None else res.stride(0), int(trans_a), int(trans_b), dt(a), dt(b), dt(out),
                               dt(out) if res is None else dt(res), _stream()), "mgx_gemm_generic")
    return out


class _Linear(torch.autograd.Function):
    """y = x @ W (+ b) with W stored [K, N] (Conv1d, base_layers.py:54-57) or [N, K] (nn.Linear / a tied embedding
    table: alibi_unified.py:137, relative_transformer.py:116-119) on ops.gemm; the three GEMMs of its backward too."""

    @staticmethod
    def forward(ctx, x2, w, bias, w_is_nk):
        ctx.save_for_backward(x2, w)
        ctx.w_is_nk = w_is_nk
        ctx.has_bias = bias is not None
        return gemm(x2, w, trans_b=w_is_nk, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy = _mat(dy, "dy")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy, w, trans_b=not ctx.w_is_nk, out_dtype=x2.dtype)
        if ctx.needs_input_grad[1]:
            dw = gemm(dy, x2, trans_a=True, out_dtype=w.dtype) if ctx.w_is_nk else gemm(x2, dy, trans_a=True, out_dtype=w.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0, dtype=torch.float32)
        return dx, dw, db, None


def linear(x, w, bias=None, w_is_nk=True):
    """x [..., K] -> [..., N]: F.linear (w [N, K]) or Conv1d's x @ w + b (w [K, N], w_is_nk=False), differentiable,
    on the generic libmgx GEMM.  bias fp32 (or any float dtype: read as fp32)."""
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), w, None if bias is None else bias.float(), w_is_nk)
    return y.view(*lead, y.shape[-1])


def grad_slot(master, shape):
    """Where a weight gradient should be written: the parameter's slot of its data-parallel gradient bucket
    (data_parallel.GradientBucketer tags every parameter with `mgx_grad_slot`), viewed as `shape`, when the parameter has
    no gradient yet (autograd then adopts the returned tensor as p.grad, and the bucket needs no copy); None otherwise
    (no bucketer, or a gradient is being accumulated: autograd would add the slot to itself).  A slot is handed out
    at most once per backward: a parameter that feeds two producer nodes of one graph (shared weights, a block applied
    twice) gets it for the first and None for the others, whose gradients autograd then adds; the bucketer's
    post-accumulate hook releases the claim."""
    slot = getattr(master, "mgx_grad_slot", None)
    if slot is None or master.grad is not None or slot.dtype != torch.float32 or getattr(master, "mgx_slot_claimed", False):
        return None
    master.mgx_slot_claimed = True
    return slot.view(shape)


# ---------------------------------------------------------------------------
# softmax cross-entropy
# ---------------------------------------------------------------------------
_XENT_SUMS = lib.mgx_xent_sums_len()

# Data parallelism (data_parallel.GradientBucketer installs it): maps this rank's CE denominators (sum of target
# weights; fp32 tensor) to the value every rank divides by, so that per-rank losses average to the loss of the
# concatenated batch.  None: single process, the local denominator is the normaliser.
LOSS_DENOMINATOR_HOOK = None


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, weight, ignore_index, col0, V, den_hook):
        _need(logits, "logits")
        ld = logits.shape[-1]
        T = logits.numel() // ld
        tgt = targets.reshape(-1)
        if tgt.dtype != torch.int64:
            tgt = tgt.to(torch.int64)
        tgt = tgt.contiguous()
        _need(tgt, "targets", torch.int64)
        if tgt.numel() != T:
            raise ValueError(f"targets have {tgt.numel()} entries for {T} logit rows")
        if weight is not None:
            _need(weight, "weight", torch.float32)
            if weight.numel() != V:
                raise ValueError("class weight size mismatch")
        dev = logits.device
        row_loss = torch.empty(T, dtype=torch.float32, device=dev)
        row_lse = torch.empty(T, dtype=torch.float32, device=dev)
        sums = torch.empty(_XENT_SUMS, dtype=torch.float32, device=dev)
        ii = -100 if ignore_index is None else int(ignore_index)
        check(lib.mgx_softmax_xent_fwd(_p(logits), _p(tgt), _p(weight), _p(row_loss), _p(row_lse), _p(sums), T, V, ld,
                                       col0, ii, _dt(logits), _stream()), "mgx_softmax_xent_fwd")
        if den_hook is not None:
            sums[1:2].copy_(den_hook(sums[1:2]))                   # the backward kernel divides by sums[1] too
        ctx.save_for_backward(logits, tgt, weight, row_lse, sums)
        ctx.meta = (T, V, ld, col0, ii)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, gloss):
        logits, tgt, weight, row_lse, sums = ctx.saved_tensors
        T, V, ld, col0, ii = ctx.meta
        g = gloss.reshape(1).to(torch.float32).contiguous()
        # other column ranges of a shared logits buffer belong to other heads: start from zero
        if V == ld:
            dlogits = torch.empty_like(logits)
        elif col0 == 0 and ld - V < 64:
            dlogits = torch.empty_like(logits)           # one head with padding columns (ops._PaddedHead): zero only those
            dlogits.view(-1, ld)[:, V:].zero_()
        else:
            dlogits = torch.zeros_like(logits)
        check(lib.mgx_softmax_xent_bwd(_p(logits), _p(tgt), _p(weight), _p(row_lse), _p(sums), _p(g), _p(dlogits), T, V,
                                       ld, col0, ii, _dt(logits), _stream()), "mgx_softmax_xent_bwd")
        return dlogits, None, None, None, None, None, None


def _denominator_hook():
    # Read outside the autograd node (grad mode is always off inside Function.forward).  Evaluations under
    # torch.no_grad() - a validation loss, possibly on one rank only - keep the rank-local normaliser and exchange
    # nothing; with gradients enabled every rank must evaluate the criterion the same number of times.
    return LOSS_DENOMINATOR_HOOK if torch.is_grad_enabled() else None


def softmax_cross_entropy(logits, targets, weight=None, ignore_index=None, col0=0, V=None):
    """Mean (weighted) CE of logits[..., col0:col0+V] against targets; fp32 scalar."""
    V = logits.shape[-1] - col0 if V is None else V
    return _SoftmaxXent.apply(logits, targets, weight, ignore_index, col0, V, _denominator_hook())


class _SoftmaxXentMulti(torch.autograd.Function):
    """Several heads that share one logits buffer (alibi_separated.py:189-198: notes | velocities | durations in
    column ranges of the packed projection): one autograd node, ONE gradient buffer that every head's backward
    kernel fills in its own column range (three zero-filled [T, ld] buffers and their sum otherwise)."""

    @staticmethod
    def forward(ctx, logits, specs, den_hook, *tensors):
        # specs: tuple of (ignore_index, col0, V); tensors: (targets_k, weight_k or None) per head
        _need(logits, "logits")
        ld = logits.shape[-1]
        T = logits.numel() // ld
        dev = logits.device
        saved = []
        for k, (ii, col0, V) in enumerate(specs):
            tgt, weight = tensors[2 * k], tensors[2 * k + 1]
            tgt = tgt.reshape(-1)
            if tgt.dtype != torch.int64:
                tgt = tgt.to(torch.int64)
            tgt = tgt.contiguous()
            if tgt.numel() != T:
                raise ValueError(f"targets have {tgt.numel()} entries for {T} logit rows")
            if weight is not None:
                _need(weight, "weight", torch.float32)
                if weight.numel() != V:
                    raise ValueError("class weight size mismatch")
            row_loss = torch.empty(T, dtype=torch.float32, device=dev)
            row_lse = torch.empty(T, dtype=torch.float32, device=dev)
            sums = torch.empty(_XENT_SUMS, dtype=torch.float32, device=dev)
            iik = -100 if ii is None else int(ii)
            check(lib.mgx_softmax_xent_fwd(_p(logits), _p(tgt), _p(weight), _p(row_loss), _p(row_lse), _p(sums), T, V, ld,
                                           col0, iik, _dt(logits), _stream()), "mgx_softmax_xent_fwd")
            saved += [tgt, weight, row_lse, sums]
        all_sums = saved[3::4]
        if den_hook is not None:                                    # one exchange for every head's denominator
            den = den_hook(torch.stack([t[1] for t in all_sums]))
            for k, t in enumerate(all_sums):
                t[1:2].copy_(den[k:k + 1])
        losses = [t[0] / t[1] for t in all_sums]
        ctx.save_for_backward(logits, *saved)
        ctx.specs = specs
        return tuple(losses)

    @staticmethod
    def backward(ctx, *glosses):
        logits, *saved = ctx.saved_tensors
        ld = logits.shape[-1]
        T = logits.numel() // ld
        dlogits = torch.empty_like(logits)
        flat = dlogits.view(-1, ld)
        covered = sorted((col0, col0 + V) for _, col0, V in ctx.specs)
        at = 0
        for lo, hi in covered + [(ld, ld)]:            # columns no head owns (padding) carry no gradient
            if lo > at:
                flat[:, at:lo].zero_()
            at = max(at, hi)
        for k, (ii, col0, V) in enumerate(ctx.specs):
            tgt, weight, row_lse, sums = saved[4 * k:4 * k + 4]
            g = glosses[k].reshape(1).to(torch.float32).contiguous()
            iik = -100 if ii is None else int(ii)
            check(lib.mgx_softmax_xent_bwd(_p(logits), _p(tgt), _p(weight), _p(row_lse), _p(sums), _p(g), _p(dlogits), T, V,
                                           ld, col0, iik, _dt(logits), _stream()), "mgx_softmax_xent_bwd")
        return (dlogits, None, None) + (None,) * (2 * len(ctx.specs))


def softmax_cross_entropy_multi(logits, heads):
    """heads: sequence of (targets, weight | None, ignore_index | None, col0, V).  Returns one fp32 loss per head."""
    specs = tuple((ii, int(col0), int(V)) for _, _, ii, col0, V in heads)
    tensors = []
    for tgt, w, _, _, _ in heads:
        tensors += [tgt, w]
    return _SoftmaxXentMulti.apply(logits, specs, _denominator_hook(), *tensors)


# ---------------------------------------------------------------------------
# token embedding with a GEMM backward (alibi_unified.py:130-131)
# ---------------------------------------------------------------------------
def _onehot_grad(tokens, dh, V, tag):
    """dTable [V, d] fp32 = onehot(tokens)^T . dh on gemm_tn_kernel: the one-hot matrix [T, Vp] is bf16 zeros and
    ones, the products are exact, the sums fp32 in the kernel's fixed order.  dh may be a column slice (row stride >
    its width) of the gradient of a concatenated embedding."""
    T = tokens.numel()
    Vp = (V + 63) // 64 * 64
    onehot = torch.zeros(T, Vp, dtype=torch.bfloat16, device=dh.device)
    onehot.scatter_(1, tokens.reshape(T, 1).to(torch.int64), 1.0)
    return gemm_tn(onehot, dh, tag=tag)[:V]


class _EmbedBF16(torch.autograd.Function):
    """bf16 rows of an fp32 table.  Backward: a one-hot GEMM (_onehot_grad) instead of the library's cast + sort +
    segmented reduction (~0.2 ms per step for a 618-row table)."""

    @staticmethod
    def forward(ctx, tokens, table):
        ctx.save_for_backward(tokens)
        V, d = table.shape
        ctx.V = V
        tok = tokens.reshape(-1)
        if tok.dtype != torch.int64:
            tok = tok.to(torch.int64)
        tok = tok.contiguous()
        out = torch.empty(*tokens.shape, d, dtype=torch.bfloat16, device=table.device)
        check(lib.mgx_embed_rows_bf16(_p(tok), _p(table.detach()), V, d, tok.numel(), _p(out), _stream()), "mgx_embed_rows_bf16")
        return out

    @staticmethod
    def backward(ctx, dh):
        (tokens,) = ctx.saved_tensors
        d = dh.shape[-1]
        return None, _onehot_grad(tokens, dh.reshape(tokens.numel(), d).contiguous(), ctx.V, "gemm_tn_embed")


def embed_bf16_supported(tokens, table):
    return (table.is_cuda and tokens.is_cuda and table.is_contiguous() and table.dtype == torch.float32
            and table.shape[1] % 8 == 0 and table.shape[0] <= 4096
            and torch.is_grad_enabled() and table.requires_grad)


def embed_bf16(tokens, table):
    return _EmbedBF16.apply(tokens, table)


class _EmbedCatBF16(torch.autograd.Function):
    """Three token streams x [B,3,S] -> concat(embed_k(x[:, k])) [B,S,d_n+d_v+d_d] bf16 in ONE launch
    (alibi_separated.py:135-139: three gathers, a concatenation and the autocast cast); backward: one one-hot GEMM
    per table on the matching column slice of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, *tables):
        B, n, S = x.shape
        tok = x if x.dtype == torch.int64 else x.to(torch.int64)
        tok = tok.contiguous()
        widths = [t.shape[1] for t in tables]
        out = torch.empty(B, S, sum(widths), dtype=torch.bfloat16, device=x.device)
        import ctypes as C
        tabs = (C.c_void_p * n)(*[t.detach().data_ptr() for t in tables])
        vs = (C.c_int * n)(*[t.shape[0] for t in tables])
        ws = (C.c_int * n)(*widths)
        check(lib.mgx_embed_cat_bf16(_p(tok), n, B, S, tabs, vs, ws, _p(out), _stream()), "mgx_embed_cat_bf16")
        ctx.save_for_backward(tok)
        ctx.meta = ([t.shape[0] for t in tables], widths)
        return out

    @staticmethod
    def backward(ctx, dh):
        (tok,) = ctx.saved_tensors
        vs, widths = ctx.meta
        B, n, S = tok.shape
        dh2 = dh.reshape(B * S, sum(widths))
        if not dh2.is_contiguous():
            dh2 = dh2.contiguous()
        grads, c0 = [], 0
        for k in range(n):
            grads.append(_onehot_grad(tok[:, k], dh2[:, c0:c0 + widths[k]], vs[k], "gemm_tn_embed"))
            c0 += widths[k]
        return (None, *grads)


def embed_cat_bf16_supported(x, tables):
    return (x.is_cuda and x.dim() == 3 and x.shape[1] == len(tables) <= 4 and torch.is_grad_enabled()
            and all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.requires_grad
                    and t.shape[1] % 8 == 0 and t.shape[0] <= 4096 for t in tables))


def embed_cat_bf16(x, tables):
    return _EmbedCatBF16.apply(x, *tables)


# ---------------------------------------------------------------------------
# output projection with a padded class dimension (alibi_unified.py:136-137)
# ---------------------------------------------------------------------------
class _PaddedHead(torch.autograd.Function):
    """logits_pad [T, Vp] = h [T, d] @ w_pad[Vp, d]^T + b_pad, where w_pad stacks the rows of one or more output
    projections (alibi_unified.py:136-137: 618 classes; alibi_separated.py:151-154: 92 + 130 + 203) and pads them
    with zero rows to a multiple of 64: none of these vocabularies is a multiple of 8, which would keep all three
    GEMMs of the head off the MFMA kernels.  Padded columns of the incoming gradient are zero (the CE kernels write
    only their own columns)."""

    @staticmethod
    def forward(ctx, h, w_pad, w_pad_t, b_pad, sizes, *masters):
        ctx.save_for_backward(h, w_pad_t)
        ctx.sizes = sizes
        return gemm_nt(h, w_pad, b_pad)

    @staticmethod
    def backward(ctx, dp):
        h, w_pad_t = ctx.saved_tensors
        dp = dp.contiguous()
        dw = gemm_tn(dp, h, tag="gemm_tn_head")              # [Vp, d] fp32
        db = colsum(dp) if ctx.needs_input_grad[6] else None # no bias: the tied heads of old_models/*
        dh = gemm_nt(dp, w_pad_t)                            # dp [T, Vp] @ w_pad [Vp, d]
        grads, r0 = [], 0
        for V in ctx.sizes:                                  # rows of each stacked projection
            grads += [dw[r0:r0 + V], None if db is None else db[r0:r0 + V]]
            r0 += V
        return (dh, None, None, None, None, *grads)


def padded_head_supported(h, w):
    return h.dtype == torch.bfloat16 and h.is_cuda and w.dtype == torch.float32 and h.shape[-1] % 64 == 0


def padded_head(h, w_pad, w_pad_t, b_pad, masters):
    """h [..., d] bf16; masters: (weight_0 [V_0, d], bias_0 [V_0], weight_1, bias_1, ...) fp32 autograd leaves;
    zero-padded stacked shadows w_pad [Vp, d] bf16, w_pad_t [d, Vp] bf16, b_pad [Vp] fp32.  Returns the padded
    logits [..., Vp]; the caller slices its heads out of it (the criterion goes back to the padded base)."""
    lead = h.shape[:-1]
    sizes = tuple(int(w.shape[0]) for w in masters[0::2])
    out = _PaddedHead.apply(h.reshape(-1, h.shape[-1]), w_pad, w_pad_t, b_pad, sizes, *masters)
    return out.view(*lead, out.shape[-1])


# ---------------------------------------------------------------------------
# whole post-LN decoder block as one autograd node
# ---------------------------------------------------------------------------
def _mfma_gemms(x2, d, ff):
    """bf16 blocks whose widths the tuned MFMA GEMMs take (K a multiple of 64 at every projection site) run all twelve
    GEMMs of the block (four forward, four dgrad, four weight gradients) on gemm_nt / gemm_tn; fp32 (the 1e-5 parity
    mode) and odd widths run the same twelve on the generic libmgx GEMM (ops.gemm)."""
    return x2.dtype == torch.bfloat16 and d % 64 == 0 and ff % 64 == 0


class _DecoderBlock(torch.autograd.Function):
    """alibi_unified.py:86-95 / relative_transformer.py:77-86.

    Inputs: x [B,S,d] in the compute dtype, the block's BlockWeights (compute-dtype shadows in
    both orientations) and the fp32 master parameters (autograd leaves).  bf16: projections run on
    the libmgx MFMA GEMMs (bias / residual / activation-backward epilogues, fp32 weight gradients);
    fp32 (parity mode) and shapes the kernels do not take use plain library GEMMs.
    """

    @staticmethod
    def forward(ctx, x, cfg, slopes, E_master, E, W, *masters):
        n_heads, act, p_drop, eps, training, rope = cfg
        B, S, d = x.shape
        T = B * S
        x2 = x.reshape(T, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ff = W.m_w1.shape[0]
        p = p_drop if training else 0.0
        seed1 = new_seed() if p > 0 else 0
        seed2 = new_seed() if p > 0 else 0
        mfma = _mfma_gemms(x2, d, ff)
        if mfma:
            qkv = gemm_nt(x2, W.wqkv_t, W.m_bqkv).view(B, S, 3 * d)
        else:
            qkv = gemm(x2, W.wqkv, bias=W.m_bqkv).view(B, S, 3 * d)
        if rope is not None:                      # rotary variant: rotate q | k in place, then plain causal attention
            rope_qk_(qkv, rope[0], rope[1], n_heads)
        if E is None:
            a, lse = alibi_attn_fwd(qkv, slopes, n_heads)
        else:
            a, lse = rel_attn_fwd(qkv, E, n_heads)
        a2 = a.view(T, d)
        pr = gemm_nt(a2, W.wp_t, W.m_bp) if mfma else gemm(a2, W.wp, bias=W.m_bp)
        h1, z1, mean1, rstd1 = ln_residual_fwd(pr, x2, W.m_g1, W.m_be1, eps, p, seed1)
        if mfma:
            # bias + activation + dropout in the up-projection's epilogue (two outputs: the activated tensor for
            # linear2 and, in the slot of the pre-activation, the derivative factor act'(pre) * mask/(1-p) the
            # backward multiplies by).
            pre, hact = gemm_nt(h1, W.w1, W.m_b1, act=act, p_drop=p, seed=seed2)
            v = gemm_nt(hact, W.w2, W.m_b2)
        else:
            u = gemm(h1, W.w1, trans_b=True)
            pre, hact = bias_act_fwd(u, W.m_b1, act, p, seed2)
            v = gemm(hact, W.w2, trans_b=True, bias=W.m_b2)
        h2, z2, mean2, rstd2 = ln_residual_fwd(v, h1, W.m_g2, W.m_be2, eps)
        ctx.save_for_backward(x2, qkv, a, lse, z1, mean1, rstd1, h1, pre, hact, z2, mean2, rstd2, slopes, E)
        ctx.W = W
        ctx.cfg = (n_heads, act, p, seed1, seed2, B, S, d, mfma, rope)
        return h2.view(B, S, d)

    @staticmethod
    def backward(ctx, dh2):
        x2, qkv, a, lse, z1, mean1, rstd1, h1, pre, hact, z2, mean2, rstd2, slopes, E = ctx.saved_tensors
        W = ctx.W
        n_heads, act, p, seed1, seed2, B, S, d, mfma, rope = ctx.cfg
        T = B * S
        f32 = torch.float32
        dh2 = dh2.reshape(T, d).contiguous()
        dz2, _, dg2, dbe2, db2 = ln_residual_bwd(dh2, None, z2, mean2, rstd2, W.m_g2, want_dsum=True)
        if mfma:
            dpre, db1 = gemm_nt_dfac(dz2, W.w2_t, pre, act, p)                          # dz2 @ w2 through act' (+ db1)
            # the two weight gradients of the MLP in one launch: [d, ff] and [ff, d]
            dw2, dw1 = gemm_tn_pair(dz2, hact, dpre, h1, grad_slot(W.m_w2, (d, hact.shape[1])),
                                    grad_slot(W.m_w1, (dpre.shape[1], d)))
            dh1 = gemm_nt(dpre, W.w1_t)                                                 # [T, d]
        else:
            dw2 = gemm(dz2, hact, trans_a=True, out_dtype=f32)
            dpre, db1 = bias_act_bwd(gemm(dz2, W.w2), pre, act, p, seed2)
            dw1 = gemm(dpre, h1, trans_a=True, out_dtype=f32)
            dh1 = gemm(dpre, W.w1)
        dz1, da1, dg1, dbe1, dbp = ln_residual_bwd(dh1, dz2, z1, mean1, rstd1, W.m_g1, p, seed1, want_dsum=True)
        a2 = a.view(T, d)
        if mfma:
            da = gemm_nt(da1, W.wp).view(B, S, d)                                       # da1 @ wp^T
        else:
            dwp = gemm(a2, da1, trans_a=True, out_dtype=f32)
            da = gemm(da1, W.wp, trans_b=True).view(B, S, d)
        if E is None:
            dqkv = alibi_attn_bwd(da, qkv, a, lse, slopes, n_heads)
            dE = None
        else:
            dqkv, dE = rel_attn_bwd(da, qkv, a, lse, E, n_heads)
        if rope is not None:
            rope_qk_(dqkv, rope[0], rope[1], n_heads, inverse=True)
        dqkv2 = dqkv.view(T, 3 * d)
        dbqkv = colsum(dqkv2) if dqkv2.dtype == torch.bfloat16 and (3 * d) % 8 == 0 else dqkv2.sum(0, dtype=f32)
        if mfma:
            # the two weight gradients of the attention projections in one launch: [d, d] and [d, 3d]
            dwp, dwqkv = gemm_tn_pair(a2, da1, x2, dqkv2, grad_slot(W.m_wp, (d, d)), grad_slot(W.m_wqkv, (d, 3 * d)))
            dx = gemm_nt(dqkv2, W.wqkv, None, dz1).view(B, S, d)                        # dqkv @ wqkv^T + dz1
        else:
            dwqkv = gemm(x2, dqkv2, trans_a=True, out_dtype=f32)
            # in place (res aliases out).  dz1 (and da1, which aliases it without dropout) has no reader after this point
            dx = gemm(dqkv2, W.wqkv, trans_b=True, res=dz1, out=dz1).view(B, S, d)
        return (dx, None, None, dE, None, None,
                dwqkv.unsqueeze(0), dbqkv, dwp.unsqueeze(0), dbp, dw1, db1, dw2, db2, dg1, dbe1, dg2, dbe2)


def fused_decoder_block(x, n_heads, act, p_drop, eps, training, slopes, E_master, E, W, rope=None):
    """W: base_layers.BlockWeights of the block (masters = c_attn.w, c_attn.b, c_proj.w, c_proj.b,
    linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight,
    norm2.bias)."""
    cfg = (n_heads, act, float(p_drop), float(eps), bool(training), rope)       # rope: (cos, sin) [S, dh/2] fp32 or None
    return _DecoderBlock.apply(x, cfg, slopes, E_master, E, W, *W.masters)


# ---------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------
def _dev_pos(pos0, like):
    if pos0.dtype != torch.int32 or pos0.numel() != 1 or pos0.device != like.device:
        raise ValueError("a device-side position must be one int32 on the tensors' GPU")
    return _p(pos0)


def kv_append(qkv, kcache, vcache, pos0):
    """qkv [B,R,3*d]; caches [B,H,Lmax,dh].  pos0: int, or a 1-element int32 GPU tensor read at
    execution time (HIP-graph replay)."""
    _need(qkv, "qkv")
    _need(kcache, "kcache", qkv.dtype)
    _need(vcache, "vcache", qkv.dtype)
    B, R, d3 = qkv.shape
    Bc, H, Lmax, dh = kcache.shape
    if Bc != B or 3 * H * dh != d3 or vcache.shape != kcache.shape:
        raise ValueError("kv_append shape mismatch")
    if isinstance(pos0, torch.Tensor):
        check(lib.mgx_kv_append_at(_p(qkv), _p(kcache), _p(vcache), B, R, H, dh, Lmax, _dev_pos(pos0, qkv), _dt(qkv),
                                   _stream()), "mgx_kv_append_at")
        return
    check(lib.mgx_kv_append(_p(qkv), _p(kcache), _p(vcache), B, R, H, dh, Lmax, pos0, _dt(qkv), _stream()),
          "mgx_kv_append")


def decode_attn(q, kcache, vcache, slopes, pos0):
    """q [B,R,H*dh] rows at absolute positions pos0..pos0+R-1 -> out [B,R,H*dh] (pos0 as in kv_append)."""
    _need(q, "q")
    _need(kcache, "kcache", q.dtype)
    _need(vcache, "vcache", q.dtype)
    B, R, d = q.shape
    Bc, H, Lmax, dh = kcache.shape
    if Bc != B or H * dh != d or vcache.shape != kcache.shape:
        raise ValueError("decode_attn shape mismatch")
    out = torch.empty_like(q)
    if isinstance(pos0, torch.Tensor):
        check(lib.mgx_decode_attn_at(_p(q), _p(kcache), _p(vcache), _p(slopes), _p(out), B, R, H, dh, Lmax,
                                     _dev_pos(pos0, q), _dt(q), _stream()), "mgx_decode_attn_at")
        return out
    with _timed("decode_attn"):
        check(lib.mgx_decode_attn(_p(q), _p(kcache), _p(vcache), _p(slopes), _p(out), B, R, H, dh, Lmax, pos0, _dt(q),
                                  _stream()), "mgx_decode_attn")
    return out


def decode_step_attn_supported(dtype, dh: int) -> bool:
    return dtype == torch.bfloat16 and dh in (64, 128)


def decode_step_attn(qkv, kcache, vcache, slopes, pos0):
    """kv_append + decode_attn in one launch: qkv [B,R,3*d] (R <= 4) -> out [B,R,d]; caches updated at pos0.."""
    _need(qkv, "qkv", torch.bfloat16)
    _need(kcache, "kcache", qkv.dtype)
    _need(vcache, "vcache", qkv.dtype)
    B, R, d3 = qkv.shape
    Bc, H, Lmax, dh = kcache.shape
    if Bc != B or 3 * H * dh != d3 or vcache.shape != kcache.shape:
        raise ValueError("decode_step_attn shape mismatch")
    out = torch.empty(B, R, H * dh, dtype=qkv.dtype, device=qkv.device)
    if isinstance(pos0, torch.Tensor):
        if KERNEL_EVENTS is not None and not torch.cuda.is_current_stream_capturing():
            with _timed("decode_attn"):                  # launch-by-launch pass of bench.py (events cannot be captured)
                check(lib.mgx_decode_step_attn(_p(qkv), _p(kcache), _p(vcache), _p(slopes), _p(out), B, R, H, dh, Lmax,
                                               0, _dev_pos(pos0, qkv), _dt(qkv), _stream()), "mgx_decode_step_attn")
            return out
        check(lib.mgx_decode_step_attn(_p(qkv), _p(kcache), _p(vcache), _p(slopes), _p(out), B, R, H, dh, Lmax, 0,
                                       _dev_pos(pos0, qkv), _dt(qkv), _stream()), "mgx_decode_step_attn")
        return out
    with _timed("decode_attn"):
        check(lib.mgx_decode_step_attn(_p(qkv), _p(kcache), _p(vcache), _p(slopes), _p(out), B, R, H, dh, Lmax, pos0,
                                       None, _dt(qkv), _stream()), "mgx_decode_step_attn")
    return out


_SKINNY_MODE = {None: 0, "gelu": 1, "relu": 2, "ln": 3}


def skinny_gemm_supported(x: torch.Tensor, wt: torch.Tensor) -> bool:
    """Shapes mgx_skinny_gemm_bf16 takes: bf16, at most 256 rows, N % 64 == 0, K % 64 == 0."""
    return (x.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and x.dim() == 2 and wt.dim() == 2
            and 0 < x.shape[0] <= 256 and wt.shape[0] % 64 == 0 and x.shape[1] % 64 == 0
            and x.is_contiguous() and wt.is_contiguous())


def skinny_workspace_floats(M: int, N: int, K: int) -> int:
    return int(lib.mgx_skinny_gemm_workspace_floats(M, N, K))


def skinny_gemm(x, wt, bias, workspace, act=None, res=None, gamma=None, beta=None, eps=1e-5):
    """Decode projection: epilogue(x[M,K] @ wt[N,K]^T + bias).  act: None | 'gelu' | 'relu'; with `res`:
    LayerNorm(. + res) * gamma + beta.  `workspace`: fp32 tensor of >= skinny_workspace_floats(M, N, K) elements."""
    _need(x, "x", torch.bfloat16)
    _need(wt, "wt", torch.bfloat16)
    _need(bias, "bias", torch.float32)
    _need(workspace, "workspace", torch.float32)
    M, K = x.shape
    N, K2 = wt.shape
    if K != K2 or bias.numel() != N:
        raise ValueError(f"skinny_gemm: x {tuple(x.shape)} wt {tuple(wt.shape)} bias {bias.numel()}")
    mode = _SKINNY_MODE["ln" if res is not None else act]
    if res is not None:
        _need(res, "res", torch.bfloat16)
        _need(gamma, "gamma", torch.float32)
        _need(beta, "beta", torch.float32)
        if res.shape != (M, N) or gamma.numel() != N or beta.numel() != N or act is not None:
            raise ValueError("skinny_gemm: residual/LayerNorm operands do not match (and exclude an activation)")
    if workspace.numel() < skinny_workspace_floats(M, N, K):
        raise ValueError("skinny_gemm: workspace too small")
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    check(lib.mgx_skinny_gemm_bf16(_p(x), _p(wt), _p(bias), _p(res), _p(gamma), _p(beta), float(eps), _p(out),
                                   _p(workspace), M, N, K, mode, _stream()), "mgx_skinny_gemm_bf16")
    return out


# ---------------------------------------------------------------------------
# optimizer
# ---------------------------------------------------------------------------
def adamw_step(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _need(param, "param", torch.float32)
    _need(grad, "grad")
    _need(exp_avg, "exp_avg", torch.float32)
    _need(exp_avg_sq, "exp_avg_sq", torch.float32)
    n = param.numel()
    if grad.numel() != n or exp_avg.numel() != n or exp_avg_sq.numel() != n:
        raise ValueError("adamw size mismatch")
    if shadow is not None:
        _need(shadow, "shadow", torch.bfloat16)
        if shadow.numel() != n:
            raise ValueError("shadow size mismatch")
    check(lib.mgx_adamw(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), n, lr, beta1, beta2, eps,
                        weight_decay, step, grad_scale, _dt(grad), _stream()), "mgx_adamw")


def _seq_args(seq):
    if seq.dim() != 2 or seq.dtype not in (torch.int32, torch.int64):
        raise TypeError("seq must be [B, L] int32 or int64")
    _need(seq, "seq")
    return _p(seq), int(seq.dtype == torch.int64), seq.shape[1]


def _streams_args(seq):
    """seq [B, L] (one stream) or [B, n, L]: (pointer, is_int64, L, B, n)."""
    if seq.dim() not in (2, 3) or seq.dtype not in (torch.int32, torch.int64):
        raise TypeError("seq must be [B, L] or [B, n, L], int32 or int64")
    _need(seq, "seq")
    n = 1 if seq.dim() == 2 else seq.shape[1]
    if not 1 <= n <= 4:
        raise ValueError("at most 4 token streams")
    return _p(seq), int(seq.dtype == torch.int64), seq.shape[-1], seq.shape[0], n


def sample_gather_embed(seq, pos, tables, R=3):
    """bf16 embedding rows of seq[:, (k,) pos .. pos+R-1], the streams' rows concatenated (mgx_sample_gather_embed)."""
    import ctypes as C
    sp, i64, L, B, n = _streams_args(seq)
    if len(tables) != n:
        raise ValueError(f"{len(tables)} tables for {n} token streams")
    for t in tables:
        _need(t, "table", torch.float32)
    D = sum(t.shape[1] for t in tables)
    out = torch.empty(B, R, D, dtype=torch.bfloat16, device=seq.device)
    tp = (C.c_void_p * n)(*[t.data_ptr() for t in tables])
    Vs = (C.c_int * n)(*[t.shape[0] for t in tables])
    ds = (C.c_int * n)(*[t.shape[1] for t in tables])
    check(lib.mgx_sample_gather_embed(sp, i64, L, _dev_pos(pos, seq), n, tp, Vs, ds, B, R, _p(out), _stream()),
          "mgx_sample_gather_embed")
    return out


def sample_head(h, gamma, beta, eps, w, bias, sizes, keys, pos=None, stochastic=False, seed=0, logits=None):
    """Final Norm + the stacked output heads (class counts `sizes`) of row R-1 of h [B,R,d] bf16; keys [B, n] (int64
    storage of the kernel's uint64 keys, zero on entry) receive per head the running argmax, or - stochastic - a
    Gumbel-max draw from softmax(logits) keyed by (seed, *pos, b, class)."""
    import ctypes as C
    _need(h, "h", torch.bfloat16)
    _need(w, "w", torch.bfloat16)
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (bias, "bias")):
        _need(t, nm, torch.float32)
    _need(keys, "keys", torch.int64)
    B, R, d = h.shape
    n, V = len(sizes), sum(sizes)
    if w.shape != (V, d) or bias.numel() != V or gamma.numel() != d or beta.numel() != d or keys.numel() != B * n:
        raise ValueError("sample_head: shape mismatch")
    if logits is not None:
        _need(logits, "logits", torch.float32)
        if tuple(logits.shape) != (B, V):
            raise ValueError("sample_head: logits must be [B, V]")
    if stochastic and pos is None:
        raise ValueError("sample_head: a stochastic draw needs the device-side position (its counter)")
    Vs = (C.c_int * n)(*[int(v) for v in sizes])
    check(lib.mgx_sample_head(_p(h), B, R, d, _p(gamma), _p(beta), eps, _p(w), _p(bias), n, Vs, int(bool(stochastic)),
                              int(seed) & (2 ** 64 - 1), None if pos is None else _p(pos), _p(keys),
                              None if logits is None else _p(logits), _stream()), "mgx_sample_head")


def sample_commit(seq, pos, keys):
    """seq[:, (k,) pos+1] = class held in keys[b, k]; keys = 0; pos += 1 (mgx_sample_commit)."""
    sp, i64, L, B, n = _streams_args(seq)
    _need(keys, "keys", torch.int64)
    if keys.numel() != B * n:
        raise ValueError("sample_commit: keys must hold one entry per sequence and stream")
    check(lib.mgx_sample_commit(sp, i64, L, B, n, _dev_pos(pos, seq), _p(keys), _stream()), "mgx_sample_commit")


def decode_gather_embed(seq, pos, table, R=3):
    """bf16 embedding rows of seq[:, pos .. pos+R-1] (mgx_decode_gather_embed); pos: one int32 on the device."""
    sp, i64, L = _seq_args(seq)
    _need(table, "table", torch.float32)
    V, d = table.shape
    out = torch.empty(seq.shape[0], R, d, dtype=torch.bfloat16, device=seq.device)
    check(lib.mgx_decode_gather_embed(sp, i64, L, _dev_pos(pos, seq), _p(table), V, d, seq.shape[0], R, _p(out), _stream()),
          "mgx_decode_gather_embed")
    return out


def decode_head_argmax(h, gamma, beta, eps, w, bias, keys, logits=None):
    """Final Norm + output projection + running argmax of row R-1 of h [B,R,d] bf16 into keys [B] (int64 storage of
    the kernel's uint64 keys, zero on entry); logits: optional fp32 [B,V] receiver."""
    _need(h, "h", torch.bfloat16)
    _need(w, "w", torch.bfloat16)
    for t, n in ((gamma, "gamma"), (beta, "beta"), (bias, "bias")):
        _need(t, n, torch.float32)
    _need(keys, "keys", torch.int64)
    B, R, d = h.shape
    V = w.shape[0]
    if w.shape[1] != d or bias.numel() != V or gamma.numel() != d or beta.numel() != d or keys.numel() != B:
        raise ValueError("decode_head_argmax: shape mismatch")
    if logits is not None:
        _need(logits, "logits", torch.float32)
        if tuple(logits.shape) != (B, V):
            raise ValueError("decode_head_argmax: logits must be [B, V]")
    check(lib.mgx_decode_head_argmax(_p(h), B, R, d, _p(gamma), _p(beta), eps, _p(w), _p(bias), V, _p(keys),
                                     None if logits is None else _p(logits), _stream()), "mgx_decode_head_argmax")


def decode_commit(seq, pos, keys):
    """seq[:, pos+1] = argmax class held in keys; keys = 0; pos += 1 (mgx_decode_commit)."""
    sp, i64, L = _seq_args(seq)
    _need(keys, "keys", torch.int64)
    check(lib.mgx_decode_commit(sp, i64, L, seq.shape[0], _dev_pos(pos, seq), _p(keys), _stream()), "mgx_decode_commit")


def colsum(x):
    """fp32 column sums of a bf16 matrix [T, n] (mgx_colsum_bf16), fixed summation order."""
    _need(x, "x", torch.bfloat16)
    T, n = x.shape
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.mgx_colsum_workspace_floats(n)), dtype=torch.float32, device=x.device)
    check(lib.mgx_colsum_bf16(_p(x), _p(out), _p(ws), T, n, _stream()), "mgx_colsum_bf16")
    return out


def cast_shadow(srcs, dsts, dsts_t):
    """bf16 shadows of fp32 matrices in one launch (mgx_cast_shadow_bf16): srcs[i] [R,C] fp32 -> dsts[i] [R,C] bf16
    and dsts_t[i] [C,R] bf16 (either may be None).  R and C must be multiples of 64."""
    import ctypes as C
    n = len(srcs)
    if not (len(dsts) == len(dsts_t) == n) or n == 0:
        raise ValueError("cast_shadow: list lengths differ or are empty")
    for s, d, t in zip(srcs, dsts, dsts_t):
        _need(s, "src", torch.float32)
        if s.dim() != 2:
            raise ValueError("cast_shadow: sources must be 2-D")
        if d is None and t is None:
            raise ValueError("cast_shadow: a matrix without destination")
        if d is not None:
            _need(d, "dst", torch.bfloat16)
            if tuple(d.shape) != tuple(s.shape):
                raise ValueError(f"cast_shadow: dst {tuple(d.shape)} for src {tuple(s.shape)}")
        if t is not None:
            _need(t, "dst_t", torch.bfloat16)
            if tuple(t.shape) != (s.shape[1], s.shape[0]):
                raise ValueError(f"cast_shadow: dst_t {tuple(t.shape)} for src {tuple(s.shape)}")
    arr = lambda ts: (C.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    rows = (C.c_int * n)(*[s.shape[0] for s in srcs])
    cols = (C.c_int * n)(*[s.shape[1] for s in srcs])
    check(lib.mgx_cast_shadow_bf16(n, arr(srcs), arr(dsts), arr(dsts_t), rows, cols, _stream()), "mgx_cast_shadow_bf16")


def adamw_step_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """One fused AdamW update over a list of fp32 tensors (mgx_adamw_multi): 1 launch per 32 tensors."""
    import ctypes as C
    n = len(params)
    if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == n) or n == 0:
        raise ValueError("adamw_step_multi: list lengths differ or are empty")
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        _need(p, "param", torch.float32)
        _need(g, "grad", torch.float32)
        _need(m, "exp_avg", torch.float32)
        _need(v, "exp_avg_sq", torch.float32)
        if not (g.numel() == m.numel() == v.numel() == p.numel()):
            raise ValueError("adamw_step_multi: size mismatch")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (C.c_long * n)(*[p.numel() for p in params])
    with _timed("adamw", 28.0 * sum(p.numel() for p in params)):       # p, m, v read + written, g read: 28 B per parameter
        check(lib.mgx_adamw_multi(n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), sizes, lr, beta1, beta2, eps,
                                  weight_decay, step, grad_scale, _stream()), "mgx_adamw_multi")
