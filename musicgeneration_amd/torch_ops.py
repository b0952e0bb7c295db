"""torch.library registration of the libmgx operators: ``torch.ops.mgx.*`` over the same C ABI (include/mgx.h).

The wrappers in ``ops.py`` stay the implementation (shape checks, raw pointers, the current HIP stream); this module
gives the operators a dispatcher identity: a schema, a fake (meta) kernel so they trace under FakeTensor /
torch.compile, and ``register_autograd`` formulas for the differentiable ones.  Modules that use an operator on its
own (SimpleAttention.forward, RelativeAttention.forward, Norm, the criteria) call these; the fused decoder block
(ops.fused_decoder_block) remains one autograd node built from the same kernels.

    torch.ops.mgx.alibi_attention(qkv, slopes, n_heads) -> out          differentiable in qkv
    torch.ops.mgx.rel_attention(qkv, E, n_heads) -> out                 differentiable in qkv and E
    torch.ops.mgx.layer_norm(x, gamma, beta, eps) -> y                  differentiable in x, gamma, beta
    torch.ops.mgx.gemm_nt(a, bt, bias) -> c                             a @ bt^T + bias, differentiable in a, bt, bias
    torch.ops.mgx.alibi_attn_fwd / alibi_attn_bwd / gemm_tn / colsum    building blocks (no autograd formula)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_lib = torch.library


# ---- building blocks ------------------------------------------------------------------------------------------------
@_lib.custom_op("mgx::alibi_attn_fwd", mutates_args=())
def alibi_attn_fwd(qkv: Tensor, slopes: Tensor, n_heads: int) -> Tuple[Tensor, Tensor]:
    return ops.alibi_attn_fwd(qkv, slopes, n_heads)


@alibi_attn_fwd.register_fake
def _(qkv, slopes, n_heads):
    B, S, d3 = qkv.shape
    return qkv.new_empty(B, S, d3 // 3), qkv.new_empty(B, n_heads, S, dtype=torch.float32)


@_lib.custom_op("mgx::alibi_attn_bwd", mutates_args=())
def alibi_attn_bwd(dout: Tensor, qkv: Tensor, out: Tensor, lse: Tensor, slopes: Tensor, n_heads: int) -> Tensor:
    return ops.alibi_attn_bwd(dout, qkv, out, lse, slopes, n_heads)


@alibi_attn_bwd.register_fake
def _(dout, qkv, out, lse, slopes, n_heads):
    return torch.empty_like(qkv)


@_lib.custom_op("mgx::rel_attn_fwd", mutates_args=())
def rel_attn_fwd(qkv: Tensor, E: Tensor, n_heads: int) -> Tuple[Tensor, Tensor]:
    return ops.rel_attn_fwd(qkv, E, n_heads)


@rel_attn_fwd.register_fake
def _(qkv, E, n_heads):
    B, S, d3 = qkv.shape
    return qkv.new_empty(B, S, d3 // 3), qkv.new_empty(B, n_heads, S, dtype=torch.float32)


@_lib.custom_op("mgx::rel_attn_bwd", mutates_args=())
def rel_attn_bwd(dout: Tensor, qkv: Tensor, out: Tensor, lse: Tensor, E: Tensor, n_heads: int) -> Tuple[Tensor, Tensor]:
    return ops.rel_attn_bwd(dout, qkv, out, lse, E, n_heads)


@rel_attn_bwd.register_fake
def _(dout, qkv, out, lse, E, n_heads):
    return torch.empty_like(qkv), E.new_empty(E.shape, dtype=torch.float32)


@_lib.custom_op("mgx::gemm_tn", mutates_args=())
def gemm_tn(a: Tensor, b: Tensor) -> Tensor:
    return ops.gemm_tn(a, b)


@gemm_tn.register_fake
def _(a, b):
    return a.new_empty(a.shape[1], b.shape[1], dtype=torch.float32)


@_lib.custom_op("mgx::colsum", mutates_args=())
def colsum(x: Tensor) -> Tensor:
    return ops.colsum(x)


@colsum.register_fake
def _(x):
    return x.new_empty(x.shape[1], dtype=torch.float32)


# ---- differentiable operators ---------------------------------------------------------------------------------------
@_lib.custom_op("mgx::alibi_attention", mutates_args=())
def alibi_attention(qkv: Tensor, slopes: Tensor, n_heads: int) -> Tuple[Tensor, Tensor]:
    """(out, lse); lse is returned so that the backward formula can use it (it carries no gradient)."""
    return ops.alibi_attn_fwd(qkv, slopes, n_heads)


@alibi_attention.register_fake
def _(qkv, slopes, n_heads):
    B, S, d3 = qkv.shape
    return qkv.new_empty(B, S, d3 // 3), qkv.new_empty(B, n_heads, S, dtype=torch.float32)


def _alibi_setup(ctx, inputs, output):
    qkv, slopes, n_heads = inputs
    out, lse = output
    ctx.save_for_backward(qkv, out, lse, slopes)
    ctx.n_heads = n_heads
    ctx.mark_non_differentiable(lse)


def _alibi_backward(ctx, dout, _dlse):
    qkv, out, lse, slopes = ctx.saved_tensors
    return torch.ops.mgx.alibi_attn_bwd(dout.contiguous(), qkv, out, lse, slopes, ctx.n_heads), None, None


alibi_attention.register_autograd(_alibi_backward, setup_context=_alibi_setup)


@_lib.custom_op("mgx::rel_attention", mutates_args=())
def rel_attention(qkv: Tensor, E: Tensor, n_heads: int) -> Tuple[Tensor, Tensor]:
    """Skew relative attention.  E: the table in the compute dtype, or the fp32 master (cast here): the gradient the
    autograd formula returns has E's dtype, so handing in the master keeps the kernel's fp32 dE unrounded.
    Returns (out, lse)."""
    return ops.rel_attn_fwd(qkv, E if E.dtype == qkv.dtype else E.to(qkv.dtype), n_heads)


@rel_attention.register_fake
def _(qkv, E, n_heads):
    B, S, d3 = qkv.shape
    return qkv.new_empty(B, S, d3 // 3), qkv.new_empty(B, n_heads, S, dtype=torch.float32)


def _rel_setup(ctx, inputs, output):
    qkv, E, n_heads = inputs
    out, lse = output
    ctx.save_for_backward(qkv, out, lse, E)
    ctx.n_heads = n_heads
    ctx.mark_non_differentiable(lse)


def _rel_backward(ctx, dout, _dlse):
    qkv, out, lse, E = ctx.saved_tensors
    Ec = E if E.dtype == qkv.dtype else E.to(qkv.dtype)
    dqkv, dE = torch.ops.mgx.rel_attn_bwd(dout.contiguous(), qkv, out, lse, Ec, ctx.n_heads)
    return dqkv, dE.to(E.dtype), None


rel_attention.register_autograd(_rel_backward, setup_context=_rel_setup)


@_lib.custom_op("mgx::layer_norm_fwd", mutates_args=())
def layer_norm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    y, _, mean, rstd = ops.ln_residual_fwd(x.contiguous(), None, gamma, beta, eps)
    return y, mean, rstd


@layer_norm_fwd.register_fake
def _(x, gamma, beta, eps):
    T = x.numel() // x.shape[-1]
    return torch.empty_like(x), x.new_empty(T, dtype=torch.float32), x.new_empty(T, dtype=torch.float32)


@_lib.custom_op("mgx::layer_norm_bwd", mutates_args=())
def layer_norm_bwd(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    dz, _, dg, db = ops.ln_residual_bwd(dy.contiguous(), None, x, mean, rstd, gamma)
    return dz, dg, db


@layer_norm_bwd.register_fake
def _(dy, x, mean, rstd, gamma):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    _, mean, rstd = output
    ctx.save_for_backward(x.contiguous(), mean, rstd, gamma)
    ctx.mark_non_differentiable(mean, rstd)


def _ln_backward(ctx, dy, _dm, _dr):
    x, mean, rstd, gamma = ctx.saved_tensors
    dz, dg, db = torch.ops.mgx.layer_norm_bwd(dy, x, mean, rstd, gamma)
    return dz, dg, db, None


layer_norm_fwd.register_autograd(_ln_backward, setup_context=_ln_setup)


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """Final Norm (base_layers.py:36-41) through the dispatcher."""
    return torch.ops.mgx.layer_norm_fwd(x, gamma, beta, eps)[0]


@_lib.custom_op("mgx::gemm_nt", mutates_args=())
def gemm_nt(a: Tensor, bt: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """c = a[M,K] @ bt[N,K]^T (+ bias fp32 [N]) on the MFMA kernel (bf16)."""
    return ops.gemm_nt(a, bt, bias)


@gemm_nt.register_fake
def _(a, bt, bias=None):
    return a.new_empty(a.shape[0], bt.shape[0])


def _gemm_setup(ctx, inputs, output):
    a, bt, bias = inputs
    ctx.save_for_backward(a, bt)
    ctx.has_bias = bias is not None


def _gemm_backward(ctx, dc):
    a, bt = ctx.saved_tensors
    dc = dc.contiguous()
    K = a.shape[1]
    # da = dc @ bt needs bt^T as the [K, N] "Bt" operand; dbt = dc^T @ a on the TN kernel
    da = torch.ops.mgx.gemm_nt(dc, bt.t().contiguous(), None) if dc.shape[1] % 64 == 0 and K % 8 == 0 else ops.gemm(dc, bt)
    dbt = torch.ops.mgx.gemm_tn(dc, a).to(bt.dtype) if dc.shape[1] % 8 == 0 and K % 8 == 0 else ops.gemm(dc, a, trans_a=True)
    dbias = torch.ops.mgx.colsum(dc) if ctx.has_bias else None
    return da, dbt, dbias


gemm_nt.register_autograd(_gemm_backward, setup_context=_gemm_setup)
