"""Rotary-position decoder — the reference's old_models/rotary_transformer.py API and state_dict keys on
libmgx.  q and k of the packed projection are rotated in place by mgx_rope_qk (rotate_every_two /
apply_rotary_pos_emb, rotary_transformer.py:9-21), then the flash kernels run with zero ALiBi slopes, i.e.
plain causal attention; the backward un-rotates dq | dk with the transposed rotation."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..base_layers import Attention, BlockWeights, Conv1d, Norm, ShadowCache, active_dtype, resolve_dtype, \
    tied_embed, tied_head, \
    generate_square_subsequent_mask
from ..model_baseline.alibi_unified import _activation_name


class FixedPositionalEmbedding(nn.Module):
    """rotary_transformer.py:24-42: buffer ``emb`` [max_seq, dim] = sin | cos of position x inv_freq."""

    def __init__(self, dim, max_seq_len):
        super().__init__()
        self.dim = dim
        self.inv_freq = 1. / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("emb", self._table(max_seq_len))

    def _table(self, n):
        position = torch.arange(0, n, dtype=torch.float)
        sinusoid_inp = torch.einsum("i,j->ij", position, self.inv_freq)
        return torch.cat((sinusoid_inp.sin(), sinusoid_inp.cos()), dim=-1)

    def calculate_new_embed(self, x):
        return self._table(x.shape[1]).unsqueeze(0).to(x)

    def forward(self, x):
        return self.emb[None, :x.shape[1], :].to(x)

    def tables(self, seq_len, device, as_reference=True):
        """(cos, sin) [seq_len, dim/2] fp32 on `device` for the kernel.

        as_reference: RotaryTransformer.forward passes the int64 token tensor to this module and forward() ends in
        ``.to(x)`` (rotary_transformer.py:41-42,133), so the reference multiplies q and k by the table TRUNCATED to
        integers (cos = 1 only at angle 0, sin = 0).  Parity with the reference means reproducing that."""
        emb = self.emb[:seq_len] if seq_len <= self.emb.shape[0] else self._table(seq_len)
        if as_reference:
            emb = emb.to(torch.int64)
        emb = emb.to(device=device, dtype=torch.float32)
        half = self.dim // 2
        return emb[:, half:].contiguous(), emb[:, :half].contiguous()


class SimpleAttention(Attention):
    def __init__(self, heads, n_state):
        super().__init__(heads, n_state)
        self.c_attn = Conv1d(n_state * 3, n_state)
        self.c_proj = Conv1d(n_state, n_state)
        self.register_buffer("slopes", torch.zeros(heads, dtype=torch.float32), persistent=False)   # no ALiBi bias


class DecoderBlock(nn.Module):
    """rotary_transformer.py:84-108 as one fused autograd node."""

    def __init__(self, heads, n_state, proj_forward, activation, dropout):
        super().__init__()
        self.heads = heads
        self.n_state = n_state
        self.attn = SimpleAttention(heads, n_state)
        self.linear1 = nn.Linear(n_state, proj_forward)
        self.linear2 = nn.Linear(proj_forward, n_state)
        self.norm1 = nn.LayerNorm(n_state)
        self.norm2 = nn.LayerNorm(n_state)
        self.drop = nn.Dropout(dropout)
        self.activation = activation
        self._shadow = ShadowCache()

    def forward(self, x, mask=None, pos=None):
        """pos = (cos, sin) tables [S, dh/2] fp32 (FixedPositionalEmbedding.tables); `mask` is ignored."""
        a = self.attn
        masters = (a.c_attn.w, a.c_attn.b, a.c_proj.w, a.c_proj.b, self.linear1.weight, self.linear1.bias,
                   self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                   self.norm2.weight, self.norm2.bias)
        W = BlockWeights(self._shadow, masters, self.n_state, x.dtype)
        return ops.fused_decoder_block(x, self.heads, self.activation, self.drop.p, self.norm1.eps, self.training,
                                       a.slopes, None, None, W, rope=pos)


class RotaryTransformer(nn.Module):
    def __init__(self, n_vocab, d_model, n_layers, n_heads, max_sequence, proj_forward, dropout=0.1,
                 activation=F.relu, dtype=None, intended_rotation=False) -> None:
        """intended_rotation=True applies the real sin/cos rotation instead of the reference's integer-truncated
        table (see FixedPositionalEmbedding.tables); the default reproduces the reference."""
        super().__init__()
        self.intended_rotation = intended_rotation
        self.n_vocab, self.d_model, self.n_layers = n_vocab, d_model, n_layers
        self.heads, self.max_sequence = n_heads, max_sequence
        self.compute_dtype = resolve_dtype(dtype)
        self.embedding = nn.Parameter(torch.normal(size=(n_vocab, d_model), mean=0.0, std=0.02))
        act = _activation_name(activation)
        self.decoder_layers = nn.ModuleList(
            [DecoderBlock(n_heads, d_model, proj_forward, act, dropout) for _ in range(n_layers)])
        self.norm = Norm(d_model)
        self._shadow = ShadowCache()
        self.pos_embed = FixedPositionalEmbedding(d_model // n_heads, max_sequence)
        self._register_load_state_dict_pre_hook(lambda sd, prefix, *a: sd.pop(prefix + "mask", None))

    def reference_state_dict(self):
        sd = dict(self.state_dict())
        sd["mask"] = generate_square_subsequent_mask(self.max_sequence, "cpu")
        return sd

    def forward(self, x: torch.Tensor):
        """x [B,S] -> logits [B,S,n_vocab]; tied embedding (rotary_transformer.py:125-144)."""
        batch, seq_len = x.shape
        dt = active_dtype(self.compute_dtype)
        with torch.autocast("cuda", enabled=False):
            pos = self.pos_embed.tables(seq_len, x.device, as_reference=not self.intended_rotation)
            h = tied_embed(x, self.embedding, dt)
            for layer in self.decoder_layers:
                h = layer(h, None, pos)
            h = self.norm(h)
            return tied_head(h, self.embedding, self._shadow)
