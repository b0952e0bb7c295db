"""Music-Transformer-style relative-position decoder — the reference's
old_models/relative_transformer.py API and state_dict keys on libmgx.

The skew trick (pad one column, reshape, drop a row: relative_transformer.py:18-32)
is never executed: the attention kernel evaluates rel[i,j] = q_i . E[h, max_seq-1-(i-j)]
in place, so neither the [B,H,S,S] `rel` tensor nor the scores reach HBM.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.mgx.*)
from ..base_layers import Attention, BlockWeights, Conv1d, Norm, ShadowCache, active_dtype, resolve_dtype, \
    tied_embed, tied_head, \
    generate_square_subsequent_mask
from ..model_baseline.alibi_unified import _activation_name


class RelativeAttention(Attention):
    def __init__(self, heads, n_state, max_sequence):
        super().__init__(heads, n_state)
        self.max_sequence = max_sequence
        self.c_attn = Conv1d(n_state * 3, n_state)
        self.c_proj = Conv1d(n_state, n_state)
        self.E = nn.Parameter(torch.empty(heads, max_sequence, n_state // heads))
        nn.init.xavier_normal_(self.E)

    def forward(self, x, mask=None):
        """Stand-alone path; `mask` ignored (causal inside the kernel)."""
        qkv = self.c_attn(x)
        a = torch.ops.mgx.rel_attention(qkv.contiguous(), self.E, self.heads)[0]    # torch_ops.py: fp32 dE reaches the master
        return self.c_proj(a)


class DecoderBlock(nn.Module):
    """relative_transformer.py:61-86."""

    def __init__(self, heads, max_sequence, n_state, proj_forward, activation, dropout):
        super().__init__()
        self.heads = heads
        self.max_sequence = max_sequence
        self.n_state = n_state
        self.attn = RelativeAttention(heads, n_state, max_sequence)
        self.linear1 = nn.Linear(n_state, proj_forward)
        self.linear2 = nn.Linear(proj_forward, n_state)
        self.norm1 = nn.LayerNorm(n_state)
        self.norm2 = nn.LayerNorm(n_state)
        self.drop = nn.Dropout(dropout)
        self.activation = activation
        self._shadow = ShadowCache()

    def forward(self, x, mask=None):
        a, dt = self.attn, x.dtype
        masters = (a.c_attn.w, a.c_attn.b, a.c_proj.w, a.c_proj.b, self.linear1.weight, self.linear1.bias,
                   self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                   self.norm2.weight, self.norm2.bias)
        W = BlockWeights(self._shadow, masters, self.n_state, dt)
        return ops.fused_decoder_block(x, self.heads, self.activation, self.drop.p, self.norm1.eps, self.training,
                                       None, a.E, self._shadow.get(a.E, dt), W)


class RelativeTransformer(nn.Module):
    def __init__(self, n_vocab, d_model, n_layers, n_heads, max_sequence, proj_forward, dropout=0.1,
                 activation=F.relu, dtype=None) -> None:
        super().__init__()
        self.n_vocab = n_vocab
        self.d_model = d_model
        self.n_layers = n_layers
        self.heads = n_heads
        self.max_sequence = max_sequence
        self.compute_dtype = resolve_dtype(dtype)
        self.embedding = nn.Parameter(torch.normal(size=(n_vocab, d_model), mean=0.0, std=0.02))
        act = _activation_name(activation)
        self.decoder_layers = nn.ModuleList(
            [DecoderBlock(n_heads, max_sequence, d_model, proj_forward, act, dropout) for _ in range(n_layers)])
        self.norm = Norm(d_model)
        self._shadow = ShadowCache()
        self._register_load_state_dict_pre_hook(lambda sd, prefix, *a: sd.pop(prefix + "mask", None))

    def reference_state_dict(self):
        sd = dict(self.state_dict())
        sd["mask"] = generate_square_subsequent_mask(self.max_sequence, "cpu")
        return sd

    def forward(self, x: torch.Tensor):
        """x [B,S] -> logits [B,S,n_vocab]; tied input/output embedding, no head bias
        (relative_transformer.py:107-121).  Requires S <= max_sequence, as the reference does."""
        batch, seq_len = x.shape
        if seq_len > self.max_sequence:
            raise ValueError(f"sequence length {seq_len} exceeds max_sequence {self.max_sequence}")
        dt = active_dtype(self.compute_dtype)
        with torch.autocast("cuda", enabled=False):
            h = tied_embed(x, self.embedding, dt)
            for layer in self.decoder_layers:
                h = layer(h)
            h = self.norm(h)
            return tied_head(h, self.embedding, self._shadow)
