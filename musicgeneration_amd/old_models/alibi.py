"""Earlier ALiBi model of the reference (old_models/alibi.py): the same post-LN blocks as
model_baseline/alibi_unified.py, but a raw ``embedding`` parameter N(0, 0.02) that is also the output
projection (alibi.py:107,127-129), no head bias.  Runs on the same fused blocks and flash kernels."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..base_layers import Norm, ShadowCache, active_dtype, resolve_dtype, tied_embed, tied_head
from ..model_baseline.alibi_unified import DecoderBlock, _activation_name, _drop_mask_key, construct_alibi, get_slopes  # noqa: F401


class AlibiTransformer(nn.Module):
    def __init__(self, n_vocab, embed_dim, d_model, n_layers, n_heads, max_sequence, proj_forward, dropout=0.1,
                 activation=F.gelu, dtype=None) -> None:
        super().__init__()
        if embed_dim != d_model:
            raise ValueError("embed_dim must equal d_model (the reference's forward only works then: alibi.py:127)")
        self.n_vocab, self.embed_dim, self.d_model = n_vocab, embed_dim, d_model
        self.n_layers, self.heads, self.max_sequence = n_layers, n_heads, max_sequence
        self.compute_dtype = resolve_dtype(dtype)
        self.embedding = nn.Parameter(torch.normal(size=(n_vocab, embed_dim), mean=0.0, std=0.02))
        act = _activation_name(activation)
        self.decoder_layers = nn.ModuleList(
            [DecoderBlock(n_heads, d_model, proj_forward, act, dropout) for _ in range(n_layers)])
        self.norm = Norm(d_model)
        self._shadow = ShadowCache()
        self._register_load_state_dict_pre_hook(_drop_mask_key)

    def reference_state_dict(self):
        sd = dict(self.state_dict())
        sd["mask"] = construct_alibi(self.max_sequence, self.heads)
        return sd

    def forward(self, x: torch.Tensor):
        """x [B,S] -> logits [B,S,n_vocab] (alibi.py:114-130)."""
        batch, seq_len = x.shape
        dt = active_dtype(self.compute_dtype)
        with torch.autocast("cuda", enabled=False):
            h = tied_embed(x, self.embedding, dt)
            for layer in self.decoder_layers:
                h = layer(h)
            h = self.norm(h)
            return tied_head(h, self.embedding, self._shadow)


def build(args):
    return AlibiTransformer(**args)
