"""Three-stream (note / velocity / duration) ALiBi decoder — the reference's
model_baseline/alibi_separated.py API and state_dict keys on the libmgx kernels.

The trunk is the unified model's (same DecoderBlock); the ends differ: three
embedding gathers concatenated on the feature axis (alibi_separated.py:135-139),
three linear heads (:151-154) and a three-term criterion with class weights
(:170-198).  The heads run as ONE [d, Vn+Vv+Vd] GEMM whose output the fused
softmax-CE kernel reads by column range.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..base_layers import Norm, ShadowCache, active_dtype, resolve_dtype
from ..metrics import SamplewiseAccuracy
from .alibi_unified import (DecoderBlock, SimpleAttention, _activation_name, _drop_mask_key, construct_alibi,
                            get_slopes, tqdm)


class SeparatedLogits(tuple):
    """(note, velocity, duration) logits: a plain tuple for every consumer
    (recursive_cast, unpacking), which also remembers the packed [B,S,Vn+Vv+Vd]
    buffer so the criterion can run without re-packing."""

    def __new__(cls, packed, sizes):
        n, v, d = sizes
        views = (packed[..., :n], packed[..., n:n + v], packed[..., n + v:n + v + d])
        self = super().__new__(cls, views)
        self.packed = packed
        self.sizes = sizes
        return self


class AlibiTransformer(nn.Module):
    def __init__(self, vocab_note, vocab_velocity, vocab_duration, embed_note, embed_velocity, embed_duration,
                 n_layers, n_heads, max_sequence, proj_forward, dropout=0.1, activation="gelu", dtype=None,
                 **kwargs) -> None:
        super().__init__()
        self.vocab_note = vocab_note
        self.vocab_velocity = vocab_velocity
        self.vocab_duration = vocab_duration
        self.dim_note = embed_note
        self.dim_velocity = embed_velocity
        self.dim_duration = embed_duration
        self.d_model = embed_note + embed_velocity + embed_duration
        self.n_layers = n_layers
        self.heads = n_heads
        self.max_sequence = max_sequence
        self.compute_dtype = resolve_dtype(dtype)

        self.embed_note = nn.Embedding(vocab_note, embed_note)
        self.embed_velocity = nn.Embedding(vocab_velocity, embed_velocity)
        self.embed_duration = nn.Embedding(vocab_duration, embed_duration)
        act = _activation_name(activation)
        self.decoder_layers = nn.ModuleList(
            [DecoderBlock(n_heads, self.d_model, proj_forward, act, dropout) for _ in range(n_layers)])
        self.norm = Norm(self.d_model)
        self.proj_note = nn.Linear(self.d_model, vocab_note)
        self.proj_velocity = nn.Linear(self.d_model, vocab_velocity)
        self.proj_duration = nn.Linear(self.d_model, vocab_duration)
        self._shadow = ShadowCache()
        self._register_load_state_dict_pre_hook(_drop_mask_key)

    def reference_state_dict(self):
        sd = dict(self.state_dict())
        sd["mask"] = construct_alibi(self.max_sequence, self.heads)
        return sd

    def embed_tokens(self, x, dtype):
        """x [B,3,S] -> [B,S,d_model]: three gathers, concatenated (alibi_separated.py:135-139)."""
        tables = (self.embed_note.weight, self.embed_velocity.weight, self.embed_duration.weight)
        if dtype == torch.bfloat16 and ops.embed_cat_bf16_supported(x, tables):
            return ops.embed_cat_bf16(x, tables)             # one launch; the table gradients are one-hot GEMMs
        h = torch.cat([F.embedding(x[:, k], t) for k, t in enumerate(tables)], dim=-1)
        return h.to(dtype)

    def head_weights(self, dt):
        """The three output projections stacked into one [92+130+203, d] GEMM operand, in dtype dt."""
        w = torch.cat([self.proj_note.weight, self.proj_velocity.weight, self.proj_duration.weight], 0).to(dt)
        b = torch.cat([self.proj_note.bias, self.proj_velocity.bias, self.proj_duration.bias], 0).to(dt)
        return w, b

    def head(self, h, wb=None):
        sizes = (self.vocab_note, self.vocab_velocity, self.vocab_duration)
        if wb is None and ops.padded_head_supported(h, self.proj_note.weight):
            # one [448, d] GEMM for the three heads: stacked, zero-padded shadows kept per parameter version
            ws = (self.proj_note.weight, self.proj_velocity.weight, self.proj_duration.weight)
            bs = (self.proj_note.bias, self.proj_velocity.bias, self.proj_duration.bias)
            vp = (sum(sizes) + 63) // 64 * 64
            w_pad = self._shadow.padded_stack(ws, h.dtype, vp)
            w_pad_t = self._shadow.padded_stack(ws, h.dtype, vp, transposed=True)
            b_pad = self._shadow.padded_stack(bs, torch.float32, vp)
            packed = ops.padded_head(h, w_pad, w_pad_t, b_pad, (ws[0], bs[0], ws[1], bs[1], ws[2], bs[2]))
            return SeparatedLogits(packed, sizes)
        w, b = wb if wb is not None else self.head_weights(h.dtype)
        return SeparatedLogits(ops.linear(h, w, b), sizes)     # fp32 / odd widths: mgx_gemm_generic

    def forward(self, x: torch.Tensor):
        if x.dim() != 3 or x.shape[1] != 3:
            raise ValueError(f"expected [batch, 3, seq] token ids, got {tuple(x.shape)}")
        dt = active_dtype(self.compute_dtype)
        with torch.autocast("cuda", enabled=False):
            h = self.embed_tokens(x, dt)
            for layer in self.decoder_layers:
                h = layer(h)
            h = self.norm(h)
            return self.head(h)

    def inference(self, seq: torch.Tensor, idx=0, deterministic=False):
        """alibi_separated.py:156-166 (last-position logits; flag inverted as in the reference)."""
        with torch.no_grad():
            for i in tqdm(range(idx, seq.shape[-1])):
                heads = self(seq)
                if deterministic:
                    r = [torch.multinomial(h[:, -1].float().softmax(-1), num_samples=1).squeeze(-1) for h in heads]
                else:
                    r = [torch.argmax(h[:, -1], -1) for h in heads]
                seq[:, :, i] = torch.stack(r, -1).to(seq.dtype)
        return seq


class Criterion(nn.Module):
    """Notes and velocities: class-weighted CE (w[1]=0, w[2]=0.05); durations:
    ignore_index=1; weighted sum by loss_weights (alibi_separated.py:170-198)."""

    def __init__(self, vocab_n=92, vocab_v=130, vocab_d=203,
                 loss_weights={"loss_notes": 1, "loss_velocity": 1, "loss_duration": 1}) -> None:
        super().__init__()
        self.loss_weights = loss_weights
        w_n = torch.ones(vocab_n)
        w_n[1] = 0
        w_n[2] = 0.05
        self.register_buffer("weight_n", w_n)
        w_v = torch.ones(vocab_v)
        w_v[1] = 0
        w_v[2] = 0.05
        self.register_buffer("weight_v", w_v)

    def forward(self, xs, y):
        if isinstance(xs, SeparatedLogits):
            packed = xs.packed if xs.packed.is_contiguous() else xs.packed.contiguous()
            n_, v_, d_ = xs.sizes
            # one autograd node for the three heads of the packed buffer: one gradient buffer, filled by column range
            ln, lv, ld = ops.softmax_cross_entropy_multi(packed, (
                (y[:, 0], self.weight_n, None, 0, n_), (y[:, 1], self.weight_v, None, n_, v_),
                (y[:, 2], None, 1, n_ + v_, d_)))
            losses = {"loss_notes": ln, "loss_velocity": lv, "loss_duration": ld}
        else:
            specs = (("loss_notes", self.weight_n, None), ("loss_velocity", self.weight_v, None),
                     ("loss_duration", None, 1))
            losses = {name: ops.softmax_cross_entropy(xs[k].contiguous(), y[:, k], w, ii)
                      for k, (name, w, ii) in enumerate(specs)}
        loss = sum(self.loss_weights[k] * losses[k] for k in ("loss_notes", "loss_velocity", "loss_duration"))
        losses["loss"] = loss
        return loss, losses


class Metrics(nn.Module):
    """alibi_separated.py:201-230 without torchmetrics."""
    full_state_update = False

    def __init__(self) -> None:
        super().__init__()
        self.acc_n = SamplewiseAccuracy()
        self.acc_v = SamplewiseAccuracy()
        self.acc_d = SamplewiseAccuracy()
        self.total_acc = SamplewiseAccuracy()

    def update(self, xs, y):
        n, v, d = xs
        self.acc_n.update(n, y[:, 0])
        self.acc_v.update(v, y[:, 1])
        self.acc_d.update(d, y[:, 2])
        self.total_acc.update(torch.stack([n, v, d], 1), y)

    def compute(self):
        return {"acc_notes": self.acc_n.compute(), "acc_velocity": self.acc_v.compute(),
                "acc_duration": self.acc_d.compute(), "accuracy": self.total_acc.compute()}

    def reset(self):
        for m in (self.acc_n, self.acc_v, self.acc_d, self.total_acc):
            m.reset()


def build_model_and_criterion(args):
    """alibi_separated.py:234-236."""
    c = Criterion(args.vocab_note, args.vocab_velocity, args.vocab_duration, args.loss_weights)
    return AlibiTransformer(**args), c
