"""Single-stream ALiBi decoder LM over the 618-token unified MIDI vocabulary —
same classes, constructor kwargs and state_dict keys as the reference's
model_baseline/alibi_unified.py, computing through libmgx (HIP, gfx950).

What differs from the reference on purpose:
  * no [H,S,S] mask buffer: the bias slope_h*(j-i) + causal mask live inside the
    attention kernels (alibi_unified.py:26-31,121-122 materialise 32 MiB..512 MiB);
    a ``mask`` key in a loaded state_dict is accepted and dropped, and
    ``reference_state_dict()`` re-creates it for checkpoints that must load into
    the reference;
  * a whole DecoderBlock is one autograd node (ops.fused_decoder_block);
  * extra constructor kwarg ``dtype`` ('fp32' | 'bf16'); the reference's ctor
    swallows unknown kwargs (alibi_unified.py:104), so configs stay portable;
  * ``S > max_sequence`` works (the reference casts the rebuilt mask to int64 there,
    alibi_unified.py:129 — a bug nobody can rely on).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.mgx.*)
from ..base_layers import Attention, BlockWeights, Conv1d, Norm, ShadowCache, active_dtype, resolve_dtype, \
    generate_square_subsequent_mask
from ..metrics import SamplewiseAccuracy

try:                                    # progress bars only
    from tqdm import tqdm
except Exception:                       # pragma: no cover
    def tqdm(it):
        return it


def get_slopes(n):
    """ALiBi slopes (alibi_unified.py:13-23): geometric 2^(-8i/n) for powers of
    two, otherwise the power of two below plus every other slope of the one above."""
    def pow2_slopes(k):
        first = 2 ** (-(2 ** -(math.log2(k) - 3)))
        return [first * first ** i for i in range(k)]

    if math.log2(n).is_integer():
        return pow2_slopes(n)
    low = 2 ** math.floor(math.log2(n))
    return pow2_slopes(low) + get_slopes(2 * low)[0::2][:n - low]


def construct_alibi(max_seq_len, attn_heads):
    """[H,S,S] fp32 slope_h*j + causal (alibi_unified.py:26-31).  Only used to
    emit reference-compatible checkpoints; the forward never calls it."""
    slopes = torch.tensor(get_slopes(attn_heads), dtype=torch.float32)
    ramp = torch.arange(max_seq_len, dtype=torch.float32)
    bias = (slopes[:, None, None] * ramp[None, None, :]).expand(attn_heads, max_seq_len, max_seq_len)
    return bias + generate_square_subsequent_mask(max_seq_len, "cpu")[None]


class SimpleAttention(Attention):
    """c_attn -> causal ALiBi attention -> c_proj (alibi_unified.py:34-67)."""

    def __init__(self, heads, n_state):
        super().__init__(heads, n_state)
        self.c_attn = Conv1d(n_state * 3, n_state)
        self.c_proj = Conv1d(n_state, n_state)
        self.register_buffer("slopes", torch.tensor(get_slopes(heads), dtype=torch.float32), persistent=False)

    def forward(self, x, mask=None):
        """Stand-alone use (the block fuses this path).  `mask` is ignored: the
        kernel applies the ALiBi + causal bias itself."""
        qkv = self.c_attn(x)
        a = torch.ops.mgx.alibi_attention(qkv.contiguous(), self.slopes, self.heads)[0]     # torch_ops.py
        return self.c_proj(a)


class DecoderBlock(nn.Module):
    """Post-LN block (alibi_unified.py:70-95) as one fused autograd node."""

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
        self.activation = activation            # 'gelu' | 'relu'
        self._shadow = ShadowCache()

    def masters(self):
        a = self.attn
        return (a.c_attn.w, a.c_attn.b, a.c_proj.w, a.c_proj.b, self.linear1.weight, self.linear1.bias,
                self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias)

    def weights(self, dtype):
        return BlockWeights(self._shadow, self.masters(), self.n_state, dtype)

    def forward(self, x, mask=None):
        return ops.fused_decoder_block(x, self.heads, self.activation, self.drop.p, self.norm1.eps, self.training,
                                       self.attn.slopes, None, None, self.weights(x.dtype))


def _activation_name(activation):
    if isinstance(activation, str):
        name = activation
    else:
        name = getattr(activation, "__name__", str(activation))
    if name not in ("gelu", "relu"):
        raise ValueError(f"activation {activation!r} is not supported by the fused kernels (gelu, relu)")
    return name


class AlibiTransformer(nn.Module):
    def __init__(self, vocab=618, embed_dim=512, n_layers=6, n_heads=8, max_sequence=1024, proj_forward=4096,
                 dropout=0.1, activation="gelu", dtype=None, **kwargs) -> None:
        super().__init__()
        self.vocab = vocab
        self.d_model = embed_dim
        self.n_layers = n_layers
        self.heads = n_heads
        self.max_sequence = max_sequence
        self.compute_dtype = resolve_dtype(dtype)

        self.embed = nn.Embedding(vocab, embed_dim)
        act = _activation_name(activation)
        self.decoder_layers = nn.ModuleList(
            [DecoderBlock(n_heads, embed_dim, proj_forward, act, dropout) for _ in range(n_layers)])
        self.norm = Norm(embed_dim)
        self.proj_class = nn.Linear(embed_dim, vocab)
        self._shadow = ShadowCache()
        self._register_load_state_dict_pre_hook(_drop_mask_key)

    # -- reference-compatible checkpoints ------------------------------------
    def reference_state_dict(self):
        """state_dict plus the persistent ``mask`` buffer the reference registers (alibi_unified.py:121-122)."""
        sd = dict(self.state_dict())
        sd["mask"] = construct_alibi(self.max_sequence, self.heads)
        return sd

    # -- forward ---------------------------------------------------------------
    def embed_tokens(self, x, dtype):
        if dtype == torch.bfloat16 and ops.embed_bf16_supported(x, self.embed.weight):
            return ops.embed_bf16(x, self.embed.weight)              # same forward; the table gradient is a GEMM
        return F.embedding(x, self.embed.weight).to(dtype)

    def head_weights(self, dt):
        """(weight, bias) of the output projection in dtype dt (a cast copy unless fp32)."""
        if dt == torch.float32:
            return self.proj_class.weight, self.proj_class.bias
        return self.proj_class.weight.to(dt), self.proj_class.bias.to(dt)

    def head(self, h, wb=None):
        if wb is None and ops.padded_head_supported(h, self.proj_class.weight):
            # class dimension padded to a multiple of 64 inside (all three GEMMs of the head then run on the MFMA
            # kernels); the caller sees the [.., :vocab] view, the criterion goes back to the padded base
            vp = (self.vocab + 63) // 64 * 64
            w, b = self.proj_class.weight, self.proj_class.bias
            w_pad = self._shadow.padded_stack((w,), h.dtype, vp)
            w_pad_t = self._shadow.padded_stack((w,), h.dtype, vp, transposed=True)
            b_pad = self._shadow.padded_stack((b,), torch.float32, vp)
            padded = ops.padded_head(h, w_pad, w_pad_t, b_pad, (w, b))
            logits = padded[..., :self.vocab]
            logits.mgx_padded_base = padded          # explicit carrier for Criterion (no copy, padded gradient)
            return logits
        w, b = wb if wb is not None else self.head_weights(h.dtype)
        return ops.linear(h, w, b)                             # fp32 / odd widths: mgx_gemm_generic

    def forward(self, x: torch.Tensor):
        """x [B,S] int32/int64 -> logits [B,S,vocab] in the compute dtype (alibi_unified.py:124-138)."""
        if x.dim() != 2:
            raise ValueError(f"expected [batch, seq] token ids, got {tuple(x.shape)}")
        dt = active_dtype(self.compute_dtype)
        with torch.autocast("cuda", enabled=False):
            h = self.embed_tokens(x, dt)
            for layer in self.decoder_layers:
                h = layer(h)
            h = self.norm(h)
            return self.head(h)

    def inference(self, seq: torch.Tensor, idx=0, deterministic=False):
        """alibi_unified.py:140-149, flag inversion included: ``deterministic=True``
        samples from the multinomial, False takes the argmax of the LAST position."""
        with torch.no_grad():
            for i in tqdm(range(idx, seq.shape[-1])):
                l = self(seq)[:, -1].float()
                if deterministic:
                    r = torch.multinomial(l.softmax(-1), num_samples=1).squeeze(-1)
                else:
                    r = torch.argmax(l, -1)
                seq[:, i] = r
        return seq


def _drop_mask_key(state_dict, prefix, *args):
    state_dict.pop(prefix + "mask", None)


class Criterion(nn.Module):
    """Mean CE over non-pad targets (alibi_unified.py:153-163) on the fused softmax-CE kernel."""

    def __init__(self, pad_index=None) -> None:
        super().__init__()
        self.ignore_index = pad_index

    def forward(self, xs, y):
        base = getattr(xs, "mgx_padded_base", None)
        if (base is not None and base.is_contiguous() and base.shape[:-1] == xs.shape[:-1]
                and base.data_ptr() == xs.data_ptr() and xs.stride(-1) == 1 and xs.shape[-1] <= base.shape[-1]):
            # the model's own [.., :vocab] view of its padded logits: the kernel takes the padded buffer with its row
            # stride (no contiguous copy, and the gradient arrives padded, as the head's backward wants it)
            loss = ops.softmax_cross_entropy(base, y, None, self.ignore_index, col0=0, V=xs.shape[-1])
        else:
            logits = xs if xs.is_contiguous() else xs.contiguous()
            loss = ops.softmax_cross_entropy(logits, y, None, self.ignore_index)
        losses = {"loss": loss}
        return losses["loss"], losses


class Metrics(nn.Module):
    """Accuracy on arg-maxed predictions (alibi_unified.py:166-183) without torchmetrics."""
    full_state_update = False

    def __init__(self) -> None:
        super().__init__()
        self.acc_n = SamplewiseAccuracy()

    def update(self, xs, y):
        self.acc_n.update(xs, y)

    def compute(self):
        return {"accuracy": self.acc_n.compute()}

    def reset(self):
        self.acc_n.reset()


def build_model_and_criterion(args):
    """(model, criterion) from an EasyDict of hyper-parameters (alibi_unified.py:186-188)."""
    c = Criterion(args.pad_index)
    return AlibiTransformer(**args), c
