"""CPU oracle for the MusicGeneration hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state_dict in, tensors out) fp32
restatement in plain eager PyTorch of the reference's ALiBi / relative
transformer forward, losses, training step and sampling loops.  It is the
checker for the HIP path; nothing under ``musicgeneration_amd/`` imports it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the real
reference modules from /root/reference in the build container (torchmetrics
stubbed, as SURVEY.md §8c describes), ran them on seeded inputs and committed
the inputs/outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against them.

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


# --------------------------------------------------------------------------
# ALiBi bias  (model_baseline/alibi_unified.py:13-31, base_layers.py:19-22)
# --------------------------------------------------------------------------
def alibi_slopes(n_heads: int) -> list:
    """Per-head slopes, python-float math.  alibi_unified.py:13-23.

    Power-of-two head counts give the geometric sequence 2^(-8 i / n), i=1..n;
    other counts take the sequence of the power of two below and interleave
    every other entry of the sequence of the power of two above.
    """
    def geometric(n):
        first = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [first * first ** i for i in range(n)]

    if math.log2(n_heads).is_integer():
        return geometric(n_heads)
    below = 2 ** math.floor(math.log2(n_heads))
    extra = alibi_slopes(2 * below)[0::2][: n_heads - below]
    return geometric(below) + extra


def causal_mask(size: int) -> Tensor:
    """0 on/below the diagonal, -inf above.  base_layers.py:19-22."""
    m = torch.full((size, size), float("-inf"))
    return torch.triu(m, diagonal=1)


def alibi_mask(seq_len: int, n_heads: int) -> Tensor:
    """[H,S,S] fp32: slope_h * key_index + causal.  alibi_unified.py:26-31.

    Note the bias multiplies the *key index* j (not j - i); rows are shifted
    copies as far as softmax is concerned.
    """
    slopes = torch.tensor(alibi_slopes(n_heads), dtype=torch.float32)
    ramp = torch.arange(seq_len, dtype=torch.float32)
    bias = slopes[:, None, None] * ramp[None, None, :]
    return bias.expand(n_heads, seq_len, seq_len) + causal_mask(seq_len)[None]


# --------------------------------------------------------------------------
# primitive layers  (base_layers.py)
# --------------------------------------------------------------------------
def final_norm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """Biased-variance layer norm with params g,b.  base_layers.py:36-41."""
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).square().mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g + b


def conv1d(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """x @ W + b with W stored [1, in, out].  base_layers.py:54-57."""
    nf = w.shape[-1]
    y = x.reshape(-1, x.shape[-1]) @ w.reshape(-1, nf) + b
    return y.reshape(*x.shape[:-1], nf)


def _heads(x: Tensor, n_heads: int) -> Tensor:
    """[B,S,d] -> [B,H,S,dh].  base_layers.py:69-74."""
    B, S, d = x.shape
    return x.reshape(B, S, n_heads, d // n_heads).permute(0, 2, 1, 3)


def _merge(x: Tensor) -> Tensor:
    """[B,H,S,dh] -> [B,S,d].  base_layers.py:76-79."""
    B, H, S, dh = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, S, H * dh)


def alibi_attention(q: Tensor, k: Tensor, v: Tensor, mask: Tensor) -> Tensor:
    """softmax(q k^T / sqrt(dh) + mask) v.  alibi_unified.py:40-54."""
    w = q @ k.transpose(-1, -2)
    w = w / math.sqrt(q.shape[-1])
    w = w + mask
    return w.softmax(-1) @ v


def skew_relative_logits(q: Tensor, E: Tensor, max_sequence: int) -> Tensor:
    """Music-Transformer skew: q E^T, pad one column, reshape, drop a row.

    relative_transformer.py:18-32.  q:[B,H,S,dh], E:[H,max_seq,dh] ->
    [B,H,S,S]; entry (i,j<=i) equals q_i . E[max_seq-1-(i-j)].
    """
    B, H, S, dh = q.shape
    qh = q.permute(1, 0, 2, 3).reshape(H, B * S, dh)
    e = E[:, max_sequence - S:]
    rel = (qh @ e.transpose(-1, -2)).reshape(H, B, S, S)
    rel = F.pad(rel, (1, 0))
    rel = rel.reshape(H, B, S + 1, S)[:, :, 1:]
    return rel.permute(1, 0, 2, 3)


def relative_attention(q, k, v, E, max_sequence, mask):
    """softmax((q k^T + rel) * dh^-1/2 + mask) v.  relative_transformer.py:34-44."""
    w = q @ k.transpose(-1, -2)
    w = w + skew_relative_logits(q, E, max_sequence)
    w = w * (1.0 / (q.shape[-1]) ** 0.5)
    w = w + mask
    return w.softmax(-1) @ v


_ACT = {"gelu": F.gelu, "relu": F.relu}


def rotary_tables(seq_len: int, dim: int):
    """FixedPositionalEmbedding (old_models/rotary_transformer.py:24-42): sin | cos of position x inv_freq, fp32.
    Returns (cos, sin), each [seq_len, dim/2]."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    ang = torch.einsum("i,j->ij", torch.arange(0, seq_len, dtype=torch.float), inv_freq)
    return ang.cos(), ang.sin()


def apply_rotary(t: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """rotate_every_two + apply_rotary_pos_emb (rotary_transformer.py:9-21) on [B,H,S,dh]: pair (2i, 2i+1) of
    the head dim is rotated by the angle of frequency i (each table entry is repeated for both pair members)."""
    c = cos.repeat_interleave(2, dim=-1)
    s = sin.repeat_interleave(2, dim=-1)
    x1, x2 = t[..., 0::2], t[..., 1::2]
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
    return (t * c) + (rot * s)


def decoder_block(sd: StateDict, pre: str, x: Tensor, mask: Tensor, n_heads: int,
                  activation: str, rel_max_sequence: Optional[int] = None, rotary=None, p_drop: float = 0.0) -> Tensor:
    """Post-LN block.  alibi_unified.py:86-95 with attention :56-67; relative variant
    relative_transformer.py:46-58,77-86.  p_drop > 0: training-mode dropout after the attention projection (:88) and
    after the activation (:91), torch's CPU generator (parity tests run with p_drop = 0: RNG streams cannot match)."""
    d = x.shape[-1]
    c = conv1d(x, sd[pre + "attn.c_attn.w"], sd[pre + "attn.c_attn.b"])
    q, k, v = (_heads(t, n_heads) for t in torch.split(c, d, dim=2))
    if rotary is not None:
        q, k = apply_rotary(q, *rotary), apply_rotary(k, *rotary)
    if rel_max_sequence is None:
        a = alibi_attention(q, k, v, mask)
    else:
        a = relative_attention(q, k, v, sd[pre + "attn.E"], rel_max_sequence, mask)
    a = conv1d(_merge(a), sd[pre + "attn.c_proj.w"], sd[pre + "attn.c_proj.b"])
    if p_drop > 0:
        a = F.dropout(a, p_drop, training=True)
    x = F.layer_norm(a + x, (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    h = _ACT[activation](F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"]))
    if p_drop > 0:
        h = F.dropout(h, p_drop, training=True)
    h = F.linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return F.layer_norm(x + h, (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])


def _n_layers(sd: StateDict) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("decoder_layers."))


def _trunk(sd, h, mask, n_heads, activation, rel_max_sequence=None, rotary=None, p_drop=0.0):
    for i in range(_n_layers(sd)):
        h = decoder_block(sd, f"decoder_layers.{i}.", h, mask, n_heads, activation, rel_max_sequence, rotary, p_drop)
    return final_norm(h, sd["norm.g"], sd["norm.b"])


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
def unified_forward(sd: StateDict, x: Tensor, n_heads: int, activation: str = "gelu", p_drop: float = 0.0) -> Tensor:
    """alibi_unified.py:124-138.  x:[B,S] int -> logits [B,S,V]."""
    S = x.shape[1]
    h = F.embedding(x.long(), sd["embed.weight"])
    h = _trunk(sd, h, alibi_mask(S, n_heads), n_heads, activation, p_drop=p_drop)
    return F.linear(h, sd["proj_class.weight"], sd["proj_class.bias"])


def separated_forward(sd: StateDict, x: Tensor, n_heads: int, activation: str = "gelu"):
    """alibi_separated.py:133-154.  x:[B,3,S] -> (note, velocity, duration) logits."""
    S = x.shape[2]
    xl = x.long()
    h = torch.cat([F.embedding(xl[:, 0], sd["embed_note.weight"]),
                   F.embedding(xl[:, 1], sd["embed_velocity.weight"]),
                   F.embedding(xl[:, 2], sd["embed_duration.weight"])], dim=-1)
    h = _trunk(sd, h, alibi_mask(S, n_heads), n_heads, activation)
    return tuple(F.linear(h, sd[f"proj_{n}.weight"], sd[f"proj_{n}.bias"])
                 for n in ("note", "velocity", "duration"))


def relative_forward(sd: StateDict, x: Tensor, n_heads: int, max_sequence: int,
                     activation: str = "relu") -> Tensor:
    """relative_transformer.py:107-121.  Tied input/output embedding, no head bias."""
    S = x.shape[1]
    emb = sd["embedding"]
    h = emb[x.long()]
    h = _trunk(sd, h, causal_mask(S), n_heads, activation, rel_max_sequence=max_sequence)
    return h @ emb.t()


def tied_alibi_forward(sd: StateDict, x: Tensor, n_heads: int, activation: str = "gelu") -> Tensor:
    """old_models/alibi.py:114-130: ALiBi trunk with the raw `embedding` parameter tied to the output."""
    emb = sd["embedding"]
    h = _trunk(sd, emb[x.long()], alibi_mask(x.shape[1], n_heads), n_heads, activation)
    return h @ emb.t()


def rotary_forward(sd: StateDict, x: Tensor, n_heads: int, activation: str = "relu", as_reference: bool = True) -> Tensor:
    """old_models/rotary_transformer.py:125-144: rotary q/k, plain causal mask, tied embedding.

    as_reference: the reference's forward calls ``self.pos_embed(x)`` with the int64 TOKEN tensor, and
    FixedPositionalEmbedding.forward ends in ``.to(x)`` (rotary_transformer.py:41-42,133): the sin | cos table is cast
    to int64, i.e. truncated toward zero, before it multiplies q and k (cos = 1 only where the angle is exactly
    0, sin = 0 everywhere).  That is what the reference computes, so it is what parity means; as_reference=False
    gives the intended rotation."""
    emb = sd["embedding"]
    S, d = x.shape[1], emb.shape[1]
    cos, sin = rotary_tables(S, d // n_heads)
    if as_reference:
        cos, sin = cos.to(torch.int64).to(torch.float32), sin.to(torch.int64).to(torch.float32)
    h = _trunk(sd, emb[x.long()], causal_mask(S), n_heads, activation, rotary=(cos, sin))
    return h @ emb.t()


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def unified_criterion(logits: Tensor, y: Tensor, pad_index: Optional[int]):
    """Mean CE over non-pad targets.  alibi_unified.py:158-163."""
    ii = -100 if pad_index is None else pad_index
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), y.reshape(-1).long(), ignore_index=ii)
    return loss, {"loss": loss}


def separated_class_weights(vocab_n: int, vocab_v: int) -> Tuple[Tensor, Tensor]:
    """weight[1]=0, weight[2]=0.05, else 1.  alibi_separated.py:179-187."""
    wn = torch.ones(vocab_n); wn[1] = 0; wn[2] = 0.05
    wv = torch.ones(vocab_v); wv[1] = 0; wv[2] = 0.05
    return wn, wv


def separated_criterion(preds, y: Tensor, weight_n: Tensor, weight_v: Tensor, loss_weights=None):
    """Three CEs, weighted-mean reduction for the weighted ones.  alibi_separated.py:189-198."""
    lw = loss_weights or {"loss_notes": 1, "loss_velocity": 1, "loss_duration": 1}
    n, v, d = preds
    flat = lambda t: t.reshape(-1, t.shape[-1])
    yl = y.long()
    out = {
        "loss_notes": F.cross_entropy(flat(n), yl[:, 0].reshape(-1), weight_n),
        "loss_velocity": F.cross_entropy(flat(v), yl[:, 1].reshape(-1), weight_v),
        "loss_duration": F.cross_entropy(flat(d), yl[:, 2].reshape(-1), ignore_index=1),
    }
    total = sum(lw[k] * out[k] for k in ("loss_notes", "loss_velocity", "loss_duration"))
    out["loss"] = total
    return total, out


# --------------------------------------------------------------------------
# sampling loops
# --------------------------------------------------------------------------
def sample_next_plus_one(forward: Callable[[Tensor], Tensor], seq: Tensor, idx: int = 1,
                         deterministic: bool = False, generator=None) -> Tensor:
    """model_baseline/inference_unified.py:12-21.

    Position i receives a draw from the logits at position i+1 of a forward
    over the whole (partly stale) buffer; full re-forward per token.
    """
    with torch.no_grad():
        for i in range(idx, seq.shape[-1] - 1):
            logits = forward(seq)
            if deterministic:
                r = torch.argmax(logits[:, i + 1], -1)
            else:
                r = torch.multinomial(logits[:, i + 1].softmax(-1), 1, generator=generator).squeeze(-1)
            seq[:, i] = r
    return seq


def sample_last_position(forward, seq: Tensor, idx: int = 0, deterministic: bool = False,
                         generator=None) -> Tensor:
    """AlibiTransformer.inference, alibi_unified.py:140-149 (flag is inverted:
    deterministic=True draws from the multinomial)."""
    with torch.no_grad():
        for i in range(idx, seq.shape[-1]):
            logits = forward(seq)
            if deterministic:
                r = torch.multinomial(logits[:, -1].softmax(-1), 1, generator=generator).squeeze(-1)
            else:
                r = torch.argmax(logits[:, -1], -1)
            seq[:, i] = r
    return seq


def sample_next_plus_one_separated(forward, seq: Tensor, idx: int = 1, deterministic: bool = False,
                                   generator=None) -> Tensor:
    """model_baseline/inference_separated.py:18-28.  seq:[B,3,L]."""
    with torch.no_grad():
        for i in range(idx, seq.shape[-1] - 1):
            heads = forward(seq)
            if deterministic:
                r = [torch.argmax(h[:, i + 1], -1) for h in heads]
            else:
                r = [torch.multinomial(h[:, i + 1].softmax(-1), 1, generator=generator).squeeze(-1) for h in heads]
            seq[:, :, i] = torch.stack(r, -1)
    return seq


# --------------------------------------------------------------------------
# training step  (train_utils.py:160-173, fp32, GradScaler disabled)
# --------------------------------------------------------------------------
def make_trainable(sd: StateDict) -> StateDict:
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def unified_train_step(sd: StateDict, opt: torch.optim.Optimizer, x: Tensor, y: Tensor,
                       n_heads: int, pad_index: int = 1, activation: str = "gelu", p_drop: float = 0.0) -> float:
    """forward, CE, zero_grad, backward, optimizer.step; returns the loss."""
    logits = unified_forward(sd, x, n_heads, activation, p_drop)
    loss, _ = unified_criterion(logits, y, pad_index)
    assert not torch.isnan(loss).any()
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.detach())


# --------------------------------------------------------------------------
# synthetic MIDI-grammar token stream (mimics
# data/transforms.py:12-38,149-152 vocabulary layout, vocab 618)
# --------------------------------------------------------------------------
def synthetic_unified_tokens(batch: int, seq_len: int, seed: int = 1234) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(batch, seq_len, dtype=torch.int64)
    for b in range(batch):
        n_events = seq_len  # over-generate then cut
        kind = torch.rand(n_events, generator=g) < 0.75
        note = 2 + torch.randint(0, 89, (n_events,), generator=g)
        vel = 91 + torch.randint(0, 127, (n_events,), generator=g)
        dur = 218 + torch.randint(0, 200, (n_events,), generator=g)
        shift = 418 + torch.randint(0, 200, (n_events,), generator=g)
        toks = [0]
        for e in range(n_events):
            if len(toks) >= seq_len:
                break
            if kind[e]:
                toks += [int(note[e]), int(vel[e]), int(dur[e])]
            else:
                toks.append(int(shift[e]))
        row = torch.tensor(toks[:seq_len], dtype=torch.int64)
        pad = int(torch.randint(0, seq_len // 10 + 1, (1,), generator=g))
        if pad:
            row[seq_len - pad:] = 1
        out[b] = row
    return out
