"""Primitive layers with the reference's names, parameters and state_dict keys
(base_layers.py), running on the libmgx kernels.

Compute dtype: every module computes in fp32 unless bf16 is selected, either
by the owning model's ``dtype='bf16'`` constructor kwarg or by an enclosing
``torch.autocast`` context (the reference's ``mixed_precision=True`` path,
train_utils.py:164; fp16 autocast is mapped to bf16 — no loss scaling needed).
Parameters stay fp32 masters; bf16 shadows are cached per parameter version.
"""
from __future__ import annotations

import weakref

import torch
from torch import nn

from . import ops

def generate_square_subsequent_mask(sz, device):
    """0 on/below the diagonal, -inf above (base_layers.py:19-22).  Kept for API
    compatibility; the attention kernels never materialise it."""
    return torch.triu(torch.full((sz, sz), float("-inf"), device=device), diagonal=1)


def checkpoint_wrapper(module, *args, apply=False, over_ride=None):
    """base_layers.py:6-16: activation checkpointing only when over_ride is True."""
    if over_ride is True:
        from torch.utils import checkpoint
        return checkpoint.checkpoint(lambda *x: module(*x), *args)
    return module(*args)


def resolve_dtype(requested):
    """fp32 / bf16 from a constructor kwarg value."""
    if requested in (None, "fp32", "float32", torch.float32):
        return torch.float32
    if requested in ("bf16", "bfloat16", torch.bfloat16):
        return torch.bfloat16
    raise ValueError(f"dtype must be 'fp32' or 'bf16', got {requested!r}")


def active_dtype(module_dtype):
    """bf16 under autocast (any low-precision autocast dtype maps to bf16), else the module's own."""
    if torch.is_autocast_enabled():
        return torch.bfloat16
    return module_dtype


class ShadowCache:
    """Compute-dtype copies of fp32 master parameters, refreshed when the master's version changes."""

    def __init__(self):
        self._c = {}
        self._ref = {}

    def _own(self, p):
        """Entries are keyed by id(p): make sure that id still means this tensor (a parameter that was replaced and
        freed can hand its id to a new one, whose version counter may well match the stale entry's)."""
        k = id(p)
        r = self._ref.get(k)
        if r is None or r() is not p:
            for key in [key for key in self._c if key[0] == k or (isinstance(key[0], tuple) and k in key[0])]:
                del self._c[key]
            self._ref[k] = weakref.ref(p)

    def get(self, p: torch.Tensor, dtype, shape=None):
        if dtype == p.dtype:
            t = p.detach()
            return t if shape is None else t.view(shape)
        self._own(p)
        key = (id(p), dtype)
        ent = self._c.get(key)
        if ent is None or ent[0] != p._version or ent[1].device != p.device:
            ent = (p._version, p.detach().to(dtype))
            self._c[key] = ent
        return ent[1] if shape is None else ent[1].view(shape)

    def get_t(self, p: torch.Tensor, dtype, shape):
        """Transposed contiguous compute-dtype copy of the 2-D view `shape` of p (cached like get)."""
        self._own(p)
        key = (id(p), dtype, "t")
        ent = self._c.get(key)
        if ent is None or ent[0] != p._version or ent[1].device != p.device:
            ent = (p._version, p.detach().view(shape).t().to(dtype).contiguous())
            self._c[key] = ent
        return ent[1]

    @staticmethod
    def pair_supported(p: torch.Tensor, dtype, shape) -> bool:
        return (p.is_cuda and p.dtype == torch.float32 and dtype == torch.bfloat16 and len(shape) == 2
                and shape[0] % 64 == 0 and shape[1] % 64 == 0)

    def refresh_pairs(self, items, dtype):
        """items: [(p, shape)] of fp32 masters wanted as bf16 in both orientations.  Every stale pair is rewritten
        in place by ONE mgx_cast_shadow_bf16 launch (one pass over each master for both copies; the buffers are
        allocated once), instead of a cast plus a strided transposing copy per weight per optimizer step."""
        stale = []
        for p, shape in items:
            self._own(p)
            key = (id(p), dtype, "pair")
            ent = self._c.get(key)
            if ent is None or ent[1].device != p.device:
                ent = [-1, torch.empty(shape, dtype=dtype, device=p.device),
                       torch.empty(shape[::-1], dtype=dtype, device=p.device)]
                self._c[key] = ent
            if ent[0] != p._version:
                stale.append((p, shape, ent))
        if stale:
            ops.cast_shadow([p.detach().view(shape) for p, shape, _ in stale], [e[1] for _, _, e in stale],
                            [e[2] for _, _, e in stale])
            for p, _, ent in stale:
                ent[0] = p._version

    def pair(self, p: torch.Tensor, dtype, shape, transposed: bool):
        self._own(p)
        ent = self._c.get((id(p), dtype, "pair"))
        if ent is None or ent[0] != p._version or ent[1].device != p.device:
            self.refresh_pairs([(p, shape)], dtype)
            ent = self._c[(id(p), dtype, "pair")]
        return ent[2] if transposed else ent[1]

    def refresh_many(self, params, dtype):
        """Compute-dtype copies of several (small) masters: every stale one is rewritten in place by ONE
        multi-tensor copy (a block's four bias vectors were four 5-us cast launches per layer per step)."""
        stale_dst, stale_src = [], []
        for p in params:
            if p.dtype == dtype:
                continue
            self._own(p)
            key = (id(p), dtype)
            ent = self._c.get(key)
            if ent is None or ent[1].device != p.device or not isinstance(ent, list):
                ent = [-1, torch.empty(p.shape, dtype=dtype, device=p.device)]
                self._c[key] = ent
            if ent[0] != p._version:
                stale_dst.append(ent[1])
                stale_src.append(p.detach())
                ent[0] = p._version
        if stale_dst:
            torch._foreach_copy_(stale_dst, stale_src)

    def padded_stack(self, ps, dtype, rows: int, transposed: bool = False):
        """Compute-dtype copy of the row-wise concatenation of the masters ps ([V_k, ...] each), zero-padded to
        `rows` rows (the output projections with a padded class dimension); transposed=True gives the [cols, rows]
        copy (the dX operand).  Allocated once, rewritten in place when any master's version changes."""
        for p in ps:
            self._own(p)
        key = (tuple(id(p) for p in ps), dtype, "pad", rows)
        ent = self._c.get(key)
        dev = ps[0].device
        if ent is None or ent[1].device != dev:
            pad = torch.zeros((rows,) + tuple(ps[0].shape[1:]), dtype=dtype, device=dev)
            pad_t = torch.empty((pad.shape[1], rows), dtype=dtype, device=dev) if pad.dim() == 2 else None
            ent = [None, pad, pad_t, -1]
            self._c[key] = ent
        ver = tuple(p._version for p in ps)
        if ent[0] != ver:
            r0 = 0
            for p in ps:
                ent[1][r0:r0 + p.shape[0]].copy_(p.detach())
                r0 += p.shape[0]
            ent[0] = ver
            ent[3] = -1
        if not transposed:
            return ent[1]
        if ent[3] != 0:                                    # transposed copy follows lazily
            ent[2].copy_(ent[1].t())
            ent[3] = 0
        return ent[2]

    def clear(self):
        self._c.clear()
        self._ref.clear()


class BlockWeights:
    """Compute-dtype views of one decoder block's parameters, in every orientation the GEMMs want.

    Conv1d weights are [in, out] (x @ w), nn.Linear weights [out, in].  The hand-written NT GEMM
    takes its second operand as [N, K] (K contiguous), so each weight is served both ways; the
    copies are made once per optimizer step (ShadowCache keys on the parameter version).
    """

    def __init__(self, cache: ShadowCache, masters, d: int, dtype):
        (self.m_wqkv, self.m_bqkv, self.m_wp, self.m_bp, self.m_w1, self.m_b1, self.m_w2, self.m_b2,
         self.m_g1, self.m_be1, self.m_g2, self.m_be2) = masters
        self.masters = masters
        self.dtype = dtype
        self._c, self._d = cache, d
        self._shapes = {"wqkv": (d, 3 * d), "wp": (d, d), "w1": tuple(self.m_w1.shape), "w2": tuple(self.m_w2.shape)}
        ws = [(self.m_wqkv, self._shapes["wqkv"]), (self.m_wp, self._shapes["wp"]), (self.m_w1, self._shapes["w1"]),
              (self.m_w2, self._shapes["w2"])]
        self._pairs = all(cache.pair_supported(p, dtype, shape) for p, shape in ws)
        if self._pairs:
            cache.refresh_pairs(ws, dtype)               # the block's four weights, both orientations, one launch
        if self.m_bqkv.is_cuda:
            cache.refresh_many((self.m_bqkv, self.m_bp, self.m_b1, self.m_b2), dtype)      # and its four biases

    def _w(self, name, transposed):
        p, shape = getattr(self, "m_" + name), self._shapes[name]
        if self._pairs:
            return self._c.pair(p, self.dtype, shape, transposed)
        return self._c.get_t(p, self.dtype, shape) if transposed else self._c.get(p, self.dtype, shape)

    # [K, N] orientation for x @ W, [N, K] for the NT kernel
    @property
    def wqkv(self):            # [d, 3d]
        return self._w("wqkv", False)

    @property
    def wqkv_t(self):          # [3d, d]
        return self._w("wqkv", True)

    @property
    def wp(self):              # [d, d]
        return self._w("wp", False)

    @property
    def wp_t(self):
        return self._w("wp", True)

    @property
    def w1(self):              # [ff, d]
        return self._w("w1", False)

    @property
    def w1_t(self):            # [d, ff]
        return self._w("w1", True)

    @property
    def w2(self):              # [d, ff]
        return self._w("w2", False)

    @property
    def w2_t(self):            # [ff, d]
        return self._w("w2", True)

    @property
    def bqkv(self):
        return self._c.get(self.m_bqkv, self.dtype)

    @property
    def bp(self):
        return self._c.get(self.m_bp, self.dtype)

    @property
    def b1(self):
        return self._c.get(self.m_b1, self.dtype)

    @property
    def b2(self):
        return self._c.get(self.m_b2, self.dtype)


class Norm(nn.Module):
    """Biased-variance layer norm with parameters g, b (base_layers.py:25-41) on the fused LN kernel."""

    def __init__(self, n_state, axis=-1, epsilon=1e-5):
        super().__init__()
        if axis != -1:
            raise ValueError("only axis=-1 is supported")
        self.n_state = n_state
        self.g = nn.Parameter(torch.ones([n_state]))
        self.b = nn.Parameter(torch.zeros([n_state]))
        self.axis = axis
        self.epsilon = epsilon

    def forward(self, x):
        from . import torch_ops
        return torch_ops.layer_norm(x, self.g, self.b, self.epsilon)        # torch.ops.mgx.layer_norm_fwd


class Conv1d(nn.Module):
    """x @ w + b with w stored [1, nx, nf], N(0, stdev) init (base_layers.py:44-57)."""

    def __init__(self, nf, nx, stdev=0.02):
        super().__init__()
        self.nf = nf
        self.nx = nx
        self.stdev = stdev
        self.w = nn.Parameter(torch.normal(size=[1, nx, nf], mean=0.0, std=stdev))
        self.b = nn.Parameter(torch.zeros([nf]))

    def forward(self, x):
        w = self.w.view(self.nx, self.nf)
        return ops.linear(x, w if w.dtype == x.dtype else w.to(x.dtype), self.b, w_is_nk=False)   # mgx_gemm_generic


class Attention(nn.Module):
    """Head split/merge helpers (base_layers.py:60-79).  The fused kernels read
    the packed [B,S,3,H,dh] projection directly; these exist for API parity."""

    def __init__(self, heads, n_state):
        super().__init__()
        assert n_state % heads == 0
        self.heads = heads
        self.n_state = n_state
        self.depth = n_state // heads

    def split_heads(self, x, batch, seq_len):
        return x.reshape(batch, seq_len, self.heads, self.depth).permute(0, 2, 1, 3)

    def combine_heads(self, x, batch, seq_len):
        return x.permute(0, 2, 1, 3).reshape(batch, seq_len, self.n_state)


class Mlp(nn.Module):
    """conv_fc -> gelu -> conv_proj (base_layers.py:82-93)."""

    def __init__(self, input_dim, proj_dim):
        super().__init__()
        self.input_dim = input_dim
        self.proj_dim = proj_dim
        self.conv_fc = Conv1d(proj_dim, input_dim)
        self.conv_proj = Conv1d(input_dim, proj_dim)

    def forward(self, x):
        return self.conv_proj(nn.functional.gelu(self.conv_fc(x)))


def tied_embed(x, table, dtype):
    """h = table[x] in the compute dtype for the models whose raw ``embedding`` parameter is both the input table and
    the output projection (relative_transformer.py:111, old_models/alibi.py, rotary_transformer.py).  bf16: one gather +
    cast kernel whose backward is a one-hot GEMM."""
    if dtype == torch.bfloat16 and ops.embed_bf16_supported(x, table):
        return ops.embed_bf16(x, table)
    return nn.functional.embedding(x, table if dtype == torch.float32 else table.to(dtype))


def tied_head(h, table, shadow: ShadowCache):
    """logits = h @ table^T, no bias (relative_transformer.py:116-119).  bf16: the class dimension is padded to a
    multiple of 64 inside and all three GEMMs run on the MFMA kernels; the caller gets the [.., :V] view (the padded
    buffer rides along as ``mgx_padded_base`` for the criterion)."""
    V = table.shape[0]
    if ops.padded_head_supported(h, table):
        vp = (V + 63) // 64 * 64
        w_pad = shadow.padded_stack((table,), h.dtype, vp)
        w_pad_t = shadow.padded_stack((table,), h.dtype, vp, transposed=True)
        padded = ops.padded_head(h, w_pad, w_pad_t, None, (table, None))
        logits = padded[..., :V]
        logits.mgx_padded_base = padded
        return logits
    emb = table if h.dtype == torch.float32 else table.to(h.dtype)
    return ops.linear(h, emb)                                  # fp32 / odd widths: mgx_gemm_generic
