"""KV-cached autoregressive sampling for the ALiBi models.

Reproduces the reference loops bit-for-bit in their token semantics
(model_baseline/inference_unified.py:12-21, inference_separated.py:18-28):
the token written to position i is drawn from the logits at position i+1 of a
causal forward over the whole buffer, whose positions i and i+1 still hold
their previous (stale) contents.  Instead of L-2 full forwards (O(L^2)
attention each) every step runs three token rows — position i-1 (final since
the previous step: its K/V become permanent cache entries) and the two
provisional rows i, i+1 — against a per-layer K/V cache [B,H,Lmax,dh] with the
mgx_decode_attn kernel, and keeps the logits of row i+1.

Decode is partitioned by batch only (replicas; no collective).
"""
from __future__ import annotations

import torch

from . import ops
from .base_layers import active_dtype


class KVCacheDecoder:
    """Row-wise inference engine over a musicgeneration_amd AlibiTransformer (unified or separated)."""

    def __init__(self, model, batch: int, max_len: int, dtype=None):
        self.model = model
        self.dtype = dtype or active_dtype(model.compute_dtype)
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("KVCacheDecoder needs the model on the GPU (no CPU fallback)")
        self.device = p.device
        H = model.heads
        dh = model.d_model // H
        self.batch, self.max_len = batch, max_len
        self.caches = [(torch.zeros(batch, H, max_len, dh, dtype=self.dtype, device=self.device),
                        torch.zeros(batch, H, max_len, dh, dtype=self.dtype, device=self.device))
                       for _ in model.decoder_layers]
        self._ws = None                                   # fp32 split-K workspace of the decode projections
        self._head_bias32 = None                          # separated model: the three head biases stacked, fp32
        with torch.no_grad():
            self._head_wb = tuple(t.detach() for t in model.head_weights(self.dtype))   # cast once per decoder

    @torch.no_grad()
    def rows(self, tokens: torch.Tensor, pos0) -> torch.Tensor:
        """Run token rows at absolute positions pos0.. through every layer.

        tokens: [B,R] (unified) or [B,3,R] (separated).  Their K/V are written
        into the caches at pos0..pos0+R-1 (overwriting provisional entries).
        R <= 4 uses the decode kernel; longer runs must start at pos0 == 0
        (prefill with the flash kernel).  Returns head logits for all R rows.
        pos0 is an int, or a one-element int32 GPU tensor read by the kernels at
        execution time (graph replay; R <= 4 only).
        """
        m = self.model
        h = self.trunk(m.embed_tokens(tokens, self.dtype), pos0)
        B, R, d = h.shape
        hn, _, _, _ = ops.ln_residual_fwd(h.reshape(B * R, d), None, m.norm.g, m.norm.b, m.norm.epsilon, save=False)
        return m.head(hn.view(B, R, d), self._head_wb)

    @torch.no_grad()
    def trunk(self, h: torch.Tensor, pos0) -> torch.Tensor:
        """Embedded rows [B,R,d] at positions pos0.. through every decoder block (K/V filed into the caches);
        returns the last block's output, before the final Norm."""
        m = self.model
        dt = self.dtype
        B, R, d = h.shape
        if isinstance(pos0, torch.Tensor):
            if R > 4:
                raise ValueError("a device-side position needs R <= 4")
        else:
            if pos0 + R > self.max_len:
                raise ValueError("rows beyond the cache length")
            if R > 4 and pos0 != 0:
                raise ValueError("prefill (more than 4 rows) must start at position 0")
        T = B * R
        ff = m.decoder_layers[0].linear1.weight.shape[0]
        skinny = (dt == torch.bfloat16 and T <= 256 and R <= 4
                  and d % 64 == 0 and ff % 64 == 0 and d <= 2048
                  and ops.decode_step_attn_supported(dt, d // m.decoder_layers[0].heads))
        if skinny:
            need = max(ops.skinny_workspace_floats(T, n, k) for n, k in ((3 * d, d), (d, d), (ff, d), (d, ff)))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
        for layer, (kc, vc) in zip(m.decoder_layers, self.caches):
            W = layer.weights(dt)
            x2 = h.reshape(T, d)
            if skinny:
                # decode projections on libmgx: K-split weight streaming + the block's elementwise tail in the
                # reduction (bias | bias+activation | bias+residual+LayerNorm): 8 short launches per layer
                # instead of 4 latency-bound library GEMMs + 2 LayerNorm + 1 bias-activation kernels
                ws = self._ws
                qkv = ops.skinny_gemm(x2, W.wqkv_t, W.m_bqkv, ws).view(B, R, 3 * d)
                a = ops.decode_step_attn(qkv, kc, vc, layer.attn.slopes, pos0)
                h1 = ops.skinny_gemm(a.view(T, d), W.wp_t, W.m_bp, ws, res=x2, gamma=W.m_g1, beta=W.m_be1,
                                     eps=layer.norm1.eps)
                hact = ops.skinny_gemm(h1, W.w1, W.m_b1, ws, act=layer.activation)
                h2 = ops.skinny_gemm(hact, W.w2, W.m_b2, ws, res=h1, gamma=W.m_g2, beta=W.m_be2, eps=layer.norm2.eps)
                h = h2.view(B, R, d)
                continue
            # fp32 (parity mode), prefill (R > 4) and widths off the skinny kernel's grid: the tuned NT GEMM where it
            # applies (bf16, K a multiple of 64), the generic libmgx GEMM otherwise
            nt = dt == torch.bfloat16 and d % 64 == 0 and ff % 64 == 0
            proj = (lambda xx, w_kn, w_nk, b: ops.gemm_nt(xx, w_nk(), b)) if nt else \
                (lambda xx, w_kn, w_nk, b: ops.gemm(xx, w_kn(), bias=b))
            qkv = proj(x2, lambda: W.wqkv, lambda: W.wqkv_t, W.m_bqkv).view(B, R, 3 * d)
            if R <= 4 and ops.decode_step_attn_supported(dt, d // layer.heads):
                a = ops.decode_step_attn(qkv, kc, vc, layer.attn.slopes, pos0)      # append + attend, one launch
            else:
                ops.kv_append(qkv, kc, vc, pos0)
                if R <= 4:
                    q = qkv[:, :, :d].contiguous()
                    a = ops.decode_attn(q, kc, vc, layer.attn.slopes, pos0)
                else:
                    a, _ = ops.alibi_attn_fwd(qkv, layer.attn.slopes, layer.heads)
            pr = proj(a.view(T, d), lambda: W.wp, lambda: W.wp_t, W.m_bp)
            h1, _, _, _ = ops.ln_residual_fwd(pr, x2, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps, save=False)
            u = ops.gemm_nt(h1, W.w1) if nt else ops.gemm(h1, W.w1, trans_b=True)
            _, hact = ops.bias_act_fwd(u, layer.linear1.bias, layer.activation)
            v = ops.gemm_nt(hact, W.w2, W.m_b2) if nt else ops.gemm(hact, W.w2, trans_b=True, bias=W.m_b2)
            h2, _, _, _ = ops.ln_residual_fwd(v, h1, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps, save=False)
            h = h2.view(B, R, d)
        return h

    def _sample_ends(self):
        """(embedding tables, head class counts, stacked fp32 head bias) of the unified or the separated model."""
        m = self.model
        if hasattr(m, "embed"):
            return (m.embed.weight,), (m.proj_class.weight.shape[0],), m.proj_class.bias.detach()
        tables = (m.embed_note.weight, m.embed_velocity.weight, m.embed_duration.weight)
        sizes = (m.vocab_note, m.vocab_velocity, m.vocab_duration)
        if self._head_bias32 is None:
            self._head_bias32 = torch.cat([m.proj_note.bias, m.proj_velocity.bias, m.proj_duration.bias], 0).detach().float()
        return tables, sizes, self._head_bias32

    def fused_step_supported(self, seq: torch.Tensor) -> bool:
        """The three-launch head/tail of a sampling step (csrc/sample.hip): unified [B,L] or separated [B,3,L]
        buffer, bf16, d <= 2048, fp32 embedding tables of widths that are multiples of 4."""
        m = self.model
        unified = seq.dim() == 2 and hasattr(m, "embed") and hasattr(m, "proj_class")
        separated = seq.dim() == 3 and seq.shape[1] == 3 and hasattr(m, "embed_note") and hasattr(m, "proj_note")
        if not (unified or separated) or self.dtype != torch.bfloat16:
            return False
        tables, _, _ = self._sample_ends()
        return (m.d_model % 8 == 0 and m.d_model <= 2048 and seq.dtype in (torch.int32, torch.int64)
                and all(t.dtype == torch.float32 and t.shape[1] % 4 == 0 for t in tables))

    @torch.no_grad()
    def sample_step(self, seq: torch.Tensor, pos: torch.Tensor, keys: torch.Tensor, stochastic=False, seed=0):
        """One step of inference_unified.py:14-20 / inference_separated.py:20-27 with the position in device memory:
        rows pos, pos+1, pos+2 (= i-1, i, i+1) run against the caches; seq[:, (k,) pos+1] receives, per head, the
        argmax of the logits of row pos+2 (deterministic) or a draw from their softmax (Gumbel-max, counter-based:
        seed, position, sequence, class); pos advances by one.  keys: zero-initialised int64 [B, heads] scratch."""
        m = self.model
        tables, sizes, bias = self._sample_ends()
        h = ops.sample_gather_embed(seq, pos, tables, 3)
        h = self.trunk(h, pos)
        ops.sample_head(h, m.norm.g, m.norm.b, m.norm.epsilon, self._head_wb[0], bias, sizes, keys, pos=pos,
                        stochastic=stochastic, seed=seed)
        ops.sample_commit(seq, pos, keys)


class GraphedSampler:
    """One sampling step of the reference loop (inference_unified.py:14-20) recorded as a HIP graph.

    The step reads its position from device memory (mgx_*_at entry points), gathers the three
    token rows i-1, i, i+1 from `seq`, runs them against the caches, draws token i and scatters it
    into `seq`, then advances the position: replaying the graph L-2 times is the whole loop, with
    no host round trip and ~45 kernel launches per step folded into one submission.  bf16, unified
    and separated, greedy and stochastic all take the fused head/tail of csrc/sample.hip; the
    stochastic draw is a counter-based Gumbel-max keyed by a seed taken from torch's generator when
    the sampler is built (torch.manual_seed makes a run reproducible; the stream is not
    torch.multinomial's).  fp32 (parity mode) keeps the draw in torch ops outside the graph.
    """

    def __init__(self, dec: KVCacheDecoder, seq: torch.Tensor, first: int, deterministic: bool, capture: bool = True):
        self.dec, self.seq, self.det = dec, seq, deterministic
        self.separated = seq.dim() == 3
        dev = seq.device
        self.fused = dec.fused_step_supported(seq)
        heads = 3 if self.separated else 1
        self.keys = torch.zeros(seq.shape[0], heads, dtype=torch.int64, device=dev) if self.fused else None
        self.seed = 0 if deterministic else int(torch.randint(0, 2 ** 62, (1,)).item())
        self.graph = None
        self.pos = torch.full((1,), first - 1, dtype=torch.int32, device=dev)   # position of row 0 = i-1
        self.off = torch.arange(3, device=dev)
        self.logits = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up outside capture (library handles, workspaces)
            keep = seq.clone()
            self._body()
            seq.copy_(keep)                      # the warm-up must not leave its token behind: the real
            self.pos.fill_(first - 1)            # first step has to see the stale contents of position i
        torch.cuda.current_stream(dev).wait_stream(side)
        if capture:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()

    def _body(self):
        seq = self.seq
        if self.fused:
            self.dec.sample_step(seq, self.pos, self.keys, stochastic=not self.det, seed=self.seed)
            return
        idx = (self.pos.long() + self.off)                       # i-1, i, i+1
        tok = seq.index_select(-1, idx)
        out = self.dec.rows(tok, self.pos)
        if self.separated:
            self.logits = [h[:, 2].float() for h in out]
        else:
            self.logits = out[:, 2].float()
        if self.det:
            self._write()
        self.pos.add_(1)

    def _write(self, at=None):
        seq = self.seq
        at = (self.pos.long() + 1) if at is None else at         # i
        if self.separated:
            r = torch.stack([_draw(h, self.det) for h in self.logits], -1).to(seq.dtype)     # [B,3]
            seq.index_copy_(2, at, r.unsqueeze(-1))
        else:
            seq.index_copy_(1, at, _draw(self.logits, self.det).to(seq.dtype).unsqueeze(-1))

    def step(self):
        if self.graph is None:
            self._body()                         # launch by launch, same kernels
        else:
            self.graph.replay()
        if not self.det and not self.fused:
            self._write(at=self.pos.long())      # the graph already advanced the position by one


def _draw(logits_row: torch.Tensor, deterministic: bool) -> torch.Tensor:
    if deterministic:
        return torch.argmax(logits_row, -1)
    return torch.multinomial(logits_row.float().softmax(-1), num_samples=1).squeeze(-1)


@torch.no_grad()
def sample_next_plus_one(model, seq: torch.Tensor, idx: int = 1, deterministic: bool = False,
                         progress=None, use_graph=True) -> torch.Tensor:
    """KV-cached equivalent of the reference `inference` loops; mutates and returns `seq`.

    seq [B,L] (unified) or [B,3,L] (separated), int32 or int64, on the GPU.
    use_graph: replay the step as a HIP graph (False: the same kernels, launch by launch).
    """
    separated = seq.dim() == 3
    L = seq.shape[-1]
    B = seq.shape[0]
    if idx < 1:
        raise ValueError("idx must be >= 1 (position idx-1 must exist)")
    dec = KVCacheDecoder(model, B, L)
    sl = (lambda a, b: seq[:, :, a:b]) if separated else (lambda a, b: seq[:, a:b])
    if idx >= 2:
        # positions 0..idx-2 are final before the first step
        # prefill: only the K/V of the primer matter (flash kernel when > 4 rows, decode kernel otherwise); the head is
        # not evaluated
        with torch.no_grad():
            dec.trunk(model.embed_tokens(sl(0, idx - 1), dec.dtype), 0)
    steps = range(idx, L - 1)
    if progress is not None:
        steps = progress(steps)
    if use_graph and len(range(idx, L - 1)) > 0:
        if not seq.is_contiguous():
            raise ValueError("seq must be contiguous")
        g = GraphedSampler(dec, seq, idx, deterministic)
        for _ in steps:
            g.step()
        return seq
    if dec.fused_step_supported(seq) and seq.is_contiguous() and len(range(idx, L - 1)) > 0:
        g = GraphedSampler(dec, seq, idx, deterministic, capture=False)     # the graphed step's kernels, launch by launch
        for _ in steps:
            g.step()
        return seq
    for i in steps:
        logits = dec.rows(sl(i - 1, i + 2), i - 1)        # rows i-1 (final), i, i+1 (provisional)
        if separated:
            r = torch.stack([_draw(h[:, 2], deterministic) for h in logits], -1)
            seq[:, :, i] = r.to(seq.dtype)
        else:
            seq[:, i] = _draw(logits[:, 2], deterministic).to(seq.dtype)
    return seq
