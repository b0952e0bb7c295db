"""GPU parity of each libmgx kernel (through the C ABI, via musicgeneration_amd.ops)
against the CPU oracle on the same seeded inputs.

Tolerances: fp32 kernels 1e-5 relative to the tensor's max magnitude (north_star);
bf16 kernels 2e-2, measured against the fp32 oracle evaluated on the bf16-rounded inputs.
"""
import math

import pytest
import torch

import golden_util as gu
from oracle import reference_port as O

pytestmark = pytest.mark.gpu

F32_TOL = 1e-5
BF16_TOL = 2e-2


def _ops():
    from musicgeneration_amd import ops
    return ops


def _ref_attn(qkv, H, slopes=None, E=None, max_seq=None):
    """oracle attention on packed qkv [B,S,3d] (fp32 CPU), returns [B,S,d]."""
    B, S, d3 = qkv.shape
    d = d3 // 3
    q, k, v = (O._heads(t, H) for t in torch.split(qkv, d, dim=2))
    if E is None and slopes is not None:
        ramp = torch.arange(S, dtype=torch.float32)
        mask = slopes.float()[:, None, None] * ramp[None, None, :] + O.causal_mask(S)[None]
        out = O.alibi_attention(q, k, v, mask)
    elif E is None:
        out = O.alibi_attention(q, k, v, O.alibi_mask(S, H))
    else:
        out = O.relative_attention(q, k, v, E, max_seq, O.causal_mask(S))
    return O._merge(out)


ATTN_CASES = [  # B, S, H, dh
    (2, 48, 8, 8), (1, 33, 6, 16), (2, 130, 4, 64), (1, 257, 2, 128), (1, 70, 3, 24),
]


@pytest.mark.parametrize("B,S,H,dh", ATTN_CASES)
def test_alibi_attention_fp32(B, S, H, dh):
    ops = _ops()
    g = torch.Generator().manual_seed(S * 131 + dh)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g)
    dout = torch.randn(B, S, H * dh, generator=g)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32)
    ref_in = qkv.clone().requires_grad_(True)
    ref = _ref_attn(ref_in, H)
    ref.backward(dout)
    out, lse = ops.alibi_attn_fwd(qkv.cuda(), slopes.cuda(), H)
    assert gu.rel_err(out, ref) < F32_TOL
    dqkv = ops.alibi_attn_bwd(dout.cuda(), qkv.cuda(), out, lse, slopes.cuda(), H)
    assert gu.rel_err(dqkv, ref_in.grad) < F32_TOL
    # lse definition: log sum exp of q.k/sqrt(dh) + slope*(j-i)
    q, k, _ = (O._heads(t, H) for t in torch.split(qkv, H * dh, dim=2))
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    idx = torch.arange(S)
    s = s + slopes[None, :, None, None] * (idx[None, :] - idx[:, None])[None, None] + O.causal_mask(S)
    assert gu.rel_err(lse, torch.logsumexp(s, -1)) < F32_TOL


# the last three: more (batch, head, key-block) pairs than workgroups in the persistent grid (2 per CU = 512), so that
# the dK/dV and dQ workgroups walk several pairs each (576 pairs: one or two; 1120: two or three; an odd number of key
# blocks leaves the middle one unpaired) and the tile stream crosses item boundaries; 480 pairs: one pair per workgroup
@pytest.mark.parametrize("B,S,H,dh", [(2, 130, 4, 64), (1, 257, 2, 128), (2, 384, 2, 64), (1, 512, 8, 128),
                                       (1, 100, 2, 16), (6, 300, 48, 64), (5, 520, 32, 128), (7, 300, 80, 64)])
@pytest.mark.parametrize("force_generic", [False, True])
def test_alibi_attention_bf16(B, S, H, dh, force_generic):
    ops = _ops()
    g = torch.Generator().manual_seed(S * 7 + dh)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).bfloat16()
    dout = torch.randn(B, S, H * dh, generator=g).bfloat16()
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32)
    ref_in = qkv.float().requires_grad_(True)
    ref = _ref_attn(ref_in, H)
    ref.backward(dout.float())
    out, lse = ops.alibi_attn_fwd(qkv.cuda(), slopes.cuda(), H, generic=force_generic)
    dqkv = ops.alibi_attn_bwd(dout.cuda(), qkv.cuda(), out, lse, slopes.cuda(), H, generic=force_generic)
    assert gu.rel_err(out.float(), ref) < BF16_TOL
    assert gu.rel_err(dqkv.float(), ref_in.grad) < BF16_TOL


def test_alibi_attention_bf16_asymmetric_exact():
    """Integer-valued data with one-hot queries: any swapped fragment index in the
    MFMA kernel shows up as an exact mismatch (guide: test maps with asymmetric data)."""
    ops = _ops()
    B, S, H, dh = 1, 192, 1, 64
    q = torch.zeros(B, S, dh)
    k = torch.zeros(B, S, dh)
    v = torch.zeros(B, S, dh)
    for i in range(S):
        q[0, i, i % dh] = 8.0
        k[0, i, (i * 7) % dh] = float((i % 5) - 2)
        v[0, i] = torch.arange(dh) * 0.0625 + (i % 9)
    qkv = torch.cat([q, k, v], -1).bfloat16()
    slopes = torch.tensor([0.25])
    ref = _ref_attn(qkv.float(), H, slopes=slopes)
    out, _ = ops.alibi_attn_fwd(qkv.cuda(), slopes.cuda(), H)
    assert gu.rel_err(out.float(), ref) < 1e-2


@pytest.mark.parametrize("dh", [64, 128])
def test_alibi_attention_bf16_rescale_branch(dh):
    """Forces the deferred-rescale branch of the flash kernel (guide rule 26): keys are spiked so the running max
    jumps by far more than the threshold (2^8) at a chosen tile, after tiles that were accumulated against the old
    max -- both for a kernel that walks the keys forwards (spike right before the query) and for one that walks them
    backwards from the diagonal (spike 230 keys before the query); compared on the FULL tensor."""
    ops = _ops()
    B, S, H = 1, 448, 2
    g = torch.Generator().manual_seed(11)
    q = 0.3 * torch.randn(B, S, H, dh, generator=g)
    k = 0.3 * torch.randn(B, S, H, dh, generator=g)
    v = torch.randn(B, S, H, dh, generator=g)
    for j, ahead in ((70, 40), (200, 40), (330, 40), (10, 230), (100, 230)):
        k[0, j] = 30.0 * q[0, j + ahead] / q[0, j + ahead].norm(dim=-1, keepdim=True)   # aligned with a later query:
        # score 30 * |q| / sqrt(dh) = 9, i.e. 13 in log2 units
    qkv = torch.cat([q.reshape(B, S, -1), k.reshape(B, S, -1), v.reshape(B, S, -1)], -1).bfloat16()
    dout = torch.randn(B, S, H * dh, generator=g).bfloat16()
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32)
    ref_in = qkv.float().requires_grad_(True)
    ref = _ref_attn(ref_in, H)
    ref.backward(dout.float())
    out, lse = ops.alibi_attn_fwd(qkv.cuda(), slopes.cuda(), H)
    assert gu.rel_err(out.float(), ref) < BF16_TOL
    dqkv = ops.alibi_attn_bwd(dout.cuda(), qkv.cuda(), out, lse, slopes.cuda(), H)
    assert gu.rel_err(dqkv.float(), ref_in.grad) < BF16_TOL


@pytest.mark.parametrize("B,S,H,dh,M", [(2, 40, 8, 8, 64), (1, 64, 4, 16, 64), (2, 150, 2, 64, 160)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relative_attention(B, S, H, dh, M, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(S + dh + M)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).to(dtype)
    E = (0.5 * torch.randn(H, M, dh, generator=g)).to(dtype)
    dout = torch.randn(B, S, H * dh, generator=g).to(dtype)
    ref_in = qkv.float().requires_grad_(True)
    ref_E = E.float().requires_grad_(True)
    ref = _ref_attn(ref_in, H, E=ref_E, max_seq=M)
    ref.backward(dout.float())
    out, lse = ops.rel_attn_fwd(qkv.cuda(), E.cuda(), H)
    dqkv, dE = ops.rel_attn_bwd(dout.cuda(), qkv.cuda(), out, lse, E.cuda(), H)
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(out.float(), ref) < tol
    assert gu.rel_err(dqkv.float(), ref_in.grad) < tol
    assert gu.rel_err(dE, ref_E.grad) < tol


@pytest.mark.parametrize("B,S,H,dh,M", [(2, 150, 2, 64, 160), (1, 600, 3, 64, 640), (2, 257, 2, 128, 264),
                                         (1, 384, 2, 64, 384), (1, 31, 1, 64, 32), (1, 129, 2, 128, 1024), (3, 200, 2, 64, 256), (5, 96, 1, 64, 128)])
@pytest.mark.parametrize("force_generic", [False, True])
def test_relative_attention_bf16_mfma(B, S, H, dh, M, force_generic):
    """Skew relative attention on the MFMA path (ring skew through LDS, dE as a GEMM over the stored dG blocks) and on the
    generic path, against the oracle's pad/reshape skew in fp32 on the same bf16-rounded inputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(S * 3 + dh + M)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).bfloat16()
    E = (0.5 * torch.randn(H, M, dh, generator=g)).bfloat16()
    dout = torch.randn(B, S, H * dh, generator=g).bfloat16()
    ref_in = qkv.float().requires_grad_(True)
    ref_E = E.float().requires_grad_(True)
    ref = _ref_attn(ref_in, H, E=ref_E, max_seq=M)
    ref.backward(dout.float())
    out, lse = ops.rel_attn_fwd(qkv.cuda(), E.cuda(), H, generic=force_generic)
    dqkv, dE = ops.rel_attn_bwd(dout.cuda(), qkv.cuda(), out, lse, E.cuda(), H, generic=force_generic)
    assert gu.rel_err(out.float(), ref) < BF16_TOL
    q, k, _ = (O._heads(t, H) for t in torch.split(qkv.float(), H * dh, dim=2))
    w = (q @ k.transpose(-1, -2) + O.skew_relative_logits(q, E.float(), M)) / math.sqrt(dh) + O.causal_mask(S)
    assert float((lse.cpu() - torch.logsumexp(w, -1)).abs().max()) < 2e-2
    assert gu.rel_err(dqkv.float(), ref_in.grad) < BF16_TOL
    assert gu.rel_err(dE, ref_E.grad) < BF16_TOL
    assert float(dE[:, :M - S].abs().max()) == 0.0 if M > S else True


@pytest.mark.parametrize("B,S,H,dh,M", [(1, 2048, 2, 64, 4096), (1, 2304, 1, 64, 4096)])
def test_relative_attention_bf16_mfma_long(B, S, H, dh, M):
    """configs[4]'s geometry (max_sequence 4096, dh 64) at S >= 2048: ring skew, band pairing and dE indexing all
    depend on S and M.  Reference: the closed form of the skew (pinned against the reference's
    pad/reshape by tests/test_oracle_golden.py::test_relative_tiny) evaluated with torch ops in fp32 on the GPU,
    on the same bf16-rounded inputs; autograd gives dqkv and dE."""
    ops = _ops()
    g = torch.Generator().manual_seed(S + M)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).bfloat16().cuda()
    E = (0.5 * torch.randn(H, M, dh, generator=g)).bfloat16().cuda()
    dout = torch.randn(B, S, H * dh, generator=g).bfloat16().cuda()
    ref_in = qkv.float().requires_grad_(True)
    ref_E = E.float().requires_grad_(True)
    d = H * dh
    q, k, v = (t.view(B, S, H, dh).transpose(1, 2) for t in ref_in.split(d, -1))
    idx = torch.arange(S, device="cuda")
    dist = (idx.view(S, 1) - idx.view(1, S)).clamp_min(0)                  # i - j for j <= i
    qe = q @ ref_E[:, M - S:].transpose(-1, -2)                            # [B,H,S,S]: q_i . E[M-S+r]
    rel = torch.gather(qe, 3, (S - 1 - dist).view(1, 1, S, S).expand(B, H, S, S))   # r = S-1-(i-j)
    w = (q @ k.transpose(-1, -2) + rel) / math.sqrt(dh)
    w = w.masked_fill(idx.view(1, S) > idx.view(S, 1), float("-inf"))
    ref = (w.softmax(-1) @ v).transpose(1, 2).reshape(B, S, d)
    ref.backward(dout.float())
    out, lse = ops.rel_attn_fwd(qkv, E, H)
    dqkv, dE = ops.rel_attn_bwd(dout, qkv, out, lse, E, H)
    assert gu.rel_err(out.float(), ref) < BF16_TOL
    assert float((lse - torch.logsumexp(w, -1)).abs().max()) < 2e-2
    assert gu.rel_err(dqkv.float(), ref_in.grad) < BF16_TOL
    assert gu.rel_err(dE, ref_E.grad) < BF16_TOL
    assert float(dE[:, :M - S].abs().max()) == 0.0


@pytest.mark.parametrize("T,d", [(96, 64), (37, 96), (128, 512), (64, 1024), (20, 2048)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res", [True, False])
def test_layernorm_residual(T, d, dtype, with_res):
    ops = _ops()
    g = torch.Generator().manual_seed(T + d)
    a = torch.randn(T, d, generator=g).to(dtype)
    res = torch.randn(T, d, generator=g).to(dtype) if with_res else None
    gamma = 1 + 0.1 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    dy = torch.randn(T, d, generator=g).to(dtype)
    dy2 = torch.randn(T, d, generator=g).to(dtype)
    a_r = a.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    zin = a_r + (res.float() if with_res else 0)
    if dtype == torch.bfloat16:
        zin = zin + (zin.bfloat16().float() - zin).detach()     # the kernel normalises the stored (rounded) sum
    ref = O.final_norm(zin, gm, bt)
    ref.backward(dy.float() + dy2.float())
    y, z, mean, rstd = ops.ln_residual_fwd(a.cuda(), None if res is None else res.cuda(), gamma.cuda(), beta.cuda())
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(y.float(), ref) < tol
    dz, da, dg, db = ops.ln_residual_bwd(dy.cuda(), dy2.cuda(), z, mean, rstd, gamma.cuda())
    assert da.data_ptr() == dz.data_ptr()
    gtol = 2e-5 if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(dz.float(), a_r.grad) < gtol
    assert gu.rel_err(dg, gm.grad) < gtol
    assert gu.rel_err(db, bt.grad) < gtol
    # fused bias gradient of the branch: column sums of da, same pass, same dz/dgamma/dbeta
    dz3, da3, dg3, db3, dsum = ops.ln_residual_bwd(dy.cuda(), dy2.cuda(), z, mean, rstd, gamma.cuda(), want_dsum=True)
    assert torch.equal(dz3, dz) and torch.equal(dg3, dg) and torch.equal(db3, db)
    assert gu.rel_err(dsum, a_r.grad.sum(0)) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    # inference form: no saved tensors, same y
    y2, z2, _, _ = ops.ln_residual_fwd(a.cuda(), None if res is None else res.cuda(), gamma.cuda(), beta.cuda(), save=False)
    assert z2 is None
    if dtype == torch.float32:
        assert torch.equal(y2, y)


def test_layernorm_dropout_mask_consistent():
    """Dropout: forward and backward regenerate the same mask from (seed, index);
    kept fraction ~ 1-p; kept elements scaled by 1/(1-p)."""
    ops = _ops()
    T, d, p, seed = 64, 512, 0.25, 1234567
    a = torch.ones(T, d).cuda()
    res = torch.zeros(T, d).cuda()
    gamma, beta = torch.ones(d).cuda(), torch.zeros(d).cuda()
    y, z, mean, rstd = ops.ln_residual_fwd(a, res, gamma, beta, 1e-5, p, seed)
    kept = z > 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
    assert torch.allclose(z[kept], torch.full_like(z[kept], 1 / (1 - p)))
    dz, da, _, _ = ops.ln_residual_bwd(torch.randn(T, d).cuda(), None, z, mean, rstd, gamma, p, seed)
    assert torch.equal(da != 0, kept & (dz != 0))
    assert torch.allclose(da[kept], dz[kept] / (1 - p))
    *_, dsum = ops.ln_residual_bwd(torch.randn(T, d, generator=torch.Generator().manual_seed(1)).cuda(), None, z, mean,
                                   rstd, gamma, p, seed, want_dsum=True)
    _, da2, _, _ = ops.ln_residual_bwd(torch.randn(T, d, generator=torch.Generator().manual_seed(1)).cuda(), None, z, mean,
                                       rstd, gamma, p, seed)
    assert torch.allclose(dsum, da2.sum(0), rtol=1e-4, atol=1e-4)      # sums the masked gradient, not dz


@pytest.mark.parametrize("T,n", [(50, 128), (33, 4096), (257, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "relu"])
def test_bias_act(T, n, dtype, act):
    ops = _ops()
    g = torch.Generator().manual_seed(T + n)
    u = torch.randn(T, n, generator=g).to(dtype)
    bias = 0.5 * torch.randn(n, generator=g)
    dh = torch.randn(T, n, generator=g).to(dtype)
    u_r = u.float().requires_grad_(True)
    b_r = bias.clone().requires_grad_(True)
    pre_ref = u_r + b_r
    if dtype == torch.bfloat16:
        pre_ref = pre_ref + (pre_ref.bfloat16().float() - pre_ref).detach()
    ref = O._ACT[act](pre_ref)
    ref.backward(dh.float())
    pre, h = ops.bias_act_fwd(u.cuda().clone(), bias.cuda(), act, inplace_pre=False)
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(h.float(), ref) < tol
    dpre, dbias = ops.bias_act_bwd(dh.cuda(), pre, act)
    assert gu.rel_err(dpre.float(), u_r.grad) < tol
    assert gu.rel_err(dbias, b_r.grad) < (2e-5 if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("T,V", [(96, 618), (40, 92), (300, 203), (5, 130)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("weighted", [False, True])
def test_softmax_xent(T, V, dtype, weighted):
    ops = _ops()
    g = torch.Generator().manual_seed(T + V)
    logits = (3 * torch.randn(T, V, generator=g)).to(dtype)
    y = torch.randint(0, V, (T,), generator=g)
    y[::7] = 1
    y[3] = 2
    w = None
    if weighted:
        w = torch.ones(V)
        w[1], w[2] = 0.0, 0.05
    ref_in = logits.float().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, y, w, ignore_index=-100 if weighted else 1)
    (ref * 1.7).backward()
    x = logits.cuda().requires_grad_(True)
    loss = ops.softmax_cross_entropy(x, y.cuda(), None if w is None else w.cuda(), None if weighted else 1)
    (loss * 1.7).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(x.grad.float(), ref_in.grad) < tol


def test_softmax_xent_column_slices():
    """Three heads in one [T, 425] logits buffer (separated model layout)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    T = 64
    logits = torch.randn(T, 425, generator=g)
    ys = [torch.randint(0, v, (T,), generator=g) for v in (92, 130, 203)]
    x = logits.cuda().requires_grad_(True)
    ref_in = logits.clone().requires_grad_(True)
    total, ref_total = 0, 0
    for (c0, V), y in zip(((0, 92), (92, 130), (222, 203)), ys):
        total = total + ops.softmax_cross_entropy(x, y.cuda(), None, 1, col0=c0, V=V)
        ref_total = ref_total + torch.nn.functional.cross_entropy(ref_in[:, c0:c0 + V], y, ignore_index=1)
    total.backward()
    ref_total.backward()
    assert abs(float(total) - float(ref_total)) < 1e-5 * abs(float(ref_total))
    assert gu.rel_err(x.grad, ref_in.grad) < F32_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,dh,R,pos0", [(3, 8, 8, 3, 0), (2, 4, 16, 2, 37), (2, 8, 64, 3, 200), (1, 8, 128, 1, 511),
                                           (2, 2, 128, 4, 300)])
def test_decode_attention(dtype, B, H, dh, R, pos0):
    """Rows pos0..pos0+R-1 against the cache == the same rows of full causal attention."""
    ops = _ops()
    L = pos0 + R
    Lmax = L + 5
    d = H * dh
    g = torch.Generator().manual_seed(pos0 + dh)
    qkv = torch.randn(B, L, 3 * d, generator=g).to(dtype)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32)
    ref = _ref_attn(qkv.float(), H)[:, pos0:]
    kc = torch.zeros(B, H, Lmax, dh, dtype=dtype).cuda()
    vc = torch.zeros_like(kc)
    qkv_g = qkv.cuda()
    if pos0 > 0:
        ops.kv_append(qkv_g[:, :pos0].contiguous(), kc, vc, 0)
    ops.kv_append(qkv_g[:, pos0:].contiguous(), kc, vc, pos0)
    k_ref = O._heads(qkv[:, :, d:2 * d].float(), H)
    assert torch.equal(kc[:, :, :L].float().cpu(), k_ref)
    q = qkv_g[:, pos0:, :d].contiguous()
    out = ops.decode_attn(q, kc, vc, slopes.cuda(), pos0)
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    assert gu.rel_err(out.float(), ref) < tol


@pytest.mark.parametrize("B,H,dh,R,pos0", [(64, 8, 128, 3, 700), (2, 8, 128, 3, 0), (3, 8, 64, 2, 1), (64, 8, 64, 4, 129),
                                           (1, 8, 128, 3, 2045), (4, 8, 128, 3, 4158), (2, 8, 64, 3, 4200)])
def test_decode_attention_streaming_bf16(B, H, dh, R, pos0):
    """The bf16 streaming kernel (both workgroup widths, every R), host- and device-side positions,
    against full causal attention in fp32 on the same bf16-rounded inputs."""
    ops = _ops()
    L = pos0 + R
    Lmax = L + 3
    d = H * dh
    g = torch.Generator().manual_seed(pos0 + dh + B)
    qkv = torch.randn(B, L, 3 * d, generator=g).to(torch.bfloat16).cuda()
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32).cuda()
    # fp32 reference on the GPU (torch ops only; the CPU oracle is too slow at B=64)
    q, k, v = (t.view(B, L, H, dh).transpose(1, 2).float() for t in qkv.split(d, -1))
    idx = torch.arange(L, device="cuda")
    bias = slopes.view(H, 1, 1) * (idx.view(1, 1, L) - idx.view(1, L, 1))
    sc = q[:, :, pos0:] @ k.transpose(-1, -2) / dh ** 0.5 + bias[:, pos0:]
    sc = sc.masked_fill(idx.view(1, 1, 1, L) > idx[pos0:].view(1, 1, R, 1), float("-inf"))
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(B, R, d)
    kc = torch.zeros(B, H, Lmax, dh, dtype=torch.bfloat16).cuda()
    vc = torch.zeros_like(kc)
    if pos0 > 0:
        ops.kv_append(qkv[:, :pos0].contiguous(), kc, vc, 0)
    pos_dev = torch.tensor([pos0], dtype=torch.int32).cuda()
    ops.kv_append(qkv[:, pos0:].contiguous(), kc, vc, pos_dev)
    assert torch.equal(kc[:, :, :L], qkv[:, :, d:2 * d].view(B, L, H, dh).transpose(1, 2))
    qn = qkv[:, pos0:, :d].contiguous()
    out = ops.decode_attn(qn, kc, vc, slopes, pos0)
    out_dev = ops.decode_attn(qn, kc, vc, slopes, pos_dev)
    assert torch.equal(out, out_dev)
    # fused step (append + attend in one launch) vs the two separate calls: same caches; the outputs agree to
    # bf16 rounding (the new keys are folded in by another wave, so fp32 sums associate differently)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    if pos0 > 0:
        ops.kv_append(qkv[:, :pos0].contiguous(), kc2, vc2, 0)
    out_f = ops.decode_step_attn(qkv[:, pos0:].contiguous(), kc2, vc2, slopes, pos0)
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc)
    assert gu.rel_err(out_f.float(), out.float()) < 1e-2 and gu.rel_err(out_f.float(), ref) < BF16_TOL
    kc2[:, :, pos0:] = 0
    out_fd = ops.decode_step_attn(qkv[:, pos0:].contiguous(), kc2, vc2, slopes, pos_dev)
    assert torch.equal(out_fd, out_f) and torch.equal(kc2, kc)
    assert gu.rel_err(out.float(), ref) < BF16_TOL
    assert float((out.float() - ref).abs().max()) < 0.03 * float(ref.abs().max())


def test_adamw_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(1000, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=1e-4)
    p = p0.clone().cuda()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    shadow = torch.empty(1000, dtype=torch.bfloat16).cuda()
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adamw_step(p, grad.cuda(), m, v, shadow, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step)
    assert float((p.cpu() - ref.detach()).abs().max()) < 1e-6
    assert torch.equal(shadow, p.bfloat16())


def test_adamw_multi_matches_torch():
    """Multi-tensor launch (mgx_adamw_multi): ragged sizes incl. one that is not a multiple of 4 and more
    tensors than one launch holds; three steps against torch.optim.AdamW."""
    ops = _ops()
    g = torch.Generator().manual_seed(10)
    sizes = [618, 1000, 4096 * 3 + 4, 7, 1024 * 1024] + [64 + 4 * i for i in range(35)]
    p0 = [torch.randn(n, generator=g) for n in sizes]
    refs = [t.clone().requires_grad_(True) for t in p0]
    opt = torch.optim.AdamW(refs, lr=1e-3, weight_decay=1e-4)
    ps = [t.clone().cuda() for t in p0]
    ms = [torch.zeros_like(t) for t in ps]
    vs = [torch.zeros_like(t) for t in ps]
    for step in range(1, 4):
        grads = [torch.randn(n, generator=g) for n in sizes]
        for r, gr in zip(refs, grads):
            r.grad = gr.clone()
        opt.step()
        # gradients as views at odd element offsets of one buffer (the slices of a fused-head gradient look like
        # this): the kernel must not assume 16-byte alignment
        flat = torch.empty(sum(sizes) + len(sizes), device="cuda")
        gv, off = [], 1
        for n, gr in zip(sizes, grads):
            gv.append(flat[off:off + n].copy_(gr))
            off += n + 1
        ops.adamw_step_multi(ps, gv, ms, vs, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step)
    for p, r in zip(ps, refs):
        assert float((p.cpu() - r.detach()).abs().max()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_qk_matches_reference_formula_and_inverts(dtype):
    from oracle import reference_port as O
    ops = _ops()
    B, S, H, dh = 2, 37, 3, 16
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).to(dtype)
    cos, sin = O.rotary_tables(S, dh)
    q, k, v = (t.reshape(B, S, H, dh).permute(0, 2, 1, 3).float() for t in qkv.split(H * dh, -1))
    want = torch.cat([O.apply_rotary(t, cos, sin).permute(0, 2, 1, 3).reshape(B, S, H * dh) for t in (q, k)]
                     + [v.permute(0, 2, 1, 3).reshape(B, S, H * dh)], -1)
    got = ops.rope_qk_(qkv.clone().cuda(), cos.cuda(), sin.cuda(), H)
    tol = 1e-6 if dtype == torch.float32 else 4e-2          # bf16: one rounding of values up to ~5
    assert float((got.float().cpu() - want).abs().max()) < tol
    assert torch.equal(got[..., 2 * H * dh:].cpu(), qkv[..., 2 * H * dh:])              # v untouched
    back = ops.rope_qk_(got.clone(), cos.cuda(), sin.cuda(), H, inverse=True)
    assert float((back.float().cpu() - qkv.float()).abs().max()) < (1e-5 if dtype == torch.float32 else 8e-2)


@pytest.mark.parametrize("M,N,K", [(192, 3072, 1024), (192, 1024, 4096), (3, 1024, 1024), (256, 4096, 1024),
                                   (50, 128, 192), (17, 64, 64), (100, 256, 2048), (192, 1024, 1024), (65, 8192, 512)])
@pytest.mark.parametrize("mode", ["bias", "gelu", "relu", "ln"])
def test_skinny_gemm(M, N, K, mode):
    """Decode projections (mgx_skinny_gemm_bf16) against fp32 torch on the same bf16 operands."""
    ops = _ops()
    if mode == "ln" and N > 2048:
        pytest.skip("LayerNorm epilogue covers N <= 2048")
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = 0.1 * torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).bfloat16()
    gamma, beta = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    ws = torch.empty(ops.skinny_workspace_floats(M, N, K), dtype=torch.float32).cuda()
    pre = (x.float() @ wt.float().t() + bias).bfloat16().float()
    if mode == "ln":
        out = ops.skinny_gemm(x.cuda(), wt.cuda(), bias.cuda(), ws, res=res.cuda(), gamma=gamma.cuda(), beta=beta.cuda())
        ref = torch.nn.functional.layer_norm(pre + res.float(), (N,), gamma, beta, 1e-5)
    else:
        act = None if mode == "bias" else mode
        out = ops.skinny_gemm(x.cuda(), wt.cuda(), bias.cuda(), ws, act=act)
        ref = pre if act is None else getattr(torch.nn.functional, act)(pre)
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    # one bf16 rounding of the result + accumulation-order noise on the pre-activation's own rounding
    assert float((out.float().cpu() - ref).abs().max()) < 0.03 * float(ref.abs().max().clamp_min(1.0))
    assert gu.rel_err(out.float(), ref) < BF16_TOL


def test_bad_arguments_fail_loudly():
    ops = _ops()
    from musicgeneration_amd._lib import MgxError
    with pytest.raises(RuntimeError):
        ops.alibi_attn_fwd(torch.zeros(1, 4, 24), torch.ones(1), 1)            # CPU tensor: no fallback
    with pytest.raises(MgxError):
        ops.ln_residual_fwd(torch.zeros(4, 6).cuda(), None, torch.ones(6).cuda(), torch.zeros(6).cuda())  # d % 4
    with pytest.raises(ValueError):
        ops.alibi_attn_fwd(torch.zeros(1, 4, 25).cuda(), torch.ones(1).cuda(), 1)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 264, 128), (1000, 3072, 1024), (77, 624, 512), (512, 1024, 4096)])
@pytest.mark.parametrize("epi", ["none", "bias", "bias_res", "gelu", "relu"])
def test_gemm_nt(M, N, K, epi):
    """Hand-written NT GEMM vs fp32 matmul on the bf16-rounded operands; asymmetric random data."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    bt = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g) if epi != "none" else None
    res = torch.randn(M, N, generator=g).bfloat16() if epi == "bias_res" else None
    ref = a.float() @ bt.float().t()
    if bias is not None:
        ref = ref + bias
    if res is not None:
        ref = ref + res.float()
    act = epi if epi in ("gelu", "relu") else None
    out = ops.gemm_nt(a.cuda(), bt.cuda(), None if bias is None else bias.cuda(), None if res is None else res.cuda(), act)
    if act is None:
        assert gu.rel_err(out.float(), ref) < 1e-2
    else:
        fac, h = out                                  # derivative factor (for the backward) and activated output
        pre = ref.bfloat16().float().requires_grad_(True)
        y = O._ACT[act](pre)
        y.backward(torch.ones_like(y))
        assert gu.rel_err(h.float(), y.detach()) < 1e-2
        # the derivative factor is stored as one byte per element: exact for ReLU, within half a step (0.0025) of
        # GELU'(pre) - the size of the bf16 rounding of a factor near 1
        assert fac.dtype == torch.uint8
        got = ops.decode_fac(fac, act).cpu()
        if act == "relu":
            assert torch.equal(got, pre.grad)
        else:
            # (+ 0.004: the kernel's pre-activation may sit one bf16 ulp from this reference's, |GELU''| <= 0.5)
            assert float((got - pre.grad).abs().max()) <= 0.5 * ops.FAC_STEP["gelu"] + 0.004
            assert float((got - pre.grad).abs().mean()) <= 0.3 * ops.FAC_STEP["gelu"]
            assert gu.rel_err(got, pre.grad) < 1e-2


def test_gemm_nt_exact_integers():
    """Small-integer operands: every product and sum is exact in bf16/fp32, so any swapped fragment
    index or k-order mismatch is an exact mismatch."""
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    M, N, K = 512, 520, 192
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    bt = torch.randint(-3, 4, (N, K), generator=g).float()
    out = ops.gemm_nt(a.bfloat16().cuda(), bt.bfloat16().cuda())
    assert torch.equal(out.float().cpu(), (a @ bt.t()).bfloat16().float())


def test_gemm_nt_dropout_matches_bias_act_mask():
    """The fused epilogue drops exactly the elements mgx_bias_act_bwd regenerates from (seed, index), in the activated
    output and in the derivative factor alike."""
    ops = _ops()
    M, N, K, p, seed = 256, 512, 64, 0.25, 987654321
    a = torch.ones(M, K).bfloat16().cuda()
    bt = (torch.ones(N, K) / K).bfloat16().cuda()
    fac, h = ops.gemm_nt(a, bt, torch.zeros(N).cuda(), None, "relu", p, seed)
    kept = h != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
    val = ops.decode_fac(fac, "relu")                  # codes -> act'(pre) * mask (the 1 / (1 - p) is the consumer's)
    assert torch.equal(val != 0, kept)
    assert torch.equal(val[kept], torch.ones_like(val[kept])) and torch.equal(fac[~kept], torch.full_like(fac[~kept], ops.FAC_ZERO))
    # the dgrad epilogue that consumes the codes applies the same mask and the keep scale exactly
    dy = torch.ones(M, 64).bfloat16().cuda()
    w2t = (torch.ones(N, 64) / 64).bfloat16().cuda()
    dfused, _ = ops.gemm_nt_dfac(dy, w2t, fac, "relu", p)
    assert torch.equal(dfused != 0, kept)
    assert gu.rel_err(dfused[kept].float(), torch.full_like(dfused[kept].float(), 1 / (1 - p))) < 4e-3
    pre = torch.ones(M, N).bfloat16().cuda()
    dpre, _ = ops.bias_act_bwd(torch.ones(M, N).bfloat16().cuda(), pre, "relu", p, seed)
    assert torch.equal(dpre != 0, kept)


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_gemm_nt_gelu_epilogue_at_the_edges_of_its_table(p):
    """The GELU epilogue reads gelu / gelu' of the bf16 pre-activation from an LDS table (2^-20 <= |x| < 128) and redoes
    a block arithmetically when it meets anything else.  Pre-activations set through the bias (x = 0): zeros, values
    below, inside and above the table, its first and last rows, inf and nan - every one must come out as F.gelu of the
    rounded pre-activation (to a bf16 ulp), infinities and NaNs included, with the dropout of (seed, index) on top."""
    ops = _ops()
    M, N, K, seed = 300, 264, 64, 31337
    g = torch.Generator().manual_seed(9)
    bias = torch.randn(N, generator=g)
    special = [0.0, -0.0, 1e-9, -3e-8, 2e-6, 2.0 ** -20, -(2.0 ** -20), 0.9e-6, 300.0, -300.0, 100.0, -100.0, 127.5, 128.0,
               -128.0, float("inf"), float("-inf"), float("nan"), 1.0, -1.0, 0.5, -0.75, 3.0, -3.0, 5.5, -5.5, 15.9, -16.0, 1e30]
    bias[:len(special)] = torch.tensor(special)
    bias[100:100 + len(special)] = torch.tensor(special)            # again in another 16-column group of the tile
    x = torch.zeros(M, K).bfloat16().cuda()
    w = torch.randn(N, K, generator=g).bfloat16().cuda()
    fac, h = ops.gemm_nt(x, w, bias.cuda(), None, "gelu", p, seed)
    pre = bias.bfloat16().float().expand(M, N).clone().requires_grad_(True)
    y = torch.nn.functional.gelu(pre)
    torch.nan_to_num(y, nan=0.0, posinf=0.0, neginf=0.0).sum().backward()
    y, x0 = y.detach(), pre.detach()
    hf, codes = h.float().cpu(), fac.cpu()
    # a dropped element: code ZERO and +-0 (a kept one that looks the same - x <= -16 - is zero either way)
    keep = ~((codes == ops.FAC_ZERO) & (hf == 0))
    if p > 0:
        _, h_un = ops.bias_act_fwd(x0.bfloat16().cuda(), None, "gelu", p, seed)        # the mask of (seed, index)
        nz = torch.isfinite(y) & (y != 0)
        assert torch.equal((hf != 0)[nz], (h_un.float().cpu() != 0)[nz])
        assert abs(float(keep[nz].float().mean()) - (1 - p)) < 0.03
        hf = hf * (1 - p)
    else:
        assert bool(keep[torch.isfinite(y) & (y != 0)].all())
    # inf and nan stay visible (gelu(+inf) is +inf here and nan in torch's CPU kernel: non-finite either way)
    assert torch.equal(~torch.isfinite(hf) & keep, ~torch.isfinite(y) & keep)
    assert bool(torch.isnan(hf[torch.isnan(x0) & keep]).all())
    ok = torch.isfinite(y) & keep
    # bf16 rounding of the result + the 1.5e-7 absolute error of the erf formula
    tol = (2.0 ** -7) * y.abs() + 4e-7 * x0.abs().clamp(max=16.0)
    assert bool(((hf - y).abs()[ok] <= tol[ok]).all())
    big = (x0 >= 128) & ok
    assert torch.equal((hf[big] / (1 - p) if p > 0 else hf[big]).bfloat16(), (x0[big] * (1 / (1 - p) if p > 0 else 1.0)).bfloat16())
    assert float(hf[(x0 <= -16) & torch.isfinite(x0)].abs().max()) == 0.0
    got = ops.decode_fac(fac, "gelu").cpu()
    fin_g = ok & torch.isfinite(pre.grad) & torch.isfinite(x0)
    assert float((got - pre.grad).abs()[fin_g].max()) <= 0.5 * ops.FAC_STEP["gelu"] + 1e-3
    assert float(got[~keep].abs().max() if bool((~keep).any()) else 0.0) == 0.0


@pytest.mark.parametrize("K,M,N", [(64, 256, 256), (1000, 264, 520), (8192, 1024, 3072), (333, 64, 128)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_tn(K, M, N, out_dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(K, M, generator=g).bfloat16()
    b = (torch.randn(K, N, generator=g) / K ** 0.5).bfloat16()
    ref = a.float().t() @ b.float()
    out = ops.gemm_tn(a.cuda(), b.cuda(), out_dtype)
    assert out.dtype == out_dtype
    assert gu.rel_err(out.float(), ref) < (1e-4 if out_dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("K,M0,N0,M1,N1", [(8192, 1024, 4096, 4096, 1024), (8192, 1024, 1024, 1024, 3072), (1000, 264, 520, 64, 128),
                                            (333, 64, 128, 264, 256)])
def test_gemm_tn_pair(K, M0, N0, M1, N1):
    """Two weight gradients over the same token rows in one launch (shared K splits) against fp32 matmuls on the
    bf16-rounded operands and against the single-problem entry point; bitwise reproducible; writes into given outputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(K + M0 + N1)
    a0 = torch.randn(K, M0, generator=g).bfloat16().cuda()
    b0 = (torch.randn(K, N0, generator=g) / K ** 0.5).bfloat16().cuda()
    a1 = torch.randn(K, M1, generator=g).bfloat16().cuda()
    wide = (torch.randn(K, N1 + 64, generator=g) / K ** 0.5).bfloat16().cuda()
    b1 = wide[:, 8:8 + N1]                                      # a column slice of a wider matrix (row stride != N)
    out1 = torch.full((M1, N1), float("nan"), device="cuda")
    c0, c1 = ops.gemm_tn_pair(a0, b0, a1, b1, None, out1)
    assert c1.data_ptr() == out1.data_ptr()
    assert gu.rel_err(c0, a0.float().t() @ b0.float()) < 1e-4
    assert gu.rel_err(c1, a1.float().t() @ b1.float()) < 1e-4
    assert gu.rel_err(c0, ops.gemm_tn(a0, b0)) < 1e-5
    assert gu.rel_err(c1, ops.gemm_tn(a1, b1.contiguous())) < 1e-5
    d0, d1 = ops.gemm_tn_pair(a0, b0, a1, b1)
    assert torch.equal(c0, d0) and torch.equal(c1, d1)


def test_gemm_tn_exact_integers():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    K, M, N = 700, 520, 264
    a = torch.randint(-3, 4, (K, M), generator=g).float()
    b = torch.randint(-3, 4, (K, N), generator=g).float()
    out = ops.gemm_tn(a.bfloat16().cuda(), b.bfloat16().cuda(), torch.float32)
    assert torch.equal(out.cpu(), a.t() @ b)


@pytest.mark.parametrize("shapes", [[(64, 64)], [(1024, 3072), (1024, 1024), (4096, 1024), (1024, 4096)],
                                    [(128, 192)] * 19])
def test_cast_shadow(shapes):
    """mgx_cast_shadow_bf16: both bf16 orientations of fp32 masters, bit-exact against Tensor.to / .t().contiguous()
    (round to nearest even), several matrices per launch, either destination optional."""
    ops = _ops()
    g = torch.Generator().manual_seed(len(shapes))
    srcs = [(torch.randn(r, c, generator=g) * 3).cuda() for r, c in shapes]
    srcs[0].view(-1)[:4] = torch.tensor([1.00390625, 1.01171875, -0.0, 3.3895e38])     # ties, signed zero, near max
    dst = [torch.full((r, c), 7.0, dtype=torch.bfloat16).cuda() for r, c in shapes]
    dst_t = [torch.full((c, r), 7.0, dtype=torch.bfloat16).cuda() for r, c in shapes]
    ops.cast_shadow(srcs, dst, dst_t)
    for s, d, t in zip(srcs, dst, dst_t):
        want = s.to(torch.bfloat16)
        assert torch.equal(d.view(torch.int16), want.view(torch.int16))
        assert torch.equal(t.view(torch.int16), want.t().contiguous().view(torch.int16))
    only_t = [torch.zeros_like(t) for t in dst_t]
    ops.cast_shadow(srcs, [None] * len(srcs), only_t)
    only_d = [torch.zeros_like(d) for d in dst]
    ops.cast_shadow(srcs, only_d, [None] * len(srcs))
    for a, b in zip(only_t + only_d, dst_t + dst):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    with pytest.raises(RuntimeError):
        ops.cast_shadow([torch.zeros(60, 64).cuda()], [torch.zeros(60, 64, dtype=torch.bfloat16).cuda()], [None])


@pytest.mark.parametrize("seq_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("B,d,V", [(5, 1024, 618), (64, 1024, 618), (3, 64, 40), (2, 2048, 700)])
def test_decode_sampling_tail(B, d, V, seq_dtype):
    """mgx_decode_gather_embed / _head_argmax / _commit (csrc/sample.hip) against the torch ops they replace in one
    greedy step: gather + embedding + cast (bit-exact), Norm + proj_class logits (bf16 tolerance), argmax with the
    lowest index on ties (exact on the kernel's own logits), scatter at pos+1 and the position advance (exact)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + d + V)
    L, R, pos0 = 37, 3, 11
    seq = torch.randint(0, V, (B, L), generator=g).to(seq_dtype).cuda()
    table = torch.randn(V, d, generator=g).cuda()
    pos = torch.tensor([pos0], dtype=torch.int32).cuda()
    emb = ops.decode_gather_embed(seq, pos, table, R)
    want = torch.nn.functional.embedding(seq[:, pos0:pos0 + R].long(), table).to(torch.bfloat16)
    assert torch.equal(emb.view(torch.int16), want.view(torch.int16))

    h = (torch.randn(B, R, d, generator=g) * 2).bfloat16().cuda()
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
    w = (torch.randn(V, d, generator=g) / d ** 0.5).bfloat16().cuda()
    w[7] = w[3]                                               # an exact tie between classes 3 and 7
    bias = (0.1 * torch.randn(V, generator=g)).cuda()
    bias[7] = bias[3]
    keys = torch.zeros(B, dtype=torch.int64).cuda()
    logits = torch.empty(B, V, dtype=torch.float32).cuda()
    ops.decode_head_argmax(h, gamma, beta, 1e-5, w, bias, keys, logits)
    y = torch.nn.functional.layer_norm(h[:, R - 1].float(), (d,), gamma, beta, 1e-5).bfloat16().float()
    ref = (y @ w.float().t() + bias).bfloat16().float()
    assert gu.rel_err(logits, ref) < BF16_TOL
    assert torch.equal(logits[:, 3], logits[:, 7])
    before = seq.clone()
    ops.decode_commit(seq, pos, keys)
    torch.cuda.synchronize()
    assert int(pos) == pos0 + 1
    assert torch.equal(seq[:, pos0 + 1].long(), torch.argmax(logits, -1))      # first maximum, as torch
    keep = torch.ones(L, dtype=torch.bool)
    keep[pos0 + 1] = False
    assert torch.equal(seq[:, keep], before[:, keep])
    assert int(keys.abs().sum()) == 0
    # the tie itself decides a row: push classes 3 and 7 to the top
    bias2 = bias.clone()
    bias2[3] += 100.0
    bias2[7] += 100.0
    ops.decode_head_argmax(h, gamma, beta, 1e-5, w, bias2, keys)
    ops.decode_commit(seq, pos, keys)
    assert torch.equal(seq[:, pos0 + 2].long().cpu(), torch.full((B,), 3))


@pytest.mark.parametrize("seq_dtype", [torch.int32, torch.int64])
def test_sampling_tail_three_streams(seq_dtype):
    """mgx_sample_gather_embed / _head / _commit with the separated model's three token streams
    (inference_separated.py:18-28): three gathers + cat + cast (bit-exact), Norm + the three stacked heads (bf16
    tolerance), one argmax per head over that head's classes only (exact on the kernel's logits), scatter into
    seq[b, k, pos+1] (exact)."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    B, L, R, pos0 = 9, 41, 3, 17
    sizes, dims = (92, 130, 203), (256, 256, 512)
    d = sum(dims)
    seq = torch.stack([torch.randint(0, v, (B, L), generator=g) for v in sizes], 1).to(seq_dtype).cuda()
    tables = [torch.randn(v, dk, generator=g).cuda() for v, dk in zip(sizes, dims)]
    pos = torch.tensor([pos0], dtype=torch.int32).cuda()
    emb = ops.sample_gather_embed(seq, pos, tables, R)
    want = torch.cat([torch.nn.functional.embedding(seq[:, k, pos0:pos0 + R].long(), t) for k, t in enumerate(tables)],
                     -1).to(torch.bfloat16)
    assert torch.equal(emb.view(torch.int16), want.view(torch.int16))
    h = (torch.randn(B, R, d, generator=g) * 2).bfloat16().cuda()
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).cuda(), (0.1 * torch.randn(d, generator=g)).cuda()
    V = sum(sizes)
    w = (torch.randn(V, d, generator=g) / d ** 0.5).bfloat16().cuda()
    bias = (0.1 * torch.randn(V, generator=g)).cuda()
    keys = torch.zeros(B, 3, dtype=torch.int64).cuda()
    logits = torch.empty(B, V, dtype=torch.float32).cuda()
    ops.sample_head(h, gamma, beta, 1e-5, w, bias, sizes, keys, logits=logits)
    y = torch.nn.functional.layer_norm(h[:, R - 1].float(), (d,), gamma, beta, 1e-5).bfloat16().float()
    ref = (y @ w.float().t() + bias).bfloat16().float()
    assert gu.rel_err(logits, ref) < BF16_TOL
    before = seq.clone()
    ops.sample_commit(seq, pos, keys)
    torch.cuda.synchronize()
    assert int(pos) == pos0 + 1 and int(keys.abs().sum()) == 0
    off = 0
    for k, v in enumerate(sizes):
        assert torch.equal(seq[:, k, pos0 + 1].long(), torch.argmax(logits[:, off:off + v], -1)), k
        off += v
    keep = torch.ones(L, dtype=torch.bool)
    keep[pos0 + 1] = False
    assert torch.equal(seq[:, :, keep], before[:, :, keep])


def test_sampling_stochastic_draw_follows_the_softmax():
    """The stochastic branch (inference_unified.py:16-17: multinomial(softmax(logits), 1)) as a counter-based
    Gumbel-max draw: over 64 identical sequences x 400 positions the class frequencies of each head match
    softmax(that head's logits) (chi-square over the classes with >= 5 expected draws, 4.5 sigma), a re-run with the
    same seed and positions reproduces every draw, another seed does not, and sequences / positions are independent
    (not all equal)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, R, d = 64, 3, 256
    sizes = (24, 40)
    V = sum(sizes)
    h = torch.randn(1, R, d, generator=g).bfloat16().expand(B, R, d).contiguous().cuda()
    gamma, beta = torch.ones(d).cuda(), torch.zeros(d).cuda()
    w = (2.0 * torch.randn(V, d, generator=g) / d ** 0.5).bfloat16().cuda()
    bias = torch.zeros(V).cuda()
    keys = torch.zeros(B, 2, dtype=torch.int64).cuda()
    logits = torch.empty(B, V, dtype=torch.float32).cuda()
    seq = torch.zeros(B, 2, 402, dtype=torch.int32).cuda()

    def run(seed):
        seq.zero_()
        pos = torch.zeros(1, dtype=torch.int32).cuda()
        for _ in range(400):
            ops.sample_head(h, gamma, beta, 1e-5, w, bias, sizes, keys, pos=pos, stochastic=True, seed=seed, logits=logits)
            ops.sample_commit(seq, pos, keys)
        torch.cuda.synchronize()
        return seq[:, :, 1:401].clone()

    a = run(1234)
    assert torch.equal(a, run(1234))
    assert not torch.equal(a, run(99))
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[:, :, 0], a[:, :, 1])
    off = 0
    n = B * 400
    for k, v in enumerate(sizes):
        p = torch.softmax(logits[0, off:off + v].double(), -1).cpu()
        cnt = torch.bincount(a[:, k].reshape(-1).long().cpu(), minlength=v).double()
        assert int(cnt.sum()) == n and cnt.numel() == v
        big = p * n >= 5
        chi2 = float((((cnt - p * n) ** 2) / (p * n))[big].sum())
        dof = int(big.sum()) - 1
        assert dof >= 5 and chi2 < dof + 4.5 * (2 * dof) ** 0.5, (k, chi2, dof)
        off += v


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (130, 129, 17), (192, 618, 96), (257, 96, 1024), (1000, 425, 333)])
@pytest.mark.parametrize("trans_a,trans_b", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_generic(M, N, K, trans_a, trans_b):
    """mgx_gemm_generic (the fp32 parity mode's GEMM and the one for shapes off the tuned kernels' grid) against an
    fp64 product of the same values: fp32 storage to 2e-6, bf16 storage to the rounding of its output; both operand
    orientations, odd sizes, a strided operand, bias, an aliased residual, mixed input / output dtypes; bit-identical
    on a re-run."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if trans_a else (M, K), generator=g).cuda()
    wide = torch.randn((N, K + 5) if trans_b else (K, N + 5), generator=g).cuda()
    b = wide[:, :K] if trans_b else wide[:, :N]                  # row stride > width
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    A = (a.t() if trans_a else a).double()
    Bm = (b.t() if trans_b else b).double()
    want = A @ Bm + bias.double() + res.double()
    got = ops.gemm(a, b, trans_a, trans_b, bias=bias, res=res)
    assert got.dtype == torch.float32 and gu.rel_err(got, want) < 2e-6
    assert torch.equal(got, ops.gemm(a, b, trans_a, trans_b, bias=bias, res=res))
    inplace = res.clone()
    ops.gemm(a, b, trans_a, trans_b, bias=bias, res=inplace, out=inplace)
    assert torch.equal(inplace, got)
    ab, bb = a.bfloat16(), b.bfloat16()
    wantb = (ab.t() if trans_a else ab).double() @ (bb.t() if trans_b else bb).double()
    gotb = ops.gemm(ab, bb, trans_a, trans_b)
    assert gotb.dtype == torch.bfloat16 and gu.rel_err(gotb, wantb) < 4e-3
    got32 = ops.gemm(ab, bb, trans_a, trans_b, out_dtype=torch.float32)      # bf16 in, fp32 out: a weight gradient
    assert gu.rel_err(got32, wantb) < 2e-6


def test_linear_autograd_on_the_generic_gemm():
    """ops.linear (F.linear / Conv1d's x @ w + b on mgx_gemm_generic): output and the three gradients against torch's
    own fp64 autograd, both weight orientations."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    for nk in (True, False):
        x = torch.randn(3, 37, 96, generator=g).cuda().requires_grad_(True)
        w = torch.randn((50, 96) if nk else (96, 50), generator=g).cuda().requires_grad_(True)
        bvec = torch.randn(50, generator=g).cuda().requires_grad_(True)
        y = ops.linear(x, w, bvec, w_is_nk=nk)
        dy = torch.randn(y.shape, generator=g).cuda()
        y.backward(dy)
        xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, bvec))
        yd = xd @ (wd.t() if nk else wd) + bd
        yd.backward(dy.double())
        assert gu.rel_err(y, yd) < 2e-6
        for got, want in ((x.grad, xd.grad), (w.grad, wd.grad), (bvec.grad, bd.grad)):
            assert gu.rel_err(got, want) < 2e-6


@pytest.mark.parametrize("T,n", [(32768, 3072), (1000, 1544), (5, 8), (129, 512), (4097, 520)])
def test_colsum(T, n):
    """mgx_colsum_bf16 against an fp64 column sum of the same bf16 values; run twice for bitwise reproducibility."""
    ops = _ops()
    g = torch.Generator().manual_seed(T + n)
    x = torch.randn(T, n, generator=g).bfloat16().cuda()
    got = ops.colsum(x)
    ref = x.double().sum(0)
    assert float((got.double() - ref).abs().max()) < 1e-4 * max(1.0, T ** 0.5)
    assert torch.equal(got, ops.colsum(x))


@pytest.mark.parametrize("V,d,T", [(618, 1024, 4096), (92, 64, 100), (2000, 256, 777)])
def test_embedding_gemm_backward(V, d, T):
    """ops.embed_bf16: forward = F.embedding(...).to(bf16) bit for bit; backward (one-hot GEMM on gemm_tn_kernel)
    against torch's index_add-style embedding backward in fp32 on the same bf16 gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(V + T)
    table = torch.randn(V, d, generator=g).cuda().requires_grad_(True)
    tok = torch.randint(0, V, (T,), generator=g).cuda()
    tok[: T // 3] = 1                                            # a heavily repeated class (padding token)
    dh = torch.randn(T, d, generator=g).bfloat16().cuda()
    out = ops.embed_bf16(tok, table)
    assert torch.equal(out, torch.nn.functional.embedding(tok, table.detach()).bfloat16())
    out.backward(dh)
    ref = torch.zeros(V, d, dtype=torch.float64, device="cuda").index_add_(0, tok, dh.double())
    assert float((table.grad.double() - ref).abs().max()) < 1e-4 * max(1.0, (T / 3) ** 0.5)
    assert float(table.grad[tok.unique()].abs().sum()) > 0


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (300, 264, 192), (1024, 4096, 1024)])
@pytest.mark.parametrize("act", ["gelu", "relu"])
@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_gemm_nt_dfac(M, N, K, act, p_drop):
    """linear1 forward with the activation epilogue (derivative factor saved) + the dgrad GEMM that multiplies by it
    == plain GEMMs with mgx_bias_act_fwd / mgx_bias_act_bwd around them: same dropout mask, same activated output;
    the bias gradient equals the column sums of the rounded result."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w1 = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b1 = torch.randn(N, generator=g).cuda()
    dy = torch.randn(M, K, generator=g).bfloat16().cuda()
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    seed = 4242
    fac, h = ops.gemm_nt(x, w1, b1, None, act, p_drop, seed)
    assert fac.dtype == torch.uint8 and tuple(fac.shape) == (M, N)
    dpre, db = ops.gemm_nt_dfac(dy, wt, fac, act, p_drop)
    # the unfused chain on the same kernels' plain epilogues
    pre = ops.gemm_nt(x, w1, b1)
    _, h_want = ops.bias_act_fwd(pre, None, act, p_drop, seed)
    dact = ops.gemm_nt(dy, wt)
    want, want_db = ops.bias_act_bwd(dact, pre, act, p_drop, seed)
    assert gu.rel_err(h.float(), h_want.float()) < 2e-3            # two formulas of the same erf
    assert torch.equal(h == 0, h_want == 0) or float(((h == 0) != (h_want == 0)).float().mean()) < 1e-3
    assert gu.rel_err(dpre.float(), want.float()) < 1e-2          # bf16 rounding of act' and of the GEMM result
    assert float((dpre.float() - want.float()).abs().max()) <= 0.02 * float(want.float().abs().max())
    assert gu.rel_err(db, dpre.float().sum(0)) < 1e-4
    assert gu.rel_err(db, want_db) < 1e-2
    if p_drop > 0:
        # identical keep mask: whatever the unfused chain drops is exactly zero here; the byte code additionally
        # rounds |act'| < step / 2 (GELU' crosses zero near x = -0.75) to zero: well under 1 % of the elements
        assert float(dpre[want == 0].abs().max()) == 0.0 or float(((want == 0) & (dpre != 0)).float().mean()) < 1e-3
        assert float(((dpre == 0) & (want != 0)).float().mean()) < 1e-2
    # against the oracle formula in fp32
    xx = pre.float().cpu().requires_grad_(True)
    y = torch.nn.functional.gelu(xx) if act == "gelu" else torch.relu(xx)
    y.backward(torch.ones_like(y))
    if p_drop == 0:
        ref = dact.float().cpu() * xx.grad
        assert gu.rel_err(dpre.float(), ref) < BF16_TOL
        assert gu.rel_err(h.float(), y.detach()) < BF16_TOL


def test_embed_cat_matches_three_gathers():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, S = 3, 70
    tabs = [torch.randn(92, 256, generator=g).cuda().requires_grad_(True), torch.randn(130, 256, generator=g).cuda().requires_grad_(True),
            torch.randn(203, 512, generator=g).cuda().requires_grad_(True)]
    x = torch.stack([torch.randint(0, 92, (B, S), generator=g), torch.randint(0, 130, (B, S), generator=g),
                     torch.randint(0, 203, (B, S), generator=g)], 1).cuda()
    out = ops.embed_cat_bf16(x, tuple(tabs))
    want = torch.cat([torch.nn.functional.embedding(x[:, k], tabs[k]) for k in range(3)], -1).bfloat16()
    assert torch.equal(out, want)
    dh = torch.randn(B, S, 1024, generator=g).bfloat16().cuda()
    out.backward(dh)
    c0 = 0
    for k, t in enumerate(tabs):
        w = t.shape[1]
        ref = torch.zeros_like(t).index_add_(0, x[:, k].reshape(-1), dh[..., c0:c0 + w].reshape(-1, w).float())
        assert gu.rel_err(t.grad, ref) < 1e-5, k
        c0 += w


def test_softmax_xent_multi_equals_three_single_heads():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    T, sizes = 200, (92, 130, 203)
    ld = 448
    logits = torch.randn(T, ld, generator=g).bfloat16().cuda()
    ys = [torch.randint(0, v, (T,), generator=g).cuda() for v in sizes]
    ys[2][::7] = 1
    wn = torch.ones(92).cuda(); wn[1] = 0; wn[2] = 0.05
    wv = torch.ones(130).cuda(); wv[1] = 0; wv[2] = 0.05
    heads = ((ys[0], wn, None, 0, 92), (ys[1], wv, None, 92, 130), (ys[2], None, 1, 222, 203))
    coef = (1.0, 0.5, 2.0)
    a = logits.clone().requires_grad_(True)
    la = ops.softmax_cross_entropy_multi(a, heads)
    sum(c * l for c, l in zip(coef, la)).backward()
    b = logits.clone().requires_grad_(True)
    lb = [ops.softmax_cross_entropy(b, y, w, ii, col0=c0, V=v) for y, w, ii, c0, v in heads]
    sum(c * l for c, l in zip(coef, lb)).backward()
    for u, v in zip(la, lb):
        assert float(u) == float(v)
    assert torch.equal(a.grad, b.grad)
    assert float(a.grad[:, 425:].abs().max()) == 0.0


def test_torch_library_ops_registered_and_consistent():
    """torch.ops.mgx.* (musicgeneration_amd/torch_ops.py): schema + fake kernel + autograd registration pass
    torch.library.opcheck, and the differentiable operators give the wrappers' results and gradients."""
    from musicgeneration_amd import torch_ops  # noqa: F401
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, S, H, dh = 2, 96, 2, 64
    qkv = torch.randn(B, S, 3 * H * dh, generator=g).bfloat16().cuda().requires_grad_(True)
    slopes = torch.tensor(O.alibi_slopes(H), dtype=torch.float32).cuda()
    out, lse = torch.ops.mgx.alibi_attention(qkv, slopes, H)
    ref, ref_lse = ops.alibi_attn_fwd(qkv.detach(), slopes, H)
    assert torch.equal(out, ref) and torch.equal(lse, ref_lse)
    dout = torch.randn(B, S, H * dh, generator=g).bfloat16().cuda()
    out.backward(dout)
    assert torch.equal(qkv.grad, ops.alibi_attn_bwd(dout, qkv.detach(), ref, ref_lse, slopes, H))
    torch.library.opcheck(torch.ops.mgx.alibi_attention.default, (qkv.detach().requires_grad_(True), slopes, H),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    x = torch.randn(40, 256, generator=g).cuda().requires_grad_(True)
    gam, bet = torch.rand(256, generator=g).cuda().requires_grad_(True), torch.rand(256, generator=g).cuda().requires_grad_(True)
    torch.library.opcheck(torch.ops.mgx.layer_norm_fwd.default, (x, gam, bet, 1e-5),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    y = torch_ops.layer_norm(x, gam, bet)
    want = torch.nn.functional.layer_norm(x.detach(), (256,), gam.detach(), bet.detach())
    assert gu.rel_err(y, want) < F32_TOL
    a = torch.randn(130, 128, generator=g).bfloat16().cuda().requires_grad_(True)
    bt = (torch.randn(72, 128, generator=g) / 11).bfloat16().cuda().requires_grad_(True)
    bias = torch.randn(72, generator=g).cuda().requires_grad_(True)
    c = torch.ops.mgx.gemm_nt(a, bt, bias)
    want = a.detach().float() @ bt.detach().float().t() + bias.detach()
    assert gu.rel_err(c.float(), want) < BF16_TOL
    dc = torch.randn(130, 72, generator=g).bfloat16().cuda()
    c.backward(dc)
    assert gu.rel_err(a.grad.float(), dc.float() @ bt.detach().float()) < BF16_TOL
    assert gu.rel_err(bt.grad.float(), dc.float().t() @ a.detach().float()) < BF16_TOL
    assert gu.rel_err(bias.grad, dc.float().sum(0)) < 1e-4
