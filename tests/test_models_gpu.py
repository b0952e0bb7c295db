"""GPU parity of the drop-in modules (musicgeneration_amd) against the golden
vectors the real reference produced, and against the CPU oracle.

fp32: logits/loss/grads within 1e-5 (2e-5 for grads) relative to the tensor max.
bf16: logits within 2e-2.  Greedy decode: token sequences bit-exact in fp32.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import reference_port as O

pytestmark = pytest.mark.gpu


def _unified_model(z, sd, dtype=None, activation="gelu"):
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    V, d, L, H, M, ff = (int(v) for v in z["cfg"])
    m = AlibiTransformer(vocab=V, embed_dim=d, n_layers=L, n_heads=H, max_sequence=M, proj_forward=ff, dropout=0.0,
                         activation=activation, dtype=dtype)
    missing = m.load_state_dict(sd, strict=True)
    return m.cuda()


def _check_unified(name, seeded, activation="gelu"):
    from musicgeneration_amd.model_baseline.alibi_unified import Criterion
    z = gu.load(name)
    sd = gu.seeded_state_dict(z) if seeded else gu.state_dict(z)
    m = _unified_model(z, sd, activation=activation)
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    m.train()
    logits = m(x)
    assert logits.dtype == torch.float32
    assert gu.rel_err(logits, z["logits"]) < 1e-5
    loss, ld = Criterion(1).cuda()(logits, y)
    assert abs(float(loss.detach()) - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in gu.grads(z).items():
        assert gu.rel_err(named[k].grad, g) < 2e-5, k


@pytest.mark.parametrize("name", ["unified_tiny", "unified_heads6", "unified_heads12"])
def test_unified_fp32_small(name):
    _check_unified(name, False)


def test_unified_fp32_relu():
    _check_unified("unified_relu", False, activation="relu")


@pytest.mark.parametrize("name", ["unified_dh64_s256", "unified_dh128_s256", "unified_dh64_s320_l2"])
def test_unified_fp32_seeded(name):
    _check_unified(name, True)


@pytest.mark.parametrize("name", ["unified_dh64_s256", "unified_dh128_s256", "unified_dh64_s320_l2"])
def test_unified_bf16(name):
    """bf16 mode (MFMA flash attention for dh 64/128) within 2e-2 of the reference's fp32 logits."""
    from musicgeneration_amd.model_baseline.alibi_unified import Criterion
    z = gu.load(name)
    m = _unified_model(z, gu.seeded_state_dict(z), dtype="bf16")
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    logits = m(x)
    assert logits.dtype == torch.bfloat16
    assert gu.rel_err(logits.float(), z["logits"]) < 2e-2
    loss, _ = Criterion(1).cuda()(logits, y)
    assert abs(float(loss.detach()) - float(z["loss"])) < 2e-2 * abs(float(z["loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in gu.grads(z).items():
        assert named[k].grad.dtype == torch.float32
        assert gu.rel_err(named[k].grad, g) < 6e-2, k      # gradients: looser, informational bound
    # the autocast route selects bf16 as well
    m32 = _unified_model(z, gu.seeded_state_dict(z))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert m32(x).dtype == torch.bfloat16


@pytest.mark.parametrize("dtype,tol", [(None, 1e-5), ("bf16", 2e-2)])
def test_unified_build_config_full_depth(dtype, tol):
    """BASELINE configs[1]'s model at full depth and length (d=1024 L=6 H=8 ff=4096, S=2048, B=1) against the
    reference's own logits (tests/golden/make_golden_r2.py; stored: every 8th position + the last 128) and loss:
    fp32 within 1e-5, bf16 (MFMA flash attention, every projection on libmgx) within 2e-2."""
    from musicgeneration_amd.model_baseline.alibi_unified import Criterion
    z = gu.load("unified_build_l6_s2048")
    m = _unified_model(z, gu.seeded_state_dict(z), dtype=dtype).eval()
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    with torch.no_grad():
        logits = m(x)
        loss, _ = Criterion(1).cuda()(logits, y)
    rows = torch.from_numpy(z["rows"]).cuda()
    got = logits[0, rows].float().cpu()
    want = torch.from_numpy(z["logits_rows"])
    assert float((got - want).abs().max()) < tol * float(z["logits_absmax"])
    assert abs(float(loss) - float(z["loss"])) < tol * abs(float(z["loss"]))


@pytest.mark.parametrize("dtype,tol", [(None, 1e-5), ("bf16", 2e-2)])
def test_separated_build_width(dtype, tol):
    """BASELINE configs[2]'s width (embeddings 256/256/512 -> d=1024, 8 heads of 128), one layer, S=256, with the
    pad (1) and down-weighted (2) classes present: the three logit tensors and the four losses against the
    reference; bf16 runs the packed 425-column head, the weighted column-slice CE and the dh=128 MFMA attention."""
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer, Criterion
    z = gu.load("separated_dh128_s256")
    cfg = [int(v) for v in z["cfg"]]
    m = AlibiTransformer(*cfg, dropout=0.0, dtype=dtype)
    m.load_state_dict(gu.seeded_state_dict(z))
    m = m.cuda()
    crit = Criterion(92, 130, 203, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2}).cuda()
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    preds = m(x)
    for p, k in zip(preds, ("logit_n", "logit_v", "logit_d")):
        assert gu.rel_err(p.float(), z[k]) < tol, k
    loss, ld = crit(preds, y)
    for k in ("loss", "loss_notes", "loss_velocity", "loss_duration"):
        assert abs(float(ld[k].detach()) - float(z[k])) < tol * abs(float(z[k])), k
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in gu.grads(z).items():
        assert named[k].grad.dtype == torch.float32
        assert gu.rel_err(named[k].grad, g) < (2e-5 if dtype is None else 6e-2), k    # bf16 gradients: informational bound


def test_bf16_decode_trunk_matches_flash_forward_and_oracle():
    """KVCacheDecoder's bf16 trunk (skinny-GEMM projections + the streaming decode-attention kernel: what the decode
    bench times) at d=1024, three provisional rows against a context of 2100: against the model's own flash forward
    over the whole sequence and against the CPU oracle in fp32 (2e-2 on the logits, north_star's bf16 bar)."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    from musicgeneration_amd.decode import KVCacheDecoder
    torch.manual_seed(7)
    m = AlibiTransformer(vocab=618, embed_dim=1024, n_layers=2, n_heads=8, max_sequence=2304, proj_forward=4096,
                         dropout=0.0, dtype="bf16").cuda().eval()
    B, ctx, R = 2, 2100, 3
    seq = torch.randint(2, 618, (B, ctx + R), generator=torch.Generator().manual_seed(8)).cuda()
    dec = KVCacheDecoder(m, B, ctx + R)
    with torch.no_grad():
        dec.rows(seq[:, :ctx], 0)
        part = dec.rows(seq[:, ctx:], ctx).float()
        full = m(seq)[:, ctx:].float()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = O.unified_forward(sd, seq.cpu(), 8)[:, ctx:]
    assert gu.rel_err(full, want) < 2e-2
    assert gu.rel_err(part, want) < 2e-2
    assert gu.rel_err(part, full) < 2e-2


def test_full_size_properties_of_the_relative_configuration():
    """BASELINE configs[4] at full size (RelativeTransformer d=512 L=6 H=8 ff=2048, max_sequence 4096, bf16, batch 2 x
    seq 4096), where the oracle's [B,H,S,S] tensors are too slow: size-independent properties of the same kernels.
      * determinism: two forward+backward passes give bit-identical logits and gradients;
      * causality: changing tokens at positions >= p leaves the logits of positions < p bit-identical (p on a 64-query
        block boundary);
      * at S = 2048 < max_sequence the rows of dE the skew never reaches (< max_seq - S) are exactly zero."""
    from musicgeneration_amd.old_models.relative_transformer import RelativeTransformer
    torch.manual_seed(21)
    m = RelativeTransformer(618, 512, 6, 8, 4096, 2048, dropout=0.0, dtype="bf16").cuda().train()
    g = torch.Generator().manual_seed(22)
    x = torch.randint(2, 618, (2, 4096), generator=g).cuda()
    y = torch.roll(x, -1, dims=1)

    def run(xx, yy):
        m.zero_grad(set_to_none=True)
        logits = m(xx)
        loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, 618), yy.reshape(-1))
        loss.backward()
        return logits.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}, float(loss.detach())

    l1, g1, loss1 = run(x, y)
    l2, g2, loss2 = run(x, y)
    assert np.isfinite(loss1) and loss1 == loss2
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    p = 3072
    x2 = x.clone()
    x2[:, p:] = torch.randint(2, 618, (2, 4096 - p), generator=g).cuda()
    l3, _, _ = run(x2, y)
    assert torch.equal(l3[:, :p], l1[:, :p])
    assert not torch.equal(l3[:, p:], l1[:, p:])
    _, g4, _ = run(x[:, :2048].contiguous(), y[:, :2048].contiguous())
    for n, gr in g4.items():
        if n.endswith(".E"):
            assert float(gr[:, :2048].abs().max()) == 0.0, n
            assert float(gr[:, 2048:].abs().max()) > 0.0, n


def test_state_dict_keys_match_reference():
    z = gu.load("unified_tiny")
    ref_keys = {k[3:] for k in z.files if k.startswith("sd.")}
    m = _unified_model(z, gu.state_dict(z))
    assert set(m.state_dict().keys()) == ref_keys
    # the reference's persistent mask buffer: accepted on load, re-created on request
    sd = dict(m.state_dict())
    sd["mask"] = torch.zeros(8, 64, 64)
    m.load_state_dict(sd)
    ref_sd = m.reference_state_dict()
    assert torch.equal(ref_sd["mask"], O.alibi_mask(64, 8))


def test_separated_fp32():
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer, Criterion
    z = gu.load("separated_tiny")
    m = AlibiTransformer(92, 130, 203, 16, 16, 32, 2, 8, 64, 128, dropout=0.0)
    m.load_state_dict(gu.state_dict(z))
    m = m.cuda()
    crit = Criterion(92, 130, 203, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2}).cuda()
    assert sorted(crit.state_dict().keys()) == list(z["crit_state_keys"])
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    preds = m(x)
    assert isinstance(preds, tuple) and len(preds) == 3
    for p, k in zip(preds, ("logit_n", "logit_v", "logit_d")):
        assert gu.rel_err(p, z[k]) < 1e-5
    loss, ld = crit(preds, y)
    for k in ("loss", "loss_notes", "loss_velocity", "loss_duration"):
        assert abs(float(ld[k].detach()) - float(z[k])) < 1e-5 * abs(float(z[k])), k
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in gu.grads(z).items():
        assert gu.rel_err(named[k].grad, g) < 2e-5, k
    # plain tuple of tensors (not the packed fast path) gives the same loss
    loss2, _ = crit(tuple(p.detach().clone() for p in preds), y)
    assert abs(float(loss2) - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))


def test_separated_fused_adamw_matches_torch_adamw():
    """Two optimizer steps of the separated model: FusedAdamW (multi-tensor kernel; the three head gradients are
    odd-offset slices of one fused [d, 92+130+203] gradient) against torch.optim.AdamW on the same gradients."""
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer, Criterion
    from musicgeneration_amd.optim import FusedAdamW
    z = gu.load("separated_tiny")
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    finals = []
    for opt_cls in (torch.optim.AdamW, FusedAdamW):
        m = AlibiTransformer(92, 130, 203, 16, 16, 32, 2, 8, 64, 128, dropout=0.0)
        m.load_state_dict(gu.state_dict(z))
        m = m.cuda()
        crit = Criterion(92, 130, 203, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2}).cuda()
        opt = opt_cls(m.parameters(), lr=1e-3, weight_decay=1e-4)
        for _ in range(2):
            loss, _ = crit(m(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
        finals.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    for k, v in finals[0].items():
        assert float((finals[1][k] - v).abs().max()) < 2e-6, k


@pytest.mark.parametrize("S", [48, 64])
def test_relative_fp32(S):
    from musicgeneration_amd.old_models.relative_transformer import RelativeTransformer
    z = gu.load("relative_tiny")
    m = RelativeTransformer(618, 64, 2, 8, 64, 128, dropout=0.0)
    m.load_state_dict(gu.state_dict(z))
    m = m.cuda()
    x, y = torch.from_numpy(z[f"s{S}.x"]).cuda(), torch.from_numpy(z[f"s{S}.y"]).cuda()
    logits = m(x)
    assert gu.rel_err(logits, z[f"s{S}.logits"]) < 1e-5
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 618), y.reshape(-1))
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in gu.grads(z, f"s{S}.grad.").items():
        assert gu.rel_err(named[k].grad, g) < 2e-5, k


@pytest.mark.parametrize("dtype,tol", [(None, 1e-5), ("bf16", 2e-2)])
def test_relative_seeded(dtype, tol):
    from musicgeneration_amd.old_models.relative_transformer import RelativeTransformer
    z = gu.load("relative_dh64_s320")
    V, d, L, H, M, ff = (int(v) for v in z["cfg"])
    m = RelativeTransformer(V, d, L, H, M, ff, dropout=0.0, dtype=dtype)
    m.load_state_dict(gu.seeded_state_dict(z))
    m = m.cuda()
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    logits = m(x)
    assert gu.rel_err(logits.float(), z["logits"]) < tol
    if dtype is None:
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, V), y.reshape(-1))
        loss.backward()
        named = dict(m.named_parameters())
        for k, g in gu.grads(z).items():
            assert gu.rel_err(named[k].grad, g) < 2e-5, k


def test_train_step_matches_reference():
    """Three AdamW steps (train_utils.py:160-173 body) reproduce the reference's losses and weights;
    both torch.optim.AdamW and the fused mgx AdamW."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion
    from musicgeneration_amd.optim import FusedAdamW
    z = gu.load("unified_train3")
    for opt_cls in (torch.optim.AdamW, FusedAdamW):
        m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=8, max_sequence=64, proj_forward=128, dropout=0.0)
        m.load_state_dict(gu.state_dict(z, "sd0."))
        m = m.cuda()
        crit = Criterion(1)
        opt = opt_cls(m.parameters(), lr=1e-3, weight_decay=1e-4)
        for step in range(3):
            x = torch.from_numpy(z["x"][step]).cuda()
            y = torch.roll(x, -1, dims=1)
            y[:, -2:] = 1
            m.train()
            loss, _ = crit(m(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            assert abs(float(loss.detach()) - float(z["losses"][step])) < 1e-5 * float(z["losses"][step])
        for k, v in gu.state_dict(z, "sd3.").items():
            assert float((m.state_dict()[k].cpu() - v).abs().max()) < 2e-5, (opt_cls.__name__, k)


def test_greedy_decode_bit_exact_unified():
    """KV-cached decode (fp32) reproduces the reference loop's token sequences exactly,
    including the i+1 quirk and the primer overwrite (inference_unified.py:12-21)."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    from musicgeneration_amd.model_baseline.inference_unified import inference
    z = gu.load("greedy_unified")
    m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=8, max_sequence=64, proj_forward=128, dropout=0.0)
    m.load_state_dict(gu.state_dict(z))
    m = m.cuda().eval()
    for idx in (1, 9):
        seq = torch.from_numpy(z[f"idx{idx}.seq_in"].copy()).cuda()
        out = inference(m, seq, idx=idx, deterministic=True)
        assert out.data_ptr() == seq.data_ptr()
        assert np.array_equal(out.cpu().numpy(), z[f"idx{idx}.seq_out"]), idx
    for idx in (0, 9):
        seq = torch.from_numpy(z[f"last{idx}.seq_in"].copy()).cuda()
        out = m.inference(seq, idx=idx, deterministic=False)
        assert np.array_equal(out.cpu().numpy(), z[f"last{idx}.seq_out"]), idx


def test_greedy_decode_bit_exact_separated():
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer
    from musicgeneration_amd.model_baseline.inference_separated import inference
    z = gu.load("greedy_separated")
    m = AlibiTransformer(92, 130, 203, 16, 16, 32, 2, 8, 64, 128, dropout=0.0)
    m.load_state_dict(gu.state_dict(z))
    m = m.cuda().eval()
    for idx in (1, 9):
        seq = torch.from_numpy(z[f"idx{idx}.seq_in"].copy()).cuda()
        out = inference(m, seq, idx=idx, deterministic=True)
        assert np.array_equal(out.cpu().numpy(), z[f"idx{idx}.seq_out"]), idx
    seq = torch.from_numpy(z["last0.seq_in"].copy()).cuda()
    assert np.array_equal(m.inference(seq, idx=0, deterministic=False).cpu().numpy(), z["last0.seq_out"])


def test_decode_matches_full_forward_long():
    """Size-independent property at a kernel-shaped config: cached decode logits == full-forward
    logits at every step of a 300-token greedy continuation (d=512, dh=64, fp32), and the
    generated sequence equals the oracle's full-recompute loop on the CPU for the first steps."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    from musicgeneration_amd.decode import KVCacheDecoder, sample_next_plus_one
    torch.manual_seed(3)
    m = AlibiTransformer(vocab=618, embed_dim=512, n_layers=2, n_heads=8, max_sequence=512, proj_forward=1024,
                         dropout=0.0).cuda().eval()
    B, L = 2, 96
    seq = torch.zeros(B, L, dtype=torch.int32).cuda()
    seq[:, 1:17] = torch.randint(2, 618, (B, 16), generator=torch.Generator().manual_seed(4)).cuda().int()
    ref_seq = seq.clone()
    out = sample_next_plus_one(m, seq, idx=17, deterministic=True)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = O.sample_next_plus_one(lambda s: O.unified_forward(sd, s, 8), ref_seq.cpu(), idx=17, deterministic=True)
    assert torch.equal(out.cpu(), ref)
    # provisional-row logits equal the full forward's rows
    dec = KVCacheDecoder(m, B, L)
    dec.rows(out[:, :40], 0)
    part = dec.rows(out[:, 40:43], 40)
    with torch.no_grad():
        full = m(out[:, :43])
    assert gu.rel_err(part, full[:, 40:43]) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graphed_decode_equals_eager(dtype):
    """The HIP-graph replay of the sampling step (device-side position) produces exactly the
    tokens of the step-by-step loop, greedy, at a kernel-shaped config."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    from musicgeneration_amd.decode import sample_next_plus_one
    torch.manual_seed(5)
    m = AlibiTransformer(vocab=618, embed_dim=512, n_layers=2, n_heads=4, max_sequence=512, proj_forward=1024,
                         dropout=0.0, dtype=dtype).cuda().eval()
    B, L = 3, 80
    seq = torch.zeros(B, L, dtype=torch.int64).cuda()
    seq[:, 1:9] = torch.randint(2, 618, (B, 8), generator=torch.Generator().manual_seed(6)).cuda()
    a = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=True, use_graph=True)
    b = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=True, use_graph=False)
    assert torch.equal(a, b)
    assert int((a[:, 9:L - 1] != 0).sum()) > 0
    # stochastic path: runs, stays in the vocabulary, leaves the last position untouched
    c = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=False, use_graph=True)
    assert int(c.max()) < 618 and torch.equal(c[:, -1], seq[:, -1]) and torch.equal(c[:, :8], seq[:, :8])


@pytest.mark.parametrize("deterministic", [True, False])
def test_graphed_decode_separated_bf16(deterministic):
    """The separated model (inference_separated.py:18-28) on the fused sampling head/tail, bf16, at a kernel-shaped
    width (d = 256 + 256 + 512): the HIP-graph replay writes exactly the tokens of the same kernels launched one by
    one - greedy, and stochastic with the same generator state (counter-based draw) - tokens stay inside each
    stream's vocabulary, primer and last column are untouched; greedy also equals the unfused step (rows() + argmax)
    wherever the two top logits of the fused step are not within bf16 rounding of each other."""
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer
    from musicgeneration_amd.decode import KVCacheDecoder, sample_next_plus_one
    torch.manual_seed(5)
    m = AlibiTransformer(92, 130, 203, 256, 256, 512, 2, 8, 128, 1024, dropout=0.0, dtype="bf16").cuda().eval()
    B, L = 4, 72
    g = torch.Generator().manual_seed(6)
    seq = torch.zeros(B, 3, L, dtype=torch.int32).cuda()
    seq[:, :, 1:9] = torch.stack([torch.randint(2, v - 1, (B, 8), generator=g) for v in (92, 130, 203)], 1).cuda().int()
    assert KVCacheDecoder(m, B, L).fused_step_supported(seq)
    torch.manual_seed(11)
    a = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=deterministic, use_graph=True)
    torch.manual_seed(11)
    b = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=deterministic, use_graph=False)
    assert torch.equal(a, b)
    for k, v in enumerate((92, 130, 203)):
        assert int(a[:, k].max()) < v and int(a[:, k].min()) >= 0
    assert torch.equal(a[:, :, :9], seq[:, :, :9]) and torch.equal(a[:, :, -1], seq[:, :, -1])
    assert int((a[:, :, 9:L - 1] != 0).sum()) > 0
    if not deterministic:
        torch.manual_seed(12)
        c = sample_next_plus_one(m, seq.clone(), idx=9, deterministic=False, use_graph=True)
        assert not torch.equal(a, c)                            # another seed, another continuation
        return
    # the unfused step on the same prefix: its argmax agrees unless the decision is inside bf16 rounding
    dec = KVCacheDecoder(m, B, L)
    i = 30
    dec.rows(a[:, :, :i - 1], 0)
    heads = dec.rows(torch.cat([a[:, :, i - 1:i], seq[:, :, i:i + 2]], 2), i - 1)    # rows i, i+1 still stale (zeros)
    for k, hk in enumerate(heads):
        top2 = hk[:, 2].float().topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 0.05 * top2[:, 0].abs().clamp_min(1.0)
        assert torch.equal(hk[:, 2].float().argmax(-1)[clear], a[:, k, i].long()[clear]), k


def test_training_loop_end_to_end(tmp_path):
    """training_loop with the reference's signature: loss decreases on a repeating batch,
    checkpoints carry the reference's keys ('optmizer', 'grad_scaler', model 'mask') and resume."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion, Metrics
    from musicgeneration_amd import train_utils as tu
    from musicgeneration_amd.synthetic import SyntheticUnifiedDataset
    torch.manual_seed(0)
    m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=4, max_sequence=64, proj_forward=128, dropout=0.1).cuda()
    crit = Criterion(1).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=3e-3, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, 50)

    class Repeat:
        def __init__(self):
            self.b = SyntheticUnifiedDataset(4, 64, seed=5).get()
        def __len__(self):
            return 30
        def get(self):
            return tuple(t.cpu().numpy() for t in self.b)      # numpy, as the reference's Ray dataset yields

    logger = tu.Logger(None, Metrics())
    n = tu.training_loop(m, Repeat(), crit, opt, sched, logger, log_scalar_metric_step=40, checkpoint_step=60,
                         epochs=1, batch_size=4, run_dir=str(tmp_path), mixed_precision=False, sync_every=4)
    assert n == 120
    losses = [h[1]["train/loss"] for h in logger.history]
    assert len(losses) >= 2 and losses[-1] < losses[0]
    assert "train/accuracy" in logger.history[0][1]
    ck = tu.get_checkpoints(str(tmp_path))
    assert ck and len(ck) <= 3
    state = torch.load(ck[-1], map_location="cpu", weights_only=True)
    assert set(state.keys()) == {"model", "optmizer", "grad_scaler", "criterion", "lr_scheduler", "steps"}
    assert "mask" in state["model"] and state["model"]["mask"].shape == (4, 64, 64)
    m2 = AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=4, max_sequence=64, proj_forward=128)
    assert tu.load_checkpoint(str(tmp_path), m2) == state["steps"]
    assert torch.equal(m2.state_dict()["embed.weight"], state["model"]["embed.weight"])


def test_nan_loss_never_reaches_a_checkpoint(tmp_path):
    """A NaN loss is detected at the deferred sync point (train_utils.py:168 asserts every step); the loop flushes
    before every checkpoint, so no <step>.tar is written once a NaN step has happened."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion, Metrics
    from musicgeneration_amd import train_utils as tu
    from musicgeneration_amd.synthetic import SyntheticUnifiedDataset
    torch.manual_seed(0)
    m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=1, n_heads=4, max_sequence=64, proj_forward=128, dropout=0.0).cuda()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 50)

    class Poisoned(Criterion):
        calls = 0
        def forward(self, xs, y):
            loss, d = super().forward(xs, y)
            Poisoned.calls += 1
            if Poisoned.calls == 3:
                loss = loss * float("nan")
                d = {"loss": loss}
            return loss, d

    class Data:
        def __init__(self):
            self.b = SyntheticUnifiedDataset(2, 32, seed=5).get()
        def __len__(self):
            return 40
        def get(self):
            return self.b

    with pytest.raises(AssertionError, match="NaN"):
        tu.training_loop(m, Data(), Poisoned(1).cuda(), opt, sched, tu.Logger(None, Metrics()), log_scalar_metric_step=10 ** 6,
                         checkpoint_step=10, epochs=1, batch_size=2, run_dir=str(tmp_path), sync_every=16)
    # checkpoint_step 10 samples = every 5 iterations: the first save would come at iteration 5, after the NaN of 3
    assert tu.get_checkpoints(str(tmp_path)) == []


def test_dropout_training_mode_runs_and_differs():
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer
    torch.manual_seed(1)
    m = AlibiTransformer(vocab=618, embed_dim=64, n_layers=1, n_heads=4, max_sequence=64, proj_forward=128, dropout=0.3).cuda()
    x = torch.randint(0, 618, (2, 32)).cuda()
    m.train()
    a, b = m(x), m(x)
    assert not torch.equal(a, b)
    m.eval()
    assert torch.equal(m(x), m(x))
    m.train()
    m(x).float().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_fused_adamw_updates_reach_the_bf16_forward():
    """The fused optimizer writes the fp32 masters from a kernel; the bf16 weight shadows of the blocks must
    follow (regression: they were keyed on a version counter the kernel did not advance, so bf16 training ran
    on stale block weights).  Same model, same batches: FusedAdamW vs torch.optim.AdamW losses stay together."""
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion
    from musicgeneration_amd.optim import FusedAdamW
    losses = {}
    for opt_cls in (torch.optim.AdamW, FusedAdamW):
        torch.manual_seed(3)
        m = AlibiTransformer(vocab=618, embed_dim=128, n_layers=2, n_heads=2, max_sequence=64, proj_forward=256,
                             dropout=0.0, dtype="bf16").cuda()
        opt = opt_cls(m.parameters(), lr=1e-2, weight_decay=1e-4)
        x = torch.randint(0, 618, (4, 64), generator=torch.Generator().manual_seed(5)).cuda()
        y = torch.roll(x, -1, dims=1)
        out = []
        for step in range(6):
            loss, _ = Criterion(1)(m(x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        W = m.decoder_layers[0].weights(torch.bfloat16)
        master = m.decoder_layers[0].linear1.weight.detach()
        assert torch.equal(W.w1, master.bfloat16())
        assert torch.equal(W.w1_t, master.t().bfloat16())
        losses[opt_cls.__name__] = out
    a, b = losses["AdamW"], losses["FusedAdamW"]
    assert a[-1] < a[0] - 1.0                                  # it trains
    assert max(abs(u - v) for u, v in zip(a, b)) < 0.05 * a[0], (a, b)


@pytest.mark.parametrize("name,dtype,tol", [("alibi_tied_tiny", None, 1e-5), ("rotary_tiny", None, 1e-5),
                                            ("rotary_dh64_s320", None, 1e-5), ("rotary_dh64_s320", "bf16", 2e-2)])
def test_old_model_variants(name, dtype, tol):
    """Tied-embedding ALiBi (old_models/alibi.py) and the rotary variant (old_models/rotary_transformer.py):
    reference state_dict in, logits and (fp32) every gradient against the reference's."""
    z = gu.load(name)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    if name.startswith("alibi"):
        from musicgeneration_amd.old_models.alibi import AlibiTransformer
        m = AlibiTransformer(618, 64, 64, 2, 8, 64, 128, dropout=0.0, dtype=dtype)
    else:
        from musicgeneration_amd.old_models.rotary_transformer import RotaryTransformer
        cfg = (618, 64, 2, 8, 64, 128) if "tiny" in name else (618, 256, 1, 4, 320, 512)
        m = RotaryTransformer(*cfg, dropout=0.0, dtype=dtype)
    m.load_state_dict(sd)
    m = m.cuda()
    x = torch.from_numpy(z["x"]).cuda()
    logits = m(x)
    assert gu.rel_err(logits.float(), z["logits"]) < tol
    if dtype is None:
        loss = torch.nn.functional.cross_entropy(logits.transpose(-1, -2), torch.roll(x, -1, 1))
        assert abs(float(loss.detach()) - float(z["loss"])) < 1e-5 * float(z["loss"])
        loss.backward()
        named = dict(m.named_parameters())
        for k in z.files:
            if k.startswith("grad."):
                assert gu.rel_err(named[k[5:]].grad, z[k]) < 2e-5, k


def test_rotary_intended_rotation_vs_oracle():
    """The non-truncated rotation (intended_rotation=True) is not what the reference computes; it is checked
    against the oracle's as_reference=False path (parity unpinned by the reference for this switch): logits and
    every gradient, so the in-place rotation of q | k and the transposed rotation of dq | dk are both exercised
    with non-trivial angles."""
    from oracle import reference_port as O
    from musicgeneration_amd.old_models.rotary_transformer import RotaryTransformer
    z = gu.load("rotary_dh64_s320")
    sd = {k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith("sd.")}
    sd["embedding"] *= 20.0                                       # N(0, 0.02) init leaves the scores ~0 and the
    sd["decoder_layers.0.attn.c_attn.w"] *= 8.0                   # softmax uniform whatever the rotation: make q.k O(1)
    m = RotaryTransformer(618, 256, 1, 4, 320, 512, dropout=0.0, intended_rotation=True)
    m.load_state_dict(sd)
    m = m.cuda()
    x = torch.from_numpy(z["x"])
    osd = O.make_trainable({k: v.clone() for k, v in sd.items() if k != "pos_embed.emb"})
    want = O.rotary_forward(osd, x, 4, as_reference=False)
    logits = m(x.cuda())
    assert gu.rel_err(logits, want) < 1e-5
    assert gu.rel_err(want, O.rotary_forward(osd, x, 4, as_reference=True)) > 1e-3    # and differs from the truncated table
    y = torch.roll(x, -1, 1)
    torch.nn.functional.cross_entropy(want.transpose(-1, -2), y).backward()
    torch.nn.functional.cross_entropy(logits.transpose(-1, -2), y.cuda()).backward()
    for k, p in m.named_parameters():
        if float(osd[k].grad.abs().max()) < 1e-7:                 # norm.b: exactly zero in exact arithmetic (tied head)
            assert float(p.grad.abs().max()) < 1e-7, k
            continue
        assert gu.rel_err(p.grad, osd[k].grad) < 2e-5, k


def test_bench_data_parallel_path_on_rccl():
    """bench.py's data-parallel step (RCCL process group, GradientBucketer: side-stream all-reduce from the grad
    hooks, join before the optimizer) with one rank on this GPU, against the plain single-GPU step: same seeds,
    same batches -> the same loss after 3 steps (averaging over a world of one is the identity)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--seq", "512",
           "--no-cpu-baseline", "--no-decode"]
    lines = []
    for extra in ([], ["--force-dp"], ["--force-dp", "--dp-mode", "inplace"]):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
        r = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=110, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert abs(lines[0]["final_loss"] - lines[2]["final_loss"]) < 2e-2
    assert lines[0]["n_gpus"] == lines[1]["n_gpus"] == 1
    assert np.isfinite(lines[1]["final_loss"])
    assert abs(lines[0]["final_loss"] - lines[1]["final_loss"]) < 2e-2      # dropout masks are seeded; bf16 step


def test_data_parallel_two_ranks_on_one_gpu_equal_the_full_batch():
    """The data-parallel training path with TWO ranks, both on this GPU (tests/dp2_worker.py): bf16 unified model,
    fused blocks, gradient-bucket hooks on a side stream, weight gradients produced in their bucket slots, the global
    CE normaliser, FusedAdamW, GradientBucketer.close().  Rank r trains on rows r::2; after every step the averaged
    gradients, the averaged loss and the updated weights equal those of one process on the whole batch (different
    summation splits: 1e-2 on gradients of a bf16 step, 1e-3 on weights).
    The bytes travel through a test-only host-staged wrapper around gloo ("gloo-host"): handing GPU tensors to gloo
    hangs in about one run of seven with two processes on one GPU, inside ProcessGroupGloo's staging copies - the
    stacks are in profiles/r03_dp2_hang_evidence.txt; RCCL does not use that path.  The parent owns the two PIDs."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29541", DP2_TRANSPORT="gloo-host", DP2_WATCHDOG="60")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(root, "tests", "dp2_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    t0 = time.time()
    while time.time() - t0 < 90 and any(p.poll() is None for p in procs):
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            p.kill()
    outs = [p.communicate()[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: rc {p.returncode}\n{out[-3000:]}"
        assert f"rank {r} ok" in out


def test_data_parallel_weight_gradients_land_in_their_bucket_slots():
    """GradientBucketer (flat mode) tags every parameter with its slot of the bucket buffer; the fused decoder block
    writes its four weight gradients straight into those slots (ops.grad_slot), so the bucket packs only the small
    tensors.  One rank on this GPU: the block weights' gradients ARE their slots before the exchange, and the gradients
    after the exchange equal those of the same model without the bucketer."""
    import os
    import torch.distributed as dist
    from musicgeneration_amd.data_parallel import GradientBucketer
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def build():
            torch.manual_seed(7)
            return AlibiTransformer(vocab=96, embed_dim=128, n_layers=2, n_heads=2, max_sequence=256, proj_forward=256,
                                    dropout=0.0, dtype="bf16").to(dev)
        g = torch.Generator().manual_seed(3)
        x = torch.randint(2, 96, (2, 192), generator=g).to(dev)
        y = torch.randint(2, 96, (2, 192), generator=g).to(dev)
        crit = Criterion().to(dev)
        ref = build()
        crit(ref(x), y)[0].backward()
        m = build()
        ddp = GradientBucketer(m, mode="flat")
        seen = {}
        for name, p in m.named_parameters():
            if p.dim() >= 2 and "decoder" in name and ("weight" in name or name.endswith(".w")):
                p.register_post_accumulate_grad_hook(
                    lambda q, name=name: seen.__setitem__(name, q.grad.data_ptr() == q.mgx_grad_slot.data_ptr()))
        crit(m(x), y)[0].backward()
        ddp.finish()
        torch.cuda.synchronize()
        assert len(seen) >= 8 and all(seen.values()), seen
        for (n1, p1), (_, p2) in zip(m.named_parameters(), ref.named_parameters()):
            assert p1.grad.data_ptr() == p1.mgx_grad_slot.data_ptr(), n1
            assert torch.equal(p1.grad, p2.grad), n1
        # a second backward without zero_grad accumulates: the slot must not be handed out again
        crit(m(x), y)[0].backward()
        ddp.finish()
        for (n1, p1), (_, p2) in zip(m.named_parameters(), ref.named_parameters()):
            assert gu.rel_err(p1.grad, 2 * p2.grad) < 1e-6, n1
    finally:
        if started:
            dist.destroy_process_group()


def test_full_size_properties_of_the_bench_configuration():
    """BASELINE configs[1] at full size (alibi_unified d=1024 L=6 H=8 ff=4096, bf16, batch 16 x seq 2048), where the
    oracle is too slow: size-independent properties of the same kernels.
      * determinism: two forward+backward passes give bit-identical logits and gradients (no atomics anywhere);
      * causality: changing tokens at positions >= p leaves the logits of positions < p bit-identical (p on a
        32-query wave boundary; inside the straddled wave, equal to bf16 tolerance);
      * linearity of the backward in the loss scale: a x4 loss gives exactly x4 gradients (power-of-two scale,
        fp32 gradient accumulation);
      * padding targets (ignore_index) carry no gradient into the logits."""
    from musicgeneration_amd.model_baseline import build_everything_unified as cfg
    args = cfg.m_args.get_copy()
    args.max_sequence = 2048
    args.dtype = "bf16"
    args.dropout = 0.0
    torch.manual_seed(11)
    model, crit = cfg.build_model_and_criterion(args)
    model = model.cuda().train()
    g = torch.Generator().manual_seed(12)
    x = torch.randint(2, 618, (16, 2048), generator=g).cuda()
    y = torch.roll(x, -1, dims=1)
    y[:, -100:] = 1                                            # pad_index tail

    def run(xx, scale=1.0):
        model.zero_grad(set_to_none=True)
        logits = model(xx)
        loss, _ = crit(logits, y)
        (loss * scale).backward()
        return logits.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, float(loss.detach())

    l1, g1, loss1 = run(x)
    l2, g2, loss2 = run(x)
    assert np.isfinite(loss1) and loss1 == loss2
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    # p on a wave's 32-query boundary: the flash kernels' deferred rescale is decided per wave (32 queries), so
    # queries that share a wave with a changed position may differ in rounding (not in the keys they see)
    p = 1536
    x2 = x.clone()
    x2[:, p:] = torch.randint(2, 618, (16, 2048 - p), generator=g).cuda()
    l3, _, _ = run(x2)
    assert torch.equal(l3[:, :p], l1[:, :p])
    assert not torch.equal(l3[:, p:], l1[:, p:])
    x3 = x.clone()
    x3[:, 1500:] = x2[:, 1500:]
    l5, _, _ = run(x3)
    assert torch.equal(l5[:, :1472], l1[:, :1472])             # whole waves before the change: bit-identical
    assert gu.rel_err(l5[:, 1472:1500].float(), l1[:, 1472:1500].float()) < 2e-2
    _, g4, _ = run(x, scale=4.0)
    for n in g1:
        assert torch.equal(g4[n], 4.0 * g1[n]), n
    # ignore_index rows: d loss / d logits is zero there
    logits = model(x).detach().requires_grad_(True)
    loss, _ = crit(logits, y)
    loss.backward()
    assert float(logits.grad[:, -100:].abs().max()) == 0.0
    assert float(logits.grad[:, :-100].abs().max()) > 0.0


def test_shadow_cache_follows_replaced_parameters():
    """The bf16 weight shadows are keyed by id(parameter): a parameter that is replaced (and freed) must not leave
    its shadow behind for whatever tensor receives the same id next."""
    from musicgeneration_amd.base_layers import ShadowCache
    c = ShadowCache()
    for trial in range(20):
        p = torch.nn.Parameter(torch.full((64, 64), float(trial)).cuda())
        s = c.get(p, torch.bfloat16)
        a = c.pair(p, torch.bfloat16, (64, 64), False)
        t = c.pair(p, torch.bfloat16, (64, 64), True)
        assert float(s[0, 0]) == float(trial) and float(a[3, 5]) == float(trial) and float(t[5, 3]) == float(trial)
        del p, s, a, t                                     # the next Parameter often reuses this id


def test_full_size_properties_of_the_separated_configuration():
    """BASELINE configs[2] at its real depth and shape (alibi_separated 256/256/512 -> d=1024, L=10, H=8, ff=4096, bf16,
    batch 8 x seq 1024 per GPU: M = 8192 token rows, so the 1024-column projections take the 128-row tiles of
    gemm_nt_kernel<..., HALF>), where the oracle is too slow: size-independent properties of the same kernels.
      * the host picks the half-tile path for these shapes (the C ABI reports the partial-sum rows it will write);
      * determinism: two forward+backward passes give bit-identical logits and gradients;
      * causality: changing tokens at positions >= p leaves the logits of positions < p bit-identical;
      * targets with class weight 0 (token 1 of the note / velocity rows) and the ignore_index of the duration row
        carry no gradient into the logits; the other targets do."""
    from musicgeneration_amd import _lib
    from musicgeneration_amd.model_baseline import build_everything_separated as cfg
    from musicgeneration_amd.synthetic import SyntheticSeparatedDataset
    B, S = 8, 1024
    assert _lib.lib.mgx_gemm_nt_dfac_colpart_rows(B * S, 1024) == 2 * (B * S // 128)      # 128-row tiles (HALF)
    assert _lib.lib.mgx_gemm_nt_dfac_colpart_rows(B * S, 4096) == 2 * (B * S // 256)      # 256-row tiles
    args = cfg.m_args.get_copy()
    args.dtype = "bf16"
    args.dropout = 0.0
    torch.manual_seed(21)
    model, crit = cfg.build_model_and_criterion(args)
    model, crit = model.cuda().train(), crit.cuda()
    x, y = SyntheticSeparatedDataset(B, S, seed=22, device="cuda").get()
    y = y.clone()
    y[:, :, -64:] = 1                                           # a tail of end tokens: weight 0 / ignore_index

    def run(xx):
        model.zero_grad(set_to_none=True)
        pred = model(xx)
        loss, parts = crit(pred, y)
        loss.backward()
        return [p.detach().clone() for p in pred], {n: p.grad.detach().clone() for n, p in model.named_parameters()}, \
            float(loss.detach())

    l1, g1, loss1 = run(x)
    l2, g2, loss2 = run(x)
    assert np.isfinite(loss1) and loss1 == loss2
    assert [tuple(t.shape) for t in l1] == [(B, S, 92), (B, S, 130), (B, S, 203)]
    for a, b in zip(l1, l2):
        assert torch.equal(a, b)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
        assert torch.isfinite(g1[n]).all(), n
    p = 768                                                     # a wave's 32-query boundary
    x2 = x.clone()
    x2[:, :, p:] = SyntheticSeparatedDataset(B, S, seed=23, device="cuda").get()[0][:, :, p:]
    l3, _, _ = run(x2)
    for a, b in zip(l1, l3):
        assert torch.equal(a[:, :p], b[:, :p])
    assert any(not torch.equal(a[:, p:], b[:, p:]) for a, b in zip(l1, l3))
    # no gradient from weight-0 / ignored targets
    pred = [t.detach().requires_grad_(True) for t in model(x)]
    loss, _ = crit(tuple(pred), y)
    loss.backward()
    for t in pred:
        assert float(t.grad[:, -64:].abs().max()) == 0.0
        assert float(t.grad[:, :-64].abs().max()) > 0.0


def test_training_loop_mixed_precision_with_a_bucketer(tmp_path):
    """train_utils.training_loop as build_everything_separated.py:97 runs it (mixed_precision=True: bf16 autocast
    policy, disabled GradScaler kept for the checkpoint key) with ddp=GradientBucketer (one RCCL rank on this GPU) and
    FusedAdamW: the loss falls on a repeating batch, the checkpoint keeps the reference's keys, close() detaches."""
    import os
    import torch.distributed as dist
    from musicgeneration_amd import ops, train_utils as tu
    from musicgeneration_amd.data_parallel import GradientBucketer
    from musicgeneration_amd.model_baseline.alibi_separated import AlibiTransformer, Criterion, Metrics
    from musicgeneration_amd.optim import FusedAdamW
    from musicgeneration_amd.synthetic import SyntheticSeparatedDataset
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29535")
    dev = torch.device("cuda", 0)
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        m = AlibiTransformer(92, 130, 203, 32, 32, 64, 2, 2, 128, 256, dropout=0.1).to(dev)     # d=128, dh=64
        crit = Criterion(92, 130, 203).to(dev)
        opt = FusedAdamW(m.parameters(), lr=3e-3, weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.StepLR(opt, 50)

        class Repeat:
            def __init__(self):
                self.b = SyntheticSeparatedDataset(4, 128, seed=5, device="cuda").get()
            def __len__(self):
                return 30
            def get(self):
                return tuple(t.cpu().numpy() for t in self.b)

        logger = tu.Logger(None, Metrics())
        with GradientBucketer(m, bucket_mb=0.05) as ddp:
            assert len(ddp.buckets) > 2
            n = tu.training_loop(m, Repeat(), crit, opt, sched, logger, log_scalar_metric_step=40, checkpoint_step=60,
                                 epochs=1, batch_size=4, run_dir=str(tmp_path), mixed_precision=True, ddp=ddp, sync_every=4)
            # the bf16 policy ran (the block weights were shadowed in bf16) and the gradients sit in their bucket slots
            assert all(p.grad.data_ptr() == p.mgx_grad_slot.data_ptr() for p in m.parameters())
        assert n == 120
        assert ops.LOSS_DENOMINATOR_HOOK is None and not hasattr(next(m.parameters()), "mgx_grad_slot")
        losses = [h[1]["train/loss"] for h in logger.history]
        assert len(losses) >= 2 and losses[-1] < losses[0], losses
        state = torch.load(tu.get_checkpoints(str(tmp_path))[-1], map_location="cpu", weights_only=True)
        assert set(state.keys()) == {"model", "optmizer", "grad_scaler", "criterion", "lr_scheduler", "steps"}
        assert "mask" in state["model"] and {"weight_n", "weight_v"} <= set(state["criterion"].keys())
    finally:
        if started:
            dist.destroy_process_group()


def test_shared_block_gets_its_bucket_slot_once_per_backward():
    """A decoder block applied twice in one graph (shared weights): both backward nodes produce gradients for the
    same parameters.  ops.grad_slot hands the parameter's bucket slot to the first producer only; autograd adds the
    second gradient, so p.grad is the sum of both (not twice the last one)."""
    import os
    import torch.distributed as dist
    from musicgeneration_amd.data_parallel import GradientBucketer
    from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    dev = torch.device("cuda", 0)
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def build():
            torch.manual_seed(7)
            m = AlibiTransformer(vocab=96, embed_dim=128, n_layers=2, n_heads=2, max_sequence=256, proj_forward=256,
                                 dropout=0.0, dtype="bf16").to(dev)
            m.decoder_layers[1] = m.decoder_layers[0]                # the same block twice
            return m
        g = torch.Generator().manual_seed(3)
        x = torch.randint(2, 96, (2, 192), generator=g).to(dev)
        y = torch.randint(2, 96, (2, 192), generator=g).to(dev)
        crit = Criterion().to(dev)
        ref = build()
        crit(ref(x), y)[0].backward()
        m = build()
        with GradientBucketer(m, mode="flat") as ddp:
            crit(m(x), y)[0].backward()
            ddp.finish()
            torch.cuda.synchronize()
            for (n1, p1), (_, p2) in zip(m.named_parameters(), ref.named_parameters()):
                assert gu.rel_err(p1.grad, p2.grad) < 1e-6, n1
    finally:
        if started:
            dist.destroy_process_group()
