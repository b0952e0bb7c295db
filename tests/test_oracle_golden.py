"""CPU: the oracle (oracle/reference_port.py) against the golden vectors the
real reference produced (tests/golden/make_golden.py).  Tolerance 1e-5
relative to the tensor's max magnitude (north_star's fp32 bar)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import reference_port as O

TOL = 1e-5


def test_slopes_and_masks():
    z = gu.load("alibi_basics")
    for n in range(1, 33):
        assert np.array_equal(np.array(O.alibi_slopes(n), dtype=np.float64), z[f"slopes_{n}"]), n
    assert torch.equal(O.alibi_mask(16, 8), torch.from_numpy(z["mask_h8_s16"]))
    assert torch.equal(O.alibi_mask(12, 6), torch.from_numpy(z["mask_h6_s12"]))
    assert torch.equal(O.causal_mask(9), torch.from_numpy(z["causal_9"]))


def _unified(name, seeded=False, activation="gelu"):
    z = gu.load(name)
    sd = gu.seeded_state_dict(z) if seeded else gu.state_dict(z)
    sd = O.make_trainable(sd)
    H = int(z["cfg"][3])
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    logits = O.unified_forward(sd, x, H, activation)
    assert gu.rel_err(logits, z["logits"]) < TOL
    loss, _ = O.unified_criterion(logits, y, 1)
    assert abs(float(loss) - float(z["loss"])) < TOL * abs(float(z["loss"]))
    loss.backward()
    for k, g in gu.grads(z).items():
        assert gu.rel_err(sd[k].grad, g) < 2e-5, k


@pytest.mark.parametrize("name", ["unified_tiny", "unified_heads6", "unified_heads12"])
def test_unified_small(name):
    _unified(name)


def test_unified_relu():
    _unified("unified_relu", activation="relu")


@pytest.mark.parametrize("name", ["unified_dh64_s256", "unified_dh128_s256", "unified_dh64_s320_l2"])
def test_unified_seeded(name):
    _unified(name, seeded=True)


def test_separated():
    z = gu.load("separated_tiny")
    sd = O.make_trainable(gu.state_dict(z))
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    preds = O.separated_forward(sd, x, 8)
    for p, k in zip(preds, ("logit_n", "logit_v", "logit_d")):
        assert gu.rel_err(p, z[k]) < TOL
    wn, wv = O.separated_class_weights(92, 130)
    assert torch.equal(wn, torch.from_numpy(z["crit_weight_n"]))
    assert torch.equal(wv, torch.from_numpy(z["crit_weight_v"]))
    loss, ld = O.separated_criterion(preds, y, wn, wv, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2})
    for k in ("loss", "loss_notes", "loss_velocity", "loss_duration"):
        assert abs(float(ld[k]) - float(z[k])) < TOL * abs(float(z[k])), k
    loss.backward()
    for k, g in gu.grads(z).items():
        assert gu.rel_err(sd[k].grad, g) < 2e-5, k


def test_unified_build_config_full_depth():
    """The oracle at configs[1]'s full depth and length (tests/golden/make_golden_r2.py)."""
    z = gu.load("unified_build_l6_s2048")
    sd = gu.seeded_state_dict(z)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    with torch.no_grad():
        logits = O.unified_forward(sd, x, 8)
        loss, _ = O.unified_criterion(logits, y, 1)
    got = logits[0, torch.from_numpy(z["rows"])]
    assert float((got - torch.from_numpy(z["logits_rows"])).abs().max()) < TOL * float(z["logits_absmax"])
    assert abs(float(loss) - float(z["loss"])) < TOL * abs(float(z["loss"]))


def test_separated_build_width():
    z = gu.load("separated_dh128_s256")
    sd = O.make_trainable(gu.seeded_state_dict(z))
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    preds = O.separated_forward(sd, x, 8)
    for p, k in zip(preds, ("logit_n", "logit_v", "logit_d")):
        assert gu.rel_err(p, z[k]) < TOL
    wn, wv = O.separated_class_weights(92, 130)
    loss, ld = O.separated_criterion(preds, y, wn, wv, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2})
    for k in ("loss", "loss_notes", "loss_velocity", "loss_duration"):
        assert abs(float(ld[k]) - float(z[k])) < TOL * abs(float(z[k])), k
    loss.backward()
    for k, g in gu.grads(z).items():
        assert gu.rel_err(sd[k].grad, g) < 2e-5, k


def test_relative_tiny():
    z = gu.load("relative_tiny")
    out = O.skew_relative_logits(torch.from_numpy(z["skew.q"]), torch.from_numpy(z["skew.E"]), 64)
    assert gu.rel_err(out, z["skew.out"]) < 1e-6
    # closed form of the skew: rel[i,j<=i] = q_i . E[max_seq-1-(i-j)]
    q, E = torch.from_numpy(z["skew.q"]), torch.from_numpy(z["skew.E"])
    i, j = 17, 5
    want = (q[1, 3, i] * E[3, 64 - 1 - (i - j)]).sum()
    assert abs(float(out[1, 3, i, j]) - float(want)) < 1e-5
    for S in (48, 64):
        sd = O.make_trainable(gu.state_dict(z))
        x, y = torch.from_numpy(z[f"s{S}.x"]), torch.from_numpy(z[f"s{S}.y"])
        logits = O.relative_forward(sd, x, 8, 64)
        assert gu.rel_err(logits, z[f"s{S}.logits"]) < TOL
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 618), y.reshape(-1))
        assert abs(float(loss) - float(z[f"s{S}.loss"])) < TOL * float(z[f"s{S}.loss"])
        loss.backward()
        for k, g in gu.grads(z, f"s{S}.grad.").items():
            assert gu.rel_err(sd[k].grad, g) < 2e-5, k


def test_relative_seeded():
    z = gu.load("relative_dh64_s320")
    sd = O.make_trainable(gu.seeded_state_dict(z))
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    logits = O.relative_forward(sd, x, 4, 384)
    assert gu.rel_err(logits, z["logits"]) < TOL
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, 618), y.reshape(-1))
    loss.backward()
    for k, g in gu.grads(z).items():
        assert gu.rel_err(sd[k].grad, g) < 2e-5, k


def test_train_step():
    z = gu.load("unified_train3")
    sd = O.make_trainable(gu.state_dict(z, "sd0."))
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, weight_decay=1e-4)
    for step in range(3):
        x = torch.from_numpy(z["x"][step])
        y = torch.roll(x, -1, dims=1)
        y[:, -2:] = 1
        loss = O.unified_train_step(sd, opt, x, y, 8)
        assert abs(loss - float(z["losses"][step])) < 1e-5 * abs(loss)
    for k, v in gu.state_dict(z, "sd3.").items():
        # Adam normalises the step, so near-zero gradients amplify rounding: the bound
        # is absolute, 2e-5 against a maximum possible movement of 3 * lr = 3e-3.
        assert float((sd[k].detach() - v).abs().max()) < 2e-5, k


def test_greedy_unified():
    z = gu.load("greedy_unified")
    sd = gu.state_dict(z)
    fwd = lambda s: O.unified_forward(sd, s, 8)
    for idx in (1, 9):
        seq = torch.from_numpy(z[f"idx{idx}.seq_in"].copy())
        out = O.sample_next_plus_one(fwd, seq, idx=idx, deterministic=True)
        assert torch.equal(out, torch.from_numpy(z[f"idx{idx}.seq_out"]))
    for idx in (0, 9):
        seq = torch.from_numpy(z[f"last{idx}.seq_in"].copy())
        out = O.sample_last_position(fwd, seq, idx=idx, deterministic=False)
        assert torch.equal(out, torch.from_numpy(z[f"last{idx}.seq_out"]))


def test_greedy_separated():
    z = gu.load("greedy_separated")
    sd = gu.state_dict(z)
    fwd = lambda s: O.separated_forward(sd, s, 8)
    for idx in (1, 9):
        seq = torch.from_numpy(z[f"idx{idx}.seq_in"].copy())
        out = O.sample_next_plus_one_separated(fwd, seq, idx=idx, deterministic=True)
        assert torch.equal(out, torch.from_numpy(z[f"idx{idx}.seq_out"]))


def test_synthetic_stream_is_valid_vocab():
    t = O.synthetic_unified_tokens(3, 257, seed=7)
    assert t.shape == (3, 257) and t.min() >= 0 and t.max() < 618 and (t[:, 0] == 0).all()


@pytest.mark.parametrize("name,fwd", [("alibi_tied_tiny", "tied"), ("rotary_tiny", "rotary"), ("rotary_dh64_s320", "rotary")])
def test_old_model_variants(name, fwd):
    """old_models/alibi.py (tied embedding) and old_models/rotary_transformer.py: logits, loss and every
    gradient of the oracle against the reference's."""
    z = gu.load(name)
    sd = O.make_trainable({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.") and k != "sd.pos_embed.emb"})
    x = torch.from_numpy(z["x"])
    heads = 8 if "tiny" in name else 4
    logits = O.tied_alibi_forward(sd, x, heads) if fwd == "tied" else O.rotary_forward(sd, x, heads)
    assert gu.rel_err(logits, z["logits"]) < TOL
    loss = torch.nn.functional.cross_entropy(logits.transpose(-1, -2), torch.roll(x, -1, 1))
    assert abs(float(loss.detach()) - float(z["loss"])) < TOL * abs(float(z["loss"]))
    loss.backward()
    for k in z.files:
        if k.startswith("grad."):
            assert gu.rel_err(sd[k[5:]].grad, z[k]) < 2 * TOL, k
    if fwd == "rotary":                                   # the sin | cos buffer of FixedPositionalEmbedding
        cos, sin = O.rotary_tables(z["sd.pos_embed.emb"].shape[0], z["sd.pos_embed.emb"].shape[1])
        assert torch.equal(torch.cat((sin, cos), -1), torch.from_numpy(z["sd.pos_embed.emb"]))
