"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Run once in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

It imports base_layers / alibi_unified / alibi_separated / relative_transformer
from /root/reference (torchmetrics is not installed, so a 6-line stub module
is put in sys.modules first: only the unused ``Metrics`` class needs it),
drives them with seeded inputs and stores inputs + outputs as .npz.  The
reference sources never travel; only these data files do.

The sampling scripts (inference_unified.py / inference_separated.py) cannot be
imported (their imports download a dataset and need ray/wandb), so their
10-line loops are restated here around the *reference model objects*.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

tm = types.ModuleType("torchmetrics")
class _Metric(torch.nn.Module):
    pass
def _Accuracy(**kw):
    return None
tm.Metric, tm.Accuracy = _Metric, _Accuracy
sys.modules["torchmetrics"] = tm
sys.path.insert(0, REF)

import base_layers as ref_base                              # noqa: E402
from model_baseline import alibi_unified as ref_uni         # noqa: E402
from model_baseline import alibi_separated as ref_sep       # noqa: E402
from old_models import relative_transformer as ref_rel      # noqa: E402

torch.set_num_threads(8)


def sd_np(module, drop=("mask",)):
    return {k: v.detach().numpy().copy() for k, v in module.state_dict().items() if k not in drop}


def grads_np(module):
    return {"grad." + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def param_spec(module):
    """JSON [(name, shape)] in named_parameters() order: the draw order of seeded_weights."""
    return np.array(json.dumps([(k, list(p.shape)) for k, p in module.named_parameters()]))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


def seeded_weights(module, seed):
    """Overwrite every parameter with RandomState(seed) draws so that big
    fixtures only need to store the seed (tests regenerate the weights)."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for k, p in module.named_parameters():
            if k.endswith(("norm1.weight", "norm2.weight", "norm.g")):
                a = 1.0 + 0.1 * rs.standard_normal(p.shape)
            elif k.endswith(("bias", ".b")):
                a = 0.02 * rs.standard_normal(p.shape)
            elif k.endswith(".E"):
                a = 0.05 * rs.standard_normal(p.shape)
            elif "embed" in k:
                a = 0.5 * rs.standard_normal(p.shape)
            else:
                fan_in = p.shape[-2] if p.dim() == 3 else p.shape[-1]
                a = rs.standard_normal(p.shape) / np.sqrt(fan_in)
            p.copy_(torch.from_numpy(a.astype(np.float32)))


# ---- slopes and mask -------------------------------------------------------
slopes = {f"slopes_{n}": np.array(ref_uni.get_slopes(n), dtype=np.float64) for n in range(1, 33)}
save("alibi_basics",
     mask_h8_s16=ref_uni.construct_alibi(16, 8).numpy(),
     mask_h6_s12=ref_uni.construct_alibi(12, 6).numpy(),
     causal_9=ref_base.generate_square_subsequent_mask(9, "cpu").numpy(),
     **slopes)


# ---- unified ---------------------------------------------------------------
def unified_case(name, seed, B, S, store_sd=True, weight_seed=None, grads="all", **kw):
    torch.manual_seed(seed)
    m = ref_uni.AlibiTransformer(dropout=0.0, **kw)
    if weight_seed is not None:
        seeded_weights(m, weight_seed)
    crit = ref_uni.Criterion(1)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randint(0, kw["vocab"], (B, S), generator=g)
    y = torch.roll(x, -1, dims=1)
    y[:, -3:] = 1                                     # exercised ignore_index
    m.train()
    logits = m(x)
    loss, _ = crit(logits, y)
    loss.backward()
    arrays = dict(x=x.numpy(), y=y.numpy(), logits=logits.detach().numpy(), loss=loss.detach().numpy(),
                  cfg=np.array([kw["vocab"], kw["embed_dim"], kw["n_layers"], kw["n_heads"],
                                kw["max_sequence"], kw["proj_forward"]]))
    if store_sd:
        arrays.update({"sd." + k: v for k, v in sd_np(m).items()})
    else:
        arrays["weight_seed"] = np.array(weight_seed)
        arrays["param_spec"] = param_spec(m)
    gr = grads_np(m)
    if grads != "all":
        gr = {k: v for k, v in gr.items() if any(k.endswith(s) for s in grads)}
    arrays.update(gr)
    save(name, **arrays)
    return m


uni_tiny = unified_case("unified_tiny", 0, 2, 48, vocab=618, embed_dim=64, n_layers=2, n_heads=8,
                        max_sequence=64, proj_forward=128)
unified_case("unified_heads6", 1, 2, 40, vocab=618, embed_dim=96, n_layers=1, n_heads=6,
             max_sequence=64, proj_forward=128)
unified_case("unified_heads12", 2, 2, 40, vocab=618, embed_dim=96, n_layers=1, n_heads=12,
             max_sequence=64, proj_forward=128)
unified_case("unified_relu", 3, 2, 33, vocab=618, embed_dim=64, n_layers=1, n_heads=4,
             max_sequence=64, proj_forward=128, activation="relu")
big_grads = ("embed.weight", "attn.c_attn.b", "attn.c_proj.b", "norm1.weight", "norm1.bias",
             "linear1.bias", "linear2.bias", "norm2.weight", "norm2.bias", "norm.g", "norm.b",
             "proj_class.bias")
unified_case("unified_dh64_s256", 4, 2, 256, store_sd=False, weight_seed=104, grads=big_grads,
             vocab=618, embed_dim=512, n_layers=1, n_heads=8, max_sequence=256, proj_forward=512)
unified_case("unified_dh128_s256", 5, 2, 256, store_sd=False, weight_seed=105, grads=big_grads,
             vocab=618, embed_dim=1024, n_layers=1, n_heads=8, max_sequence=256, proj_forward=512)
unified_case("unified_dh64_s320_l2", 6, 1, 320, store_sd=False, weight_seed=106, grads=big_grads,
             vocab=618, embed_dim=256, n_layers=2, n_heads=4, max_sequence=512, proj_forward=512)

# ---- training step (train_utils.py:160-173 body, fp32, scaler disabled) -----
torch.manual_seed(10)
m = ref_uni.AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=8, max_sequence=64,
                             proj_forward=128, dropout=0.0)
crit = ref_uni.Criterion(1)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
scaler = torch.cuda.amp.GradScaler(enabled=False)
arrays = {"sd0." + k: v for k, v in sd_np(m).items()}
g = torch.Generator().manual_seed(11)
losses = []
xs = []
for step in range(3):
    x = torch.randint(0, 618, (2, 48), generator=g)
    y = torch.roll(x, -1, dims=1)
    y[:, -2:] = 1
    xs.append(x.numpy())
    m.train()
    pred = m(x)
    total_loss, loss_dict = crit(pred, y)
    assert not torch.isnan(total_loss).any()
    opt.zero_grad()
    scaler.scale(total_loss).backward()
    scaler.step(opt)
    scaler.update()
    losses.append(float(total_loss))
arrays.update({"sd3." + k: v for k, v in sd_np(m).items()})
arrays["x"] = np.stack(xs)
arrays["losses"] = np.array(losses, dtype=np.float64)
arrays["grad_scaler_state_keys"] = np.array(sorted(scaler.state_dict().keys()))
save("unified_train3", **arrays)

# ---- separated -------------------------------------------------------------
torch.manual_seed(20)
sep = ref_sep.AlibiTransformer(92, 130, 203, 16, 16, 32, 2, 8, 64, 128, dropout=0.0)
sep_crit = ref_sep.Criterion(92, 130, 203, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2})
g = torch.Generator().manual_seed(21)
B, S = 2, 48
x = torch.stack([torch.randint(0, 92, (B, S), generator=g), torch.randint(0, 130, (B, S), generator=g),
                 torch.randint(0, 203, (B, S), generator=g)], 1)
x[:, :, 0] = 0
x[:, :, -5:] = 1
x[0, :, 7] = 2
y = torch.roll(x, -1, dims=2)
y[:, :, -1] = 1
preds = sep(x)
loss, ld = sep_crit(preds, y)
loss.backward()
arrays = dict(x=x.numpy(), y=y.numpy(), logit_n=preds[0].detach().numpy(), logit_v=preds[1].detach().numpy(),
              logit_d=preds[2].detach().numpy(), loss=loss.detach().numpy(),
              loss_notes=ld["loss_notes"].detach().numpy(), loss_velocity=ld["loss_velocity"].detach().numpy(),
              loss_duration=ld["loss_duration"].detach().numpy(),
              crit_weight_n=sep_crit.weight_n.numpy(), crit_weight_v=sep_crit.weight_v.numpy(),
              crit_state_keys=np.array(sorted(sep_crit.state_dict().keys())))
arrays.update({"sd." + k: v for k, v in sd_np(sep).items()})
arrays.update(grads_np(sep))
save("separated_tiny", **arrays)

# ---- relative --------------------------------------------------------------
torch.manual_seed(30)
rel = ref_rel.RelativeTransformer(618, 64, 2, 8, 64, 128, dropout=0.0)
arrays = {"sd." + k: v for k, v in sd_np(rel).items()}
for S in (48, 64):
    g = torch.Generator().manual_seed(31 + S)
    x = torch.randint(0, 618, (2, S), generator=g)
    y = torch.roll(x, -1, dims=1)
    rel.zero_grad()
    logits = rel(x)
    loss = torch.nn.functional.cross_entropy(logits.transpose(-1, -2), y)
    loss.backward()
    arrays.update({f"s{S}.x": x.numpy(), f"s{S}.y": y.numpy(), f"s{S}.logits": logits.detach().numpy(),
                   f"s{S}.loss": loss.detach().numpy()})
    arrays.update({f"s{S}." + k: v for k, v in grads_np(rel).items()})
# the skew alone, on a random q and E (relative_transformer.py:18-32)
att = rel.decoder_layers[0].attn
g = torch.Generator().manual_seed(33)
q = torch.randn(2, 8, 40, 8, generator=g)
arrays["skew.q"] = q.numpy()
arrays["skew.E"] = att.E.detach().numpy()
arrays["skew.out"] = att.relative_attn(q, att.E, 2, 40).detach().numpy()
save("relative_tiny", **arrays)

torch.manual_seed(34)
rel = ref_rel.RelativeTransformer(618, 256, 1, 4, 384, 256, dropout=0.0)
seeded_weights(rel, 134)
g = torch.Generator().manual_seed(35)
x = torch.randint(0, 618, (2, 320), generator=g)
y = torch.roll(x, -1, dims=1)
logits = rel(x)
loss = torch.nn.functional.cross_entropy(logits.transpose(-1, -2), y)
loss.backward()
gr = {k: v for k, v in grads_np(rel).items() if k.endswith((".E", "c_attn.b", "norm.g", "embedding"))}
save("relative_dh64_s320", x=x.numpy(), y=y.numpy(), logits=logits.detach().numpy(), loss=loss.detach().numpy(),
     weight_seed=np.array(134), param_spec=param_spec(rel), cfg=np.array([618, 256, 1, 4, 384, 256]), **gr)


# ---- sampling loops --------------------------------------------------------
def loop_next_plus_one(model, seq, idx=1, deterministic=False):
    # restates inference_unified.py:12-21 (tqdm dropped)
    with torch.no_grad():
        for i in range(idx, seq.shape[-1] - 1):
            n = model(seq)
            if not deterministic:
                r = torch.multinomial(n[:, i + 1].softmax(-1), num_samples=1)
            else:
                r = torch.argmax(n[:, i + 1], -1)
            seq[:, i] = r
    return seq


def loop_next_plus_one_sep(model, seq, idx=1, deterministic=False):
    # restates inference_separated.py:18-28
    with torch.no_grad():
        for i in range(idx, seq.shape[-1] - 1):
            n, v, d = model(seq)
            if not deterministic:
                r = map(lambda t: torch.multinomial(t[:, i + 1].softmax(-1), num_samples=1), [n, v, d])
            else:
                r = map(lambda t: torch.argmax(t[:, i + 1], -1), [n, v, d])
            r = torch.stack(list(r), -1)
            seq[:, :, i] = r
    return seq


import tqdm as _tqdm                                         # noqa: E402
ref_uni.tqdm = lambda it: it                                 # silence progress bars
ref_sep.tqdm = lambda it: it

torch.manual_seed(40)
gm = ref_uni.AlibiTransformer(vocab=618, embed_dim=64, n_layers=2, n_heads=8, max_sequence=64,
                              proj_forward=128, dropout=0.0)
seeded_weights(gm, 140)          # non-degenerate weights -> well separated argmax
gm.eval()
arrays = {"sd." + k: v for k, v in sd_np(gm).items()}
g = torch.Generator().manual_seed(41)
primer = torch.randint(2, 618, (3, 8), generator=g).to(torch.int32)
for idx in (1, 9):
    seq = torch.zeros(3, 40, dtype=torch.int32)
    seq[:, 1:9] = primer
    arrays[f"idx{idx}.seq_in"] = seq.numpy().copy()
    out = loop_next_plus_one(gm, seq, idx=idx, deterministic=True)
    arrays[f"idx{idx}.seq_out"] = out.numpy().copy()
for idx in (0, 9):
    seq = torch.zeros(3, 24, dtype=torch.int32)
    seq[:, 1:9] = primer
    arrays[f"last{idx}.seq_in"] = seq.numpy().copy()
    out = gm.inference(seq, idx=idx, deterministic=False)     # argmax branch (flag inverted)
    arrays[f"last{idx}.seq_out"] = out.numpy().copy()
save("greedy_unified", **arrays)

torch.manual_seed(50)
gs = ref_sep.AlibiTransformer(92, 130, 203, 16, 16, 32, 2, 8, 64, 128, dropout=0.0)
seeded_weights(gs, 150)
gs.eval()
arrays = {"sd." + k: v for k, v in sd_np(gs).items()}
g = torch.Generator().manual_seed(51)
primer = torch.stack([torch.randint(2, 92, (2, 8), generator=g), torch.randint(2, 130, (2, 8), generator=g),
                      torch.randint(2, 203, (2, 8), generator=g)], 1).to(torch.int32)
for idx in (1, 9):
    seq = torch.zeros(2, 3, 32, dtype=torch.int32)
    seq[:, :, 1:9] = primer
    arrays[f"idx{idx}.seq_in"] = seq.numpy().copy()
    out = loop_next_plus_one_sep(gs, seq, idx=idx, deterministic=True)
    arrays[f"idx{idx}.seq_out"] = out.numpy().copy()
seq = torch.zeros(2, 3, 20, dtype=torch.int32)
seq[:, :, 1:9] = primer
arrays["last0.seq_in"] = seq.numpy().copy()
arrays["last0.seq_out"] = gs.inference(seq, idx=0, deterministic=False).numpy().copy()
save("greedy_separated", **arrays)
print("done")
