"""Round-2 golden vectors at the BASELINE configurations, again by RUNNING THE REAL REFERENCE
(see make_golden.py for the import set-up; this script shares its helpers).

    python tests/golden/make_golden_r2.py

* unified_build_l6_s2048 — configs[1]'s model at its full depth and sequence length:
  AlibiTransformer(618, 1024, 6, 8, 2048, 4096, dropout=0), RandomState(206) weights, B=1, S=2048.
  Stored: the seed, the tokens, the loss and the logits of every 8th position plus the last 128
  (384 rows x 618: the whole tensor would be 5 MB).
* separated_dh128_s256 — configs[2]'s model width (embeddings 256/256/512 -> d=1024, 8 heads of 128),
  one layer, S=256, tokens 1 and 2 present: three logit tensors + the loss dict + the small gradients.
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
sys.path.insert(0, os.path.dirname(HERE))

from model_baseline import alibi_unified as ref_uni         # noqa: E402
from model_baseline import alibi_separated as ref_sep       # noqa: E402
import golden_util as gu                                    # noqa: E402

torch.set_num_threads(8)


def param_spec(module):
    return np.array(json.dumps([(k, list(p.shape)) for k, p in module.named_parameters()]))


def load_seeded(module, seed):
    """The same RandomState(seed) weights the tests regenerate (golden_util.seeded_state_dict)."""
    z = {"param_spec": param_spec(module), "weight_seed": np.array(seed)}
    sd = gu.seeded_state_dict(z)
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(sd[k])


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


# ---- configs[1] at full depth / length -------------------------------------------------------
V, d, L, H, S, ff = 618, 1024, 6, 8, 2048, 4096
torch.manual_seed(60)
m = ref_uni.AlibiTransformer(vocab=V, embed_dim=d, n_layers=L, n_heads=H, max_sequence=S, proj_forward=ff, dropout=0.0)
load_seeded(m, 206)
m.eval()
g = torch.Generator().manual_seed(61)
x = torch.randint(0, V, (1, S), generator=g)
y = torch.roll(x, -1, dims=1)
y[:, -40:] = 1
with torch.no_grad():
    logits = m(x)
    loss, _ = ref_uni.Criterion(1)(logits, y)
rows = np.unique(np.concatenate([np.arange(0, S, 8), np.arange(S - 128, S)]))
save("unified_build_l6_s2048", x=x.numpy(), y=y.numpy(), rows=rows, logits_rows=logits[0, rows].numpy(),
     logits_absmax=np.array(float(logits.abs().max())), loss=loss.numpy(), weight_seed=np.array(206),
     param_spec=param_spec(m), cfg=np.array([V, d, L, H, S, ff]))
del m, logits

# ---- configs[2] width --------------------------------------------------------------------------
torch.manual_seed(62)
cfg = (92, 130, 203, 256, 256, 512, 1, 8, 256, 512)
sep = ref_sep.AlibiTransformer(*cfg, dropout=0.0)
load_seeded(sep, 207)
crit = ref_sep.Criterion(92, 130, 203, {"loss_notes": 1, "loss_velocity": 0.5, "loss_duration": 2})
g = torch.Generator().manual_seed(63)
B, S = 2, 256
x = torch.stack([torch.randint(2, 92, (B, S), generator=g), torch.randint(2, 130, (B, S), generator=g),
                 torch.randint(2, 203, (B, S), generator=g)], 1)
x[:, :, 0] = 0
x[0, :, -20:] = 1                      # EOS / pad tail on one row
x[1, :, 17] = 2                        # the down-weighted class
x[0, 0, 5] = 2
y = torch.roll(x, -1, dims=2)
y[:, :, -1] = 1
preds = sep(x)
loss, ld = crit(preds, y)
loss.backward()
small = ("proj_note.bias", "proj_velocity.bias", "proj_duration.bias", "norm.g", "norm.b", "embed_note.weight",
         "embed_velocity.weight", "embed_duration.weight", "proj_note.weight", "attn.c_attn.b", "linear2.bias")
gr = {"grad." + k: p.grad.numpy().copy() for k, p in sep.named_parameters() if k.endswith(small)}
save("separated_dh128_s256", x=x.numpy(), y=y.numpy(), logit_n=preds[0].detach().numpy(), logit_v=preds[1].detach().numpy(),
     logit_d=preds[2].detach().numpy(), loss=loss.detach().numpy(), loss_notes=ld["loss_notes"].detach().numpy(),
     loss_velocity=ld["loss_velocity"].detach().numpy(), loss_duration=ld["loss_duration"].detach().numpy(),
     weight_seed=np.array(207), param_spec=param_spec(sep), cfg=np.array(cfg), **gr)
print("done")
