"""Golden vectors for the two remaining attention variants of old_models/ (tied-embedding ALiBi, rotary),
produced by running the reference modules.   python tests/golden/make_golden_old.py   (build container only)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from old_models import alibi as ref_alibi                  # noqa: E402
from old_models import rotary_transformer as ref_rot       # noqa: E402

torch.set_num_threads(8)


def run(name, model, x, drop):
    model.train()                                           # dropout=0: train == eval, grads flow
    logits = model(x)
    y = torch.roll(x, -1, 1)
    loss = torch.nn.functional.cross_entropy(logits.transpose(-1, -2), y)
    loss.backward()
    out = {"sd." + k: v.detach().numpy().copy() for k, v in model.state_dict().items() if k not in drop}
    out.update({"grad." + k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()})
    out.update(x=x.numpy(), logits=logits.detach().numpy(), loss=np.array(loss.item()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, tuple(logits.shape), float(loss))


torch.manual_seed(0)
m = ref_alibi.AlibiTransformer(n_vocab=618, embed_dim=64, d_model=64, n_layers=2, n_heads=8, max_sequence=64,
                               proj_forward=128, dropout=0.0)
run("alibi_tied_tiny", m, torch.randint(0, 618, (2, 48)), ("mask",))
torch.manual_seed(1)
m = ref_rot.RotaryTransformer(n_vocab=618, d_model=64, n_layers=2, n_heads=8, max_sequence=64, proj_forward=128, dropout=0.0)
run("rotary_tiny", m, torch.randint(0, 618, (2, 48)), ("mask",))
torch.manual_seed(2)
m = ref_rot.RotaryTransformer(n_vocab=618, d_model=256, n_layers=1, n_heads=4, max_sequence=320, proj_forward=512, dropout=0.0)
run("rotary_dh64_s320", m, torch.randint(0, 618, (2, 320)), ("mask",))
