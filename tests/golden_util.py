"""Helpers for the committed golden vectors (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def state_dict(z, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def grads(z, prefix="grad."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def seeded_state_dict(z):
    """Regenerate the RandomState(weight_seed) weights make_golden.seeded_weights
    wrote into the reference module (same draw order and scaling rules)."""
    spec = json.loads(str(z["param_spec"]))
    rs = np.random.RandomState(int(z["weight_seed"]))
    sd = {}
    for k, shape in spec:
        if k.endswith(("norm1.weight", "norm2.weight", "norm.g")):
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif k.endswith(("bias", ".b")):
            a = 0.02 * rs.standard_normal(shape)
        elif k.endswith(".E"):
            a = 0.05 * rs.standard_normal(shape)
        elif "embed" in k:
            a = 0.5 * rs.standard_normal(shape)
        else:
            fan_in = shape[-2] if len(shape) == 3 else shape[-1]
            a = rs.standard_normal(shape) / np.sqrt(fan_in)
        sd[k] = torch.from_numpy(a.astype(np.float32))
    return sd


def rel_err(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
