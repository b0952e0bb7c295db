"""Three-stream sampling with the reference's signature
(model_baseline/inference_separated.py:18-28) on the KV-cached decoder."""
import torch

from ..decode import sample_next_plus_one

try:
    from tqdm import tqdm
except Exception:                       # pragma: no cover
    tqdm = None


def dummy_model(x: torch.Tensor):
    """inference_separated.py:12-15: all-zero logits of the legacy head sizes."""
    b, _, n = x.shape
    mk = lambda l: torch.zeros(b, n, l, dtype=x.dtype, device=x.device)
    return mk(25), mk(63), mk(166)


def inference(self, seq, idx=1, deterministic=False):
    return sample_next_plus_one(self, seq, idx=idx, deterministic=deterministic, progress=tqdm)
