"""Sampling entry point with the reference's signature
(model_baseline/inference_unified.py:12-21): ``inference(model, seq, idx=1,
deterministic=False) -> seq`` filling `seq` in place, now KV-cached (O(L) per
token instead of a full re-forward).  The notebook cells of the reference
(checkpoint load, MIDI write-out) are not part of the hot path."""
from ..decode import sample_next_plus_one

try:
    from tqdm import tqdm
except Exception:                       # pragma: no cover
    tqdm = None


def inference(self, seq, idx=1, deterministic=False):
    return sample_next_plus_one(self, seq, idx=idx, deterministic=deterministic, progress=tqdm)
