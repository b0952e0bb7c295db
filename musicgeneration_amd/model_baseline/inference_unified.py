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


def reconstruct(seq, ticks: int = 480):
    """What inference_unified.py:51-55 does after sampling, without the numba / py_midicsv stack: the sampled
    buffer [B,L] (GPU) -> one RawMidiStream per row (note-on / note-off events sorted by time, ready for a MIDI
    writer) through the device token codec (musicgeneration_amd.codec).  The caller sets
    seq[:, -1] = 1 (EOS) first, as the reference script does."""
    from ..codec import UnifiedReconstruct
    return UnifiedReconstruct(ticks=ticks).raw_streams(seq)
