"""Golden vectors for the unified token codec, produced by RUNNING THE REFERENCE's own functions.

    python tests/golden/make_golden_codec.py        (build container only: needs /root/reference)

data/transforms.py and data/parse_midi_numba.py import ray, numba and py_midicsv, which are not installed;
they are only used as decorators (@ray.remote, @njit) or by the MIDI file reader/writer, so identity-decorator
stubs are put in sys.modules first.  The decorated bodies are plain numpy and run as written.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ident = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
nb = types.ModuleType("numba"); nb.njit = ident
ray = types.ModuleType("ray"); ray.remote = ident
sys.modules.update({"numba": nb, "ray": ray, "py_midicsv": types.ModuleType("py_midicsv")})
sys.path.insert(0, "/root/reference")
import data.transforms as T                    # noqa: E402
import data.parse_midi_numba as P              # noqa: E402

rng = np.random.default_rng(20261111)
out = {}


def grammar_stream(n_events):
    """(notes 0..88 with 0 = time shift, velocities, binned durations) like UnifiedEncoding after digitize"""
    notes = rng.integers(1, 89, n_events)
    shift = rng.random(n_events) < 0.35
    notes[shift] = 0
    vel = rng.integers(0, 127, n_events)
    vel[rng.random(n_events) < 0.1] = 0          # velocity-0 notes are dropped by the reconstruction
    vel[shift] = 0
    dur = rng.integers(0, 200, n_events)
    return notes.astype(np.int64), vel.astype(np.int64), dur.astype(np.int64)


enc = T.UnifiedEncoding(seq_len=256)
cases = []
for k, (n_events, seq_len, start) in enumerate([(40, 256, 0), (200, 256, 17), (1, 64, 0), (700, 1024, 333), (85, 256, 0)]):
    notes, vel, dur = grammar_stream(n_events)
    flat = T.flatten_notes(notes, vel, dur)
    assert np.array_equal(flat, T.flatten_notes_(notes, vel, dur))
    enc.seq_len = seq_len
    data = flat
    if data.shape[0] > seq_len - 2:               # transforms.py:172-176 with the random idx fixed
        data = data[start:start + seq_len - 2]
    row = enc.pad(data.copy())
    out[f"enc{k}.notes"], out[f"enc{k}.vel"], out[f"enc{k}.dur"] = notes, vel, dur
    out[f"enc{k}.meta"] = np.array([seq_len, start])
    out[f"enc{k}.flat"], out[f"enc{k}.row"] = flat, row
    cases.append(row)

rc = T.UnifiedReconstruct()
# decode inputs: the encoded rows above, plus "generated-like" rows: random tokens (misaligned triples), a row
# that starts mid-note (truncated window), rows without BOS / without EOS
dec_rows = list(cases)
dec_rows.append(np.concatenate([[0], rng.integers(2, 618, 300)]).astype(np.int64))
dec_rows.append(np.concatenate([rng.integers(2, 618, 120), [1, 1, 1, 5, 7]]).astype(np.int64))
dec_rows.append(rng.integers(2, 618, 64).astype(np.int64))
dec_rows.append(cases[3][5:900].copy())
for k, row in enumerate(dec_rows):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        stream = rc.reconstruct_stream(row)
    raw = P.stream_to_raw_stream_(stream.events) if stream.events.shape[0] else np.zeros((0, 4), np.int64)
    out[f"dec{k}.tokens"] = row
    out[f"dec{k}.events"] = stream.events
    out[f"dec{k}.raw"] = raw
out["n_enc"] = np.array(len(cases)); out["n_dec"] = np.array(len(dec_rows))
out["bin_arr"] = rc.bin_arr
np.savez_compressed(os.path.join(HERE, "codec_unified.npz"), **out)
print("wrote codec_unified.npz", len(cases), "encode cases,", len(dec_rows), "decode cases")

# ---- separated stream (SeparatedReconstruct, transforms.py:277-323) ----------------------------------------
sep = {}
src = T.SeparatedReconstruct()
rows3 = []
for k, L in enumerate([48, 300, 1023, 64, 64, 64]):
    a = np.stack([rng.integers(2, 92, L), rng.integers(2, 130, L), rng.integers(2, 203, L)]).astype(np.int64)
    a[1, rng.random(L) < 0.15] = 2                 # velocity 0 after the -2 shift: dropped unless a time shift
    a[0, rng.random(L) < 0.3] = 2                  # note 0 -> -1 after the shifts: a time shift
    if k != 3:
        a[:, 0] = 0                                # BOS column
    if k in (0, 1, 2):
        a[:, L - L // 5:] = 1                      # EOS / padding tail
    if k == 4:
        a[2, 10] = 1                               # a 1 outside the note row: argmax(arr[0] == 1) == 0 -> empty result
    if k == 5:
        a[0, 20] = 1
    rows3.append(a)
for k, a in enumerate(rows3):
    st = src.reconstruct_stream(a)
    raw = P.stream_to_raw_stream_(st.events) if st.events.shape[0] else np.zeros((0, 4), np.int64)
    sep[f"dec{k}.tokens"], sep[f"dec{k}.events"], sep[f"dec{k}.raw"] = a, st.events, raw
sep["n_dec"] = np.array(len(rows3))
np.savez_compressed(os.path.join(HERE, "codec_separated.npz"), **sep)
print("wrote codec_separated.npz", len(rows3), "decode cases")
