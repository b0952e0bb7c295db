"""Oracle (oracle/token_codec.py) vs golden vectors produced by the reference's own codec functions."""
import os

import numpy as np

from oracle import token_codec as C

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "codec_unified.npz"))


def test_bin_table_matches_reference():
    assert np.array_equal(C.duration_bin_edges(), Z["bin_arr"])


def test_encode_matches_reference():
    for k in range(int(Z["n_enc"])):
        notes, vel, dur = Z[f"enc{k}.notes"], Z[f"enc{k}.vel"], Z[f"enc{k}.dur"]
        seq_len, start = (int(v) for v in Z[f"enc{k}.meta"])
        assert np.array_equal(C.flatten_notes(notes, vel, dur), Z[f"enc{k}.flat"])
        row = C.encode_row(notes, vel, dur, seq_len, start)
        assert row.shape == (seq_len,) and np.array_equal(row, Z[f"enc{k}.row"])


def test_decode_matches_reference():
    for k in range(int(Z["n_dec"])):
        ev = C.reconstruct_events(Z[f"dec{k}.tokens"])
        assert np.array_equal(ev, Z[f"dec{k}.events"]), k
        assert np.array_equal(C.stream_to_raw_events(ev), Z[f"dec{k}.raw"]), k


def test_round_trip_of_a_clean_stream():
    rng = np.random.default_rng(1)
    notes = rng.integers(1, 89, 50); notes[2::4] = 0       # a leading time shift would be trimmed (transforms.py:105-108)
    vel = rng.integers(1, 127, 50); vel[notes == 0] = 0
    dur = rng.integers(0, 200, 50)
    row = C.encode_row(notes, vel, dur, 256)
    ev = C.reconstruct_events(row)
    assert np.array_equal(ev[:, 0], notes - 1) and np.array_equal(ev[:, 1], vel)
    assert np.array_equal(ev[:, 2], C.duration_ticks_table()[dur])


def test_separated_decode_matches_reference():
    zs = np.load(os.path.join(os.path.dirname(__file__), "golden", "codec_separated.npz"))
    for k in range(int(zs["n_dec"])):
        ev = C.reconstruct_events_separated(zs[f"dec{k}.tokens"])
        assert np.array_equal(ev, zs[f"dec{k}.events"].reshape(-1, 3)), k
        raw = C.stream_to_raw_events(ev) if ev.shape[0] else np.zeros((0, 4), np.int64)
        assert np.array_equal(raw, zs[f"dec{k}.raw"]), k
