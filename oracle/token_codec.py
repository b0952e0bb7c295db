"""CPU restatement (numpy) of the reference's unified token codec.  TEST INFRASTRUCTURE ONLY:
imported by tests/ and nothing else; the product path is musicgeneration_amd/codec.py on libmgx.

Follows (reference file:line):
  flatten_notes            data/transforms.py:11-38
  UnifiedEncoding.pad      data/transforms.py:149-152
  window + pad             data/transforms.py:172-179
  UnifiedReconstruct.reconstruct_stream   data/transforms.py:339-365
  unflatten_notes_         data/transforms.py:94-128
  SeparatedReconstruct.reconstruct_stream data/transforms.py:290-318
  stream_to_raw_stream_    data/parse_midi_numba.py:189-211
  compute_lin_bin          data/parse_midi_numba.py:251-252

Parity pinned: tests/golden/codec_*.npz were produced by tests/golden/make_golden_codec.py, which ran the
reference's own functions (numba / ray / py_midicsv stubbed out as identity decorators: the decorated
bodies are plain numpy) on seeded inputs; tests/test_codec_cpu.py checks this file against them.
"""
import numpy as np

NOTE_BINS, VELOCITY_BINS, DURATION_BINS, TICKS = 89, 127, 200, 480


def duration_bin_edges(duration_bins=DURATION_BINS):
    return np.cumsum(np.full(duration_bins, 5.0 / duration_bins, dtype=np.float64))


def duration_ticks_table(duration_bins=DURATION_BINS, ticks=TICKS):
    """(bin_arr[d] * ticks).astype(int64): the only floating-point step of the decode, done once on the host."""
    return (duration_bin_edges(duration_bins) * ticks).astype(np.int64)


def flatten_notes(notes, velocities, durations, note_bins=NOTE_BINS, velocity_bins=VELOCITY_BINS,
                  duration_bins=DURATION_BINS):
    shift_v, shift_nd, shift_d = note_bins, note_bins + velocity_bins, note_bins + velocity_bins + duration_bins
    out = []
    for n, v, d in zip(notes, velocities, durations):
        if n == 0:
            out.append(shift_d + d)
        else:
            out += [n, shift_v + v, shift_nd + d]
    return np.asarray(out, dtype=np.asarray(notes).dtype)


def encode_row(notes, velocities, durations, seq_len, start=0, **bins):
    """flatten -> window [start, start + seq_len - 2) when longer -> +2 -> [0] data [1 ...] (length seq_len)."""
    data = flatten_notes(notes, velocities, durations, **bins)
    if data.shape[0] > seq_len - 2:
        data = data[start:start + seq_len - 2]
    data = data + 2
    pad = seq_len - data.shape[0] - 2
    return np.concatenate([[0], data, np.ones(1 + pad, dtype=data.dtype)]).astype(data.dtype)


def reconstruct_events(tokens, note_bins=NOTE_BINS, velocity_bins=VELOCITY_BINS, duration_bins=DURATION_BINS,
                       ticks=TICKS):
    """tokens [L] -> MidiStream.events [N,3] = (note or -1, velocity, duration in ticks)."""
    arr = np.array(tokens, dtype=np.int64)
    if (arr == 0).any():
        arr = arr[1:]
    if (arr == 1).any():
        arr = arr[:np.argmax(arr == 1)]
    arr = arr - 2
    shift_v, shift_nd, shift_d = note_bins, note_bins + velocity_bins, note_bins + velocity_bins + duration_bins
    lo = 0
    while lo < arr.shape[0] and not (arr[lo] < note_bins):
        lo += 1
    if lo == arr.shape[0]:
        return None                      # the reference raises IndexError here (no note token at all)
    bn, bv, bd = [], [], []
    for x in arr[lo:]:
        if x < note_bins:
            bn.append(min(max(x, 0), note_bins - 1))
        elif shift_v <= x < shift_nd:
            bv.append(min(max(x - shift_v, 0), velocity_bins - 1))
        elif shift_nd <= x < shift_d:
            bd.append(min(max(x - shift_nd, 0), duration_bins - 1))
        else:
            bn.append(0)
            bv.append(0)
            bd.append(min(max(x - shift_d, 0), duration_bins - 1))
    t = min(len(bn), len(bv), len(bd))
    ev = np.asarray([bn[:t], bv[:t], bd[:t]], dtype=np.int64).reshape(3, t)
    ev[0] -= 1
    keep, = np.where((ev[1] != 0) | (ev[0] == -1))
    ev = ev[:, keep]
    ev[2] = duration_ticks_table(duration_bins, ticks)[ev[2]]
    return ev.transpose(1, 0)


def stream_to_raw_events(events):
    """MidiStream.events [N,3] -> RawMidiStream.events [2*notes,4] = (1, abs_time, key, velocity), stable by time."""
    events = np.asarray(events)
    out = []
    t = 0
    for n, v, d in events:
        if n == -1:
            t += d
        else:
            out.append([1, t, n + 21, v])
            out.append([1, t + d, n + 21, 0])
    out = np.asarray(out, dtype=events.dtype).reshape(-1, 4)
    return out[np.argsort(out[:, 1], kind="mergesort")]


def reconstruct_events_separated(tokens, note_bins=NOTE_BINS, velocity_bins=VELOCITY_BINS, duration_bins=DURATION_BINS,
                                 ticks=TICKS):
    """tokens [3,L] (note, velocity, duration rows) -> MidiStream.events [N,3] (transforms.py:290-318)."""
    arr = np.array(tokens, dtype=np.int64)
    if (arr == 0).any():
        arr = arr[:, 1:]
    if (arr == 1).any():
        arr = arr[:, :np.argmax(arr[0] == 1)]        # first 1 of the NOTE row; 0 (empty result) when it has none
    arr = arr - 2
    arr[0] = np.clip(arr[0], 0, note_bins)
    arr[1] = np.clip(arr[1], 0, velocity_bins)
    arr[2] = np.clip(arr[2], 0, duration_bins - 1)
    arr[0] -= 1
    keep, = np.where((arr[1] != 0) | (arr[0] == -1))
    arr = arr[:, keep]
    arr[2] = duration_ticks_table(duration_bins, ticks)[arr[2]]
    return arr.transpose(1, 0)
