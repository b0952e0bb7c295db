"""Unified MIDI token codec on the GPU (mgx_unified_decode_events / mgx_unified_encode_tokens).

Mirrors the reference's data/transforms.py surface for the unified stream:
``UnifiedReconstruct.reconstruct_stream`` (tokens -> MidiStream events), ``flatten_notes`` + window +
``UnifiedEncoding.pad`` (events -> token rows), and ``stream_to_raw_stream`` (data/parse_midi_numba.py:189-215),
whose running time comes from the decode kernel and whose stable sort by time stays a host argsort.
Batched: whole [B, L] sampling buffers are decoded in one launch.  No CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from ._lib import lib, check
from .ops import _need, _p, _stream


@dataclass
class MidiStream:
    """data/parse_midi_numba.py:31-40: events [N,3] = (note | -1 for a time shift, velocity, duration ticks)."""
    ticks: int
    events: np.ndarray


@dataclass
class RawMidiStream:
    """data/parse_midi_numba.py:18-28: events [N,4] = (control id, absolute time, key, velocity)."""
    ticks: int
    events: np.ndarray


@dataclass
class UnifiedReconstruct:
    """data/transforms.py:326-370 (file writing needs py_midicsv and is left to the caller)."""
    ticks: int = 480
    duration_lin_bins: int = 200
    note_bins: int = 89
    velocity_bins: int = 127

    def __post_init__(self):
        # compute_lin_bin(0, 5, bins) cumulated (parse_midi_numba.py:251-252); the float64 -> int step of
        # transforms.py:362 is done here once, the kernel only indexes the table
        self.bin_arr = np.cumsum(np.full(self.duration_lin_bins, 5.0 / self.duration_lin_bins, dtype=np.float64))
        self._ticks_table = {}

    def _table(self, device):
        t = self._ticks_table.get(device)
        if t is None:
            t = torch.from_numpy((self.bin_arr * self.ticks).astype(np.int64).astype(np.int32)).to(device)
            self._ticks_table[device] = t
        return t

    def decode_batch(self, tokens: torch.Tensor):
        """tokens [B,L] (int32/int64, GPU) -> (events [B,L,3] int32, abs_time [B,L] int32, counts [B] int32)."""
        if tokens.dim() != 2:
            raise ValueError("tokens must be [B, L]")
        tk = tokens.to(torch.int32).contiguous()
        _need(tk, "tokens", torch.int32)
        B, L = tk.shape
        events = torch.empty(B, L, 3, dtype=torch.int32, device=tk.device)
        abs_time = torch.empty(B, L, dtype=torch.int32, device=tk.device)
        counts = torch.empty(B, dtype=torch.int32, device=tk.device)
        check(lib.mgx_unified_decode_events(_p(tk), B, L, _p(self._table(tk.device)), self.note_bins, self.velocity_bins,
                                            self.duration_lin_bins, _p(events), _p(abs_time), _p(counts), _stream()),
              "mgx_unified_decode_events")
        return events, abs_time, counts

    def reconstruct_streams(self, tokens: torch.Tensor) -> List[Optional[MidiStream]]:
        events, _, counts = self.decode_batch(tokens)
        ev, n = events.cpu().numpy(), counts.cpu().numpy()
        return [None if c < 0 else MidiStream(self.ticks, ev[b, :c].astype(np.int64)) for b, c in enumerate(n)]

    def reconstruct_stream(self, arr) -> MidiStream:
        """Single row, reference signature (numpy or tensor in, MidiStream out)."""
        t = torch.as_tensor(np.asarray(arr) if not isinstance(arr, torch.Tensor) else arr)
        if not t.is_cuda:
            t = t.cuda()
        s = self.reconstruct_streams(t.reshape(1, -1))[0]
        if s is None:
            raise IndexError("no note token in the sequence")      # what the reference's while loop raises
        return s

    def raw_streams(self, tokens: torch.Tensor) -> List[Optional[RawMidiStream]]:
        """stream_to_raw_stream of every row: note-on at the event's absolute time, note-off duration later,
        stably sorted by time."""
        events, abs_time, counts = self.decode_batch(tokens)
        ev, at, n = events.cpu().numpy().astype(np.int64), abs_time.cpu().numpy().astype(np.int64), counts.cpu().numpy()
        out = []
        for b, c in enumerate(n):
            if c < 0:
                out.append(None)
                continue
            e, t = ev[b, :c], at[b, :c]
            notes = e[:, 0] != -1
            k, t0, v, d = e[notes, 0] + 21, t[notes], e[notes, 1], e[notes, 2]
            raw = np.empty((2 * k.shape[0], 4), dtype=np.int64)
            raw[0::2] = np.stack([np.ones_like(k), t0, k, v], 1)
            raw[1::2] = np.stack([np.ones_like(k), t0 + d, k, np.zeros_like(k)], 1)
            out.append(RawMidiStream(self.ticks, raw[np.argsort(raw[:, 1], kind="mergesort")]))
        return out


@dataclass
class SeparatedReconstruct(UnifiedReconstruct):
    """data/transforms.py:277-323: the three-row (note, velocity, duration) stream of the separated model."""

    def decode_batch(self, tokens: torch.Tensor):
        """tokens [B,3,L] -> (events [B,L,3], abs_time [B,L], counts [B]) int32."""
        if tokens.dim() != 3 or tokens.shape[1] != 3:
            raise ValueError("tokens must be [B, 3, L]")
        tk = tokens.to(torch.int32).contiguous()
        _need(tk, "tokens", torch.int32)
        B, _, L = tk.shape
        events = torch.empty(B, L, 3, dtype=torch.int32, device=tk.device)
        abs_time = torch.empty(B, L, dtype=torch.int32, device=tk.device)
        counts = torch.empty(B, dtype=torch.int32, device=tk.device)
        check(lib.mgx_separated_decode_events(_p(tk), B, L, _p(self._table(tk.device)), self.note_bins,
                                              self.velocity_bins, self.duration_lin_bins, _p(events), _p(abs_time),
                                              _p(counts), _stream()), "mgx_separated_decode_events")
        return events, abs_time, counts

    def reconstruct_stream(self, arr) -> MidiStream:
        t = torch.as_tensor(np.asarray(arr) if not isinstance(arr, torch.Tensor) else arr)
        if not t.is_cuda:
            t = t.cuda()
        return self.reconstruct_streams(t.reshape(1, 3, -1))[0]


def encode_unified(notes: torch.Tensor, velocities: torch.Tensor, durations: torch.Tensor, lengths: torch.Tensor,
                   starts: torch.Tensor, seq_len: int, note_bins=89, velocity_bins=127, duration_bins=200):
    """[B,Nmax] binned events (+ lengths, window starts [B]) -> token rows [B,seq_len] int32."""
    args = [t.to(torch.int32).contiguous() for t in (notes, velocities, durations, lengths, starts)]
    for t, name in zip(args, ("notes", "velocities", "durations", "lengths", "starts")):
        _need(t, name, torch.int32)
    n, v, d, ln, st = args
    B, Nmax = n.shape
    if v.shape != n.shape or d.shape != n.shape or ln.numel() != B or st.numel() != B:
        raise ValueError("encode_unified: shape mismatch")
    out = torch.empty(B, seq_len, dtype=torch.int32, device=n.device)
    check(lib.mgx_unified_encode_tokens(_p(n), _p(v), _p(d), _p(ln), _p(st), B, Nmax, seq_len, note_bins, velocity_bins,
                                        duration_bins, _p(out), _stream()), "mgx_unified_encode_tokens")
    return out
