"""Synthetic token batches with the duck type the training loop expects from
the reference's Ray-backed Dataset (data/datasetv2.py:69-97: ``get() -> (x, y)``
and ``__len__``), generated on the device from a seeded torch.Generator.

Unified stream grammar (mimics data/transforms.py:12-38,149-152; vocab 618 =
89 notes + 127 velocities + 200 durations + 200 time shifts + BOS(0) + EOS/pad(1)):
position 0 is BOS, the body is a random mix of (note, velocity, duration)
triples and single time-shift tokens, the last 0-10 % of a row is padding.
"""
from __future__ import annotations

import torch


class SyntheticUnifiedDataset:
    def __init__(self, batch_size: int, seq_len: int, steps_per_epoch: int = 100, seed: int = 1234,
                 device="cuda", vocab: int = 618):
        self.batch_size, self.seq_len, self.steps, self.vocab = batch_size, seq_len, steps_per_epoch, vocab
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def _stream(self, B, L):
        g, dev = self.gen, self.device
        # slot k of a triple cycle: 0 note, 1 velocity, 2 duration; ~25 % of cycles are a lone time shift
        n_cyc = L // 3 + 2
        lone = torch.rand(B, n_cyc, generator=g, device=dev) < 0.25
        note = 2 + torch.randint(0, 89, (B, n_cyc), generator=g, device=dev)
        vel = 91 + torch.randint(0, 127, (B, n_cyc), generator=g, device=dev)
        dur = 218 + torch.randint(0, 200, (B, n_cyc), generator=g, device=dev)
        shift = 418 + torch.randint(0, 200, (B, n_cyc), generator=g, device=dev)
        trip = torch.stack([torch.where(lone, shift, note), vel, dur], -1)          # [B,n_cyc,3]
        keep = torch.stack([torch.ones_like(lone), ~lone, ~lone], -1)
        flat = trip.reshape(B, -1)
        order = torch.argsort((~keep.reshape(B, -1)).to(torch.int8), dim=1, stable=True)   # kept tokens first, in order
        body = torch.gather(flat, 1, order)[:, :L - 1]
        toks = torch.cat([torch.zeros(B, 1, dtype=torch.int64, device=dev), body], 1)
        pad = torch.randint(0, L // 10 + 1, (B, 1), generator=g, device=dev)
        pos = torch.arange(L, device=dev)[None]
        return torch.where(pos >= L - pad, torch.ones_like(toks), toks)

    def get(self):
        """(x, y) int64 [B,S] with y = x shifted left by one (next-token targets)."""
        t = self._stream(self.batch_size, self.seq_len + 1)
        return t[:, :-1].contiguous(), t[:, 1:].contiguous()


class SyntheticSeparatedDataset:
    """[B,3,S] rows: note in [2,91), velocity in [2,129), duration in [2,202); column 0 is 0; a tail of 1s."""

    def __init__(self, batch_size: int, seq_len: int, steps_per_epoch: int = 100, seed: int = 1234, device="cuda",
                 vocabs=(92, 130, 203)):
        self.batch_size, self.seq_len, self.steps, self.vocabs = batch_size, seq_len, steps_per_epoch, vocabs
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def __len__(self):
        return self.steps

    def get(self):
        B, L, g, dev = self.batch_size, self.seq_len + 1, self.gen, self.device
        rows = [torch.randint(2, v - 1, (B, L), generator=g, device=dev) for v in self.vocabs]
        t = torch.stack(rows, 1)
        t[:, :, 0] = 0
        pad = torch.randint(0, L // 10 + 1, (B, 1, 1), generator=g, device=dev)
        pos = torch.arange(L, device=dev)[None, None]
        t = torch.where(pos >= L - pad, torch.ones_like(t), t)
        return t[:, :, :-1].contiguous(), t[:, :, 1:].contiguous()
