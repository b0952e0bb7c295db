"""Decode/encode timing of the device token codec at the decode bench shape (B=64 rows of 4161 tokens)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd.codec import UnifiedReconstruct, encode_unified
from oracle import token_codec as C
B, L = 64, 4161
rng = np.random.default_rng(0)
tok = rng.integers(2, 618, (B, L)); tok[:, 0] = 0; tok[:, -1] = 1
t = torch.from_numpy(tok).cuda().int()
rc = UnifiedReconstruct()
for _ in range(3): rc.decode_batch(t)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): rc.decode_batch(t)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
by = B * L * 4 * (1 + 3 + 1)            # tokens in, events + abs_time out (upper bound: all kept)
print(f"decode  B={B} L={L}: {us:7.1f} us  {B*L/us:7.1f} Mtok/s  {by/us/1e3:6.1f} GB/s")
t0 = time.perf_counter()
for b in range(4): C.reconstruct_events(tok[b])
cpu = (time.perf_counter() - t0) / 4
print(f"oracle (numpy/python loop, 1 core): {cpu*1e3:7.1f} ms per row -> {L/cpu/1e6:6.3f} Mtok/s")
N = 3000
notes = torch.from_numpy(rng.integers(0, 89, (B, N))).cuda().int(); vel = torch.from_numpy(rng.integers(1, 127, (B, N))).cuda().int()
dur = torch.from_numpy(rng.integers(0, 200, (B, N))).cuda().int()
ln = torch.full((B,), N).cuda().int(); st = torch.zeros(B).cuda().int()
for _ in range(3): encode_unified(notes, vel, dur, ln, st, 2049)
torch.cuda.synchronize(); s.record()
for _ in range(20): encode_unified(notes, vel, dur, ln, st, 2049)
e.record(); torch.cuda.synchronize()
print(f"encode  B={B} N={N} -> seq 2049: {s.elapsed_time(e)/20*1e3:7.1f} us")
