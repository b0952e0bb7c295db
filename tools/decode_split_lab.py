"""Lab: one sampling step as NSPLIT independent batch slices on NSPLIT streams inside one HIP graph (the projections of
one slice run under the decode attention of another) vs the single-chain step.
    python tools/decode_split_lab.py [new_tokens] [nsplit]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicgeneration_amd.model_baseline import build_everything_unified as cfg
from musicgeneration_amd.decode import sample_next_plus_one, KVCacheDecoder, GraphedSampler

new = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 64
L = 64 + new + 1
dev = torch.device("cuda:0")
args = cfg.m_args.get_copy()
args.max_sequence = L
args.dtype = "bf16"
torch.manual_seed(4321)
model, _ = cfg.build_model_and_criterion(args)
model = model.to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1234)
seq = torch.zeros(B, L, dtype=torch.int32, device=dev)
seq[:, 1:64] = torch.randint(2, 618, (B, 63), generator=g, device=dev, dtype=torch.int32)
warm = seq[:, :80].clone()
sample_next_plus_one(model, warm, idx=64, deterministic=True)
torch.cuda.synchronize()

ref = seq.clone()
t0 = time.perf_counter()
sample_next_plus_one(model, ref, idx=64, deterministic=True)
torch.cuda.synchronize()
t_ref = time.perf_counter() - t0
steps = L - 1 - 64
print(f"single chain: {t_ref / steps * 1e3:.4f} ms/step  {B * steps / t_ref:.0f} tok/s")

out = seq.clone()
idx = 64
per = B // nsplit
samplers = []
with torch.no_grad():
    for k in range(nsplit):
        part = out[k * per:(k + 1) * per]
        dec = KVCacheDecoder(model, per, L)
        dec.rows(part[:, 0:idx - 1], 0)
        samplers.append(GraphedSampler(dec, part, idx, True, capture=False))
    torch.cuda.synchronize()
    side = [torch.cuda.Stream(device=dev) for _ in range(nsplit - 1)]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        cur = torch.cuda.current_stream(dev)
        for s in side:
            s.wait_stream(cur)
        for s, smp in zip(side, samplers[1:]):
            with torch.cuda.stream(s):
                smp._body()
        samplers[0]._body()
        for s in side:
            cur.wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    t_split = time.perf_counter() - t0
print(f"{nsplit} slices:    {t_split / steps * 1e3:.4f} ms/step  {B * steps / t_split:.0f} tok/s   tokens equal: {bool(torch.equal(out, ref))}")
