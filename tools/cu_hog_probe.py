"""How much does a resident foreign kernel (k CUs held for T us from the start of every backward, on a side stream)
cost the training step?  A one-GPU stand-in for RCCL's reduction kernels in an N > 1 run.
usage: python tools/cu_hog_probe.py [short]  (needs tools/lab/libcu_hog.so, see tools/lab/cu_hog.hip)"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from musicgeneration_amd.optim import FusedAdamW
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "libcu_hog.so"))
hog.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(4321)
model, crit, data, *_ = bench.build_workload("train_unified", 16, 2048, "bf16", dev)
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
batches = [data.get() for _ in range(4)]
model.train()
side = torch.cuda.Stream()
import gc


def run(k, usec, steps=10, warm=3):
    ts = []
    for i in range(steps + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, y = batches[i % 4]
        loss, _ = crit(model(x), y)
        opt.zero_grad(set_to_none=True)
        if k:
            side.wait_stream(torch.cuda.current_stream())
            rc = hog.cu_hog_launch(k, usec, ctypes.c_void_p(side.cuda_stream))
            assert rc == 0, rc
        loss.backward()
        if k:
            torch.cuda.current_stream().wait_stream(side)
        opt.step()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts[len(ts) // 2]


run(0, 0, 4, 2)
gc.collect(); gc.freeze()
base = run(0, 0)
print(f"no hog: {base:.2f} ms/step", flush=True)
short = len(sys.argv) > 1 and sys.argv[1] == "short"
for k in ((16,) if short else (8, 16, 32, 64)):
    for usec in (2000, 4000):
        t = run(k, usec)
        print(f"hog {k:3d} CUs x {usec} us: {t:.2f} ms/step (+{t - base:.2f})", flush=True)
print(f"no hog again: {run(0, 0):.2f} ms/step", flush=True)
