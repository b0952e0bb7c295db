"""Where does the data-parallel step spend host time?  One rank on RCCL: python tools/dp_dbg.py"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench
from musicgeneration_amd.optim import FusedAdamW
from musicgeneration_amd.data_parallel import GradientBucketer
model, crit, data, *_ = bench.build_workload("train_unified", 16, 2048, "bf16", torch.device("cuda", 0))
opt = FusedAdamW(model.parameters(), lr=1e-3)
ddp = GradientBucketer(model)
batches = [data.get() for _ in range(2)]
model.train()
T = {"fwd": 0.0, "bwd": 0.0, "finish": 0.0, "opt": 0.0}
for i in range(8):
    if i == 3:
        torch.cuda.synchronize(); T = {k: 0.0 for k in T}; t_all = time.perf_counter()
    x, y = batches[i % 2]
    t = time.perf_counter(); loss, _ = crit(model(x), y); opt.zero_grad(set_to_none=True); T["fwd"] += time.perf_counter() - t
    t = time.perf_counter(); loss.backward(); T["bwd"] += time.perf_counter() - t
    t = time.perf_counter(); ddp.finish(); T["finish"] += time.perf_counter() - t
    t = time.perf_counter(); opt.step(); T["opt"] += time.perf_counter() - t
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("host ms per step:", {k: round(v / 5 * 1e3, 2) for k, v in T.items()}, "wall", round(tot / 5 * 1e3, 2))
dist.destroy_process_group()
