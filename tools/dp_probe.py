"""One-GPU probe of the data-parallel path (bench.py --force-dp): per-step wall time and caching-allocator counters.
usage: python tools/dp_probe.py [flat|inplace] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from musicgeneration_amd.optim import FusedAdamW
from musicgeneration_amd.data_parallel import GradientBucketer
mode = sys.argv[1] if len(sys.argv) > 1 else "flat"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29518")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(4321)
model, crit, data, *_ = bench.build_workload("train_unified", 16, 2048, "bf16", dev)
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
ddp = GradientBucketer(model, mode=mode) if mode != "none" else None
batches = [data.get() for _ in range(4)]
model.train()
if os.environ.get('PROBE_NOGC'):
    import gc
    gc.collect(); gc.freeze(); gc.disable()
sync = not os.environ.get('PROBE_NOSYNC')
for i in range(steps):
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, y = batches[i % 4]
    loss, _ = crit(model(x), y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if ddp is not None:
        ddp.finish()
    opt.step()
    if sync:
        torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"step {i}: {1e3 * (time.perf_counter() - t0):7.2f} ms  device_alloc {st.get('num_device_alloc')} device_free {st.get('num_device_free')} "
          f"retries {st.get('num_alloc_retries')} reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB", flush=True)
dist.destroy_process_group()
