"""Worker of the two-ranks-on-one-GPU data-parallel test (tests/test_models_gpu.py) and of tools/dp2_probe.py.

Started once per rank with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (no torchrun: the parent
owns the two PIDs and ends exactly those).  Both ranks use cuda:0.  Rank r trains the bf16 unified model on rows r::2
through GradientBucketer (bucket hooks on the side stream, weight gradients produced in their bucket slots, the global
CE normaliser) + FusedAdamW; after every step the averaged gradients, the averaged loss and the updated weights must
equal those of one process on the whole batch.

DP2_TRANSPORT = "gloo-gpu"  (GPU tensors handed to gloo) | "gloo-host" (a test-only process-group wrapper that stages
                every exchange through a host tensor on the calling stream; what carries the bytes is not under test)
DP2_WATCHDOG  = seconds after which every thread's Python stack is dumped to stderr and the process exits (a hang
                then names the collective each rank is blocked in).
"""
import faulthandler
import os
import sys

faulthandler.enable()
faulthandler.dump_traceback_later(float(os.environ.get("DP2_WATCHDOG", "45")), exit=True)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu                                                     # noqa: E402
from musicgeneration_amd import data_parallel, ops                           # noqa: E402
from musicgeneration_amd.data_parallel import GradientBucketer               # noqa: E402
from musicgeneration_amd.model_baseline.alibi_unified import AlibiTransformer, Criterion   # noqa: E402
from musicgeneration_amd.optim import FusedAdamW                             # noqa: E402

MODE = os.environ.get("DP2_MODE", "flat")
TRANSPORT = os.environ.get("DP2_TRANSPORT", "gloo-gpu")


def log(*a):
    print("rank", os.environ.get("RANK"), *a, flush=True)


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

if TRANSPORT == "gloo-host":
    class _HostWork:
        def wait(self):
            return True

    _all_reduce, _broadcast = dist.all_reduce, dist.broadcast

    def _staged_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        # stream-ordered on the caller's current stream: D2H (synchronising), CPU all-reduce, H2D
        h = t.detach().to("cpu", torch.float32)
        _all_reduce(h, op=op, group=group)
        t.copy_(h.to(t.dtype))
        return _HostWork() if async_op else None

    def _staged_broadcast(t, src=0, group=None, async_op=False):
        h = t.detach().to("cpu")
        _broadcast(h, src=src, group=group)
        t.copy_(h)
        return _HostWork() if async_op else None

    data_parallel.dist.all_reduce = _staged_all_reduce
    data_parallel.dist.broadcast = _staged_broadcast


def build(seed):
    torch.manual_seed(seed)
    return AlibiTransformer(vocab=96, embed_dim=128, n_layers=2, n_heads=2, max_sequence=256, proj_forward=256,
                            dropout=0.0, dtype="bf16").to(dev)


g = torch.Generator().manual_seed(11)
X = torch.randint(2, 96, (4, 192), generator=g).to(dev)
Y = torch.randint(2, 96, (4, 192), generator=g)
Y[0, 40:] = 1                                        # rank 0 sees far fewer non-pad targets than rank 1: the CE
Y[2, 100:] = 1                                       # normaliser has to be the global one
Y = Y.to(dev)
crit = Criterion(1).to(dev)
model = build(100 + rank)                            # different init per rank: the bucketer broadcasts rank 0's weights
log("model built")
ddp = GradientBucketer(model, bucket_mb=0.05, mode=MODE)
assert len(ddp.buckets) > 2
log("bucketer ready:", len(ddp.buckets), "buckets")


def _report():
    # a hang: which exchanges have completed, and what every native thread of this process is blocked in
    try:
        log("watchdog: bucket works completed:", [(b, bool(w.is_completed())) for b, w in list(ddp._work)])
    except Exception as e:                                                    # noqa: BLE001
        log("watchdog: work state unavailable:", repr(e))
    for tid in sorted(os.listdir("/proc/self/task")):
        try:
            comm = open(f"/proc/self/task/{tid}/comm").read().strip()
            wchan = open(f"/proc/self/task/{tid}/wchan").read().strip()
            sysc = open(f"/proc/self/task/{tid}/syscall").read().split()[0]
            log(f"watchdog: thread {tid} {comm!r} wchan={wchan} syscall={sysc}")
        except OSError:
            pass
    if _nbt is not None:                             # native frames of every thread (tools/native_bt: no debugger here)
        sys.stderr.flush()
        _nbt.nbt_dump_all()


_nbt = None
_nbt_path = os.path.join(ROOT, "tools", "native_bt", "libnative_bt.so")
if os.environ.get("DP2_NATIVE_STACKS") and os.path.exists(_nbt_path):
    import ctypes
    _nbt = ctypes.CDLL(_nbt_path)
    _nbt.nbt_install()

import threading                                                             # noqa: E402
_wd = threading.Timer(float(os.environ.get("DP2_WATCHDOG", "45")) - 8.0, _report)
_wd.daemon = True
_wd.start()
# eps = 1e-3: far above the gradients' rounding noise.  With the default 1e-8 Adam's first steps are sign-like, and the
# elements whose true gradient is zero (the key third of c_attn.b: softmax is shift-invariant) or at the level of
# the fp32 summation order turn a 1e-7 difference between "two half batches averaged" and "one full batch" into
# +-lr in the weights; the two trajectories then part through bf16 rounding flips and the step-1 comparison would
# test luck, not the exchange.
opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01, eps=1e-3)
ref = build(100)
ref_opt = FusedAdamW(ref.parameters(), lr=1e-3, weight_decay=0.01, eps=1e-3)
xs, ys = X[rank::world], Y[rank::world]
for it in range(2):
    log("step", it, "forward")
    loss, _ = crit(model(xs), ys)
    opt.zero_grad(set_to_none=True)
    log("step", it, "backward")
    loss.backward()
    log("step", it, "finish")
    ddp.finish()
    log("step", it, "exchanged")
    if it == 0 and MODE == "flat":                   # the block weights' gradients were produced in their bucket slots
        for n, p in model.named_parameters():
            assert p.grad.data_ptr() == p.mgx_grad_slot.data_ptr(), n
    # single-process reference on the whole batch (the local normaliser is the global one there)
    with ddp.local_loss_normaliser():
        rl, _ = crit(ref(X), Y)
    ref_opt.zero_grad(set_to_none=True)
    rl.backward()
    tot = loss.detach().float().cpu()
    dist.all_reduce(tot)
    assert abs(float(tot) / world - float(rl)) < 2e-3 * abs(float(rl)), (float(tot) / world, float(rl))
    if os.environ.get("DP2_VERBOSE"):
        for (n, p), q in zip(model.named_parameters(), ref.parameters()):
            log("step", it, "grad", n, "rel_err", gu.rel_err(p.grad, q.grad))
    for (n, p), q in zip(model.named_parameters(), ref.parameters()):
        assert gu.rel_err(p.grad, q.grad) < 1e-4, (it, n, gu.rel_err(p.grad, q.grad))
    opt.step()
    ref_opt.step()
    if os.environ.get("DP2_VERBOSE"):
        worst = max(((gu.rel_err(p.detach(), q.detach()), n) for (n, p), q in zip(model.named_parameters(), ref.parameters())))
        log("after step", it, "worst weight rel_err", worst)
        for (n, p), q in zip(model.named_parameters(), ref.parameters()):
            d = (p.detach() - q.detach()).abs()
            if float(d.max()) > 1e-6:
                g1, g2 = p.grad.reshape(-1), q.grad.reshape(-1)
                k = int(d.reshape(-1).argmax())
                log("   ", n, "max|dw|", float(d.max()), "n>1e-6:", int((d > 1e-6).sum()), "of", d.numel(), "grad there", float(g1[k]), float(g2[k]))
if os.environ.get("DP2_VERBOSE"):
    for (n, p), q in zip(model.named_parameters(), ref.parameters()):
        log("weight", n, "rel_err", gu.rel_err(p.detach(), q.detach()), "grad rel_err", gu.rel_err(p.grad, q.grad),
            "max|dw|", float((p.detach() - q.detach()).abs().max()))
for (n, p), q in zip(model.named_parameters(), ref.parameters()):
    assert gu.rel_err(p.detach(), q.detach()) < 1e-5, n
torch.cuda.synchronize()
ddp.close()
assert ops.LOSS_DENOMINATOR_HOOK is None
dist.barrier()
dist.destroy_process_group()
faulthandler.cancel_dump_traceback_later()
_wd.cancel()
log("ok")
