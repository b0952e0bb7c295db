"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce
on RCCL (torch.distributed backend "nccl") over xGMI, overlapped with backward.

The reference is single-process; this is new capability.  Design
for MI355X's point-to-point xGMI (7 links x ~153 GB/s per GPU): parameters are
grouped into ~32 MB buckets in reverse forward order; a bucket's averaged all-reduce
(RCCL's AVG; sum / world on gloo) is issued from the autograd hook of its last-arriving
gradient, so the exchange of layer k hides under the backward GEMMs of layers < k, and
`finish()` joins before the optimizer step.  Two exchange modes, chosen at construction:

  "flat" (default)  the bucket's gradients are packed into a persistent flat buffer (one multi-tensor
                    copy) that is all-reduced on a side stream; afterwards p.grad IS its slot of that
                    buffer (no scatter copies).  Public c10d API only.
  "inplace"         ONE grouped all-reduce over the gradient tensors where they lie (ncclGroupStart/End
                    through c10d's coalescing manager): no copy in either direction, ~0.4 ms per step
                    less on the unified model.  It uses a private torch API and has only run with one
                    RCCL rank and with two gloo ranks in this project, so it is opt-in; a failure raises
                    (there is no silent change of mode).

Works on CPU with gloo for tests (no streams there).  Multi-GPU RCCL runs are the driver's: on the
builder's one-GPU boxes the RCCL path has run with a world of one only.

The mean losses of the fused CE kernels normalise by the local sum of target weights (non-pad targets,
class weights: alibi_unified.py:161, alibi_separated.py:193-195).  Under data parallelism that makes the
averaged gradient a mean of per-rank means; `GradientBucketer` therefore installs `global_mean_denominator`
as ops.LOSS_DENOMINATOR_HOOK: every CE then divides by the all-reduced denominator / world, and the average
over ranks of the per-rank losses (and gradients) equals the single-process loss on the concatenated batch.
"""
from __future__ import annotations

import contextlib
from typing import List

import torch
import torch.distributed as dist


def global_mean_denominator(den: torch.Tensor, group=None) -> torch.Tensor:
    """den: this rank's CE denominators (sum of target weights), any shape, fp32.  Returns sum over ranks / world:
    mean over ranks of (num_r / returned) == sum_r num_r / sum_r den_r, the loss of the concatenated batch."""
    world = dist.get_world_size(group)
    out = den.clone()
    dist.all_reduce(out, group=group)
    return out / world


class GradientBucketer:
    def __init__(self, model: torch.nn.Module, bucket_mb: float = 32.0, process_group=None, tail_mb: float = 4.0,
                 mode: str = "flat", global_loss_normaliser: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if mode not in ("flat", "inplace"):
            raise ValueError(f"mode must be 'flat' or 'inplace', got {mode!r}")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        self.cuda = params[0].is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        # NCCL/RCCL averages in the collective; gloo has no AVG: pre-divide there
        self.avg = self.cuda and dist.get_backend(process_group) == "nccl"
        self.inplace = mode == "inplace"
        if self.inplace and not hasattr(dist, "_coalescing_manager"):
            raise RuntimeError("mode='inplace' needs torch.distributed._coalescing_manager; use mode='flat'")
        limit = int(bucket_mb * 1024 * 1024)
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(params):                       # backward produces gradients roughly in this order
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > limit:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        # The bucket that completes last (the first layers + the embedding) is exchanged with nothing left to
        # overlap: keep that exposed tail short by splitting its last ~tail_mb of gradients into a bucket of
        # their own (the rest of it then starts earlier, still under the backward of layer 0).
        tail_limit = int(tail_mb * 1024 * 1024)
        last = self.buckets[-1]
        if len(last) > 1 and sum(p.numel() * p.element_size() for p in last) > 2 * tail_limit:
            k, size = len(last), 0
            while k > 1 and size + last[k - 1].numel() * last[k - 1].element_size() <= tail_limit:
                k -= 1
                size += last[k].numel() * last[k].element_size()
            if 0 < k < len(last):
                self.buckets[-1:] = [last[:k], last[k:]]
        self._flat, self._views = [], []
        if not self.inplace:
            self._alloc_flat()
        self._bucket_of = {id(p): b for b, ps in enumerate(self.buckets) for p in ps}
        self._pending = [len(b) for b in self.buckets]
        self._work = []
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in params]
        # identical replicas: broadcast rank 0's weights once
        for p in model.parameters():
            dist.broadcast(p.data, src=0, group=process_group)
            torch.autograd.graph.increment_version(p)    # written through .data: version-keyed weight shadows must refresh
        for b in model.buffers():
            dist.broadcast(b.data, src=0, group=process_group)
        # The global CE normaliser is a blocking all-reduce issued from the criterion's forward: EVERY rank must
        # evaluate the criterion the same number of times with gradients enabled.  Evaluations under torch.no_grad()
        # (a validation loss on one rank) keep the local normaliser and exchange nothing.  close() (or leaving the
        # `with` block) restores whatever hook was installed before, so a later model / process group in the same
        # process does not inherit this one's group.
        self._hooked = False
        if global_loss_normaliser and self.world > 1:
            from . import ops
            self._prev_hook = ops.LOSS_DENOMINATOR_HOOK
            self._own_hook = self._denominator
            ops.LOSS_DENOMINATOR_HOOK = self._own_hook
            self._hooked = True

    def _denominator(self, den):
        return global_mean_denominator(den, self.group)

    def close(self):
        """Detach from the model's process-wide state: the CE denominator hook goes back to what it was, the
        gradient hooks and the bucket-slot tags of the parameters are removed.  Idempotent."""
        if self._hooked:
            from . import ops
            if ops.LOSS_DENOMINATOR_HOOK is self._own_hook:
                ops.LOSS_DENOMINATOR_HOOK = self._prev_hook
            self._hooked = False
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self.params:
            for tag in ("mgx_grad_slot", "mgx_slot_claimed"):
                if hasattr(p, tag):
                    delattr(p, tag)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @contextlib.contextmanager
    def local_loss_normaliser(self):
        """Criterion evaluations inside the block use the rank-local normaliser and exchange nothing (a reference
        loss on one rank, a side model)."""
        from . import ops
        saved, ops.LOSS_DENOMINATOR_HOOK = ops.LOSS_DENOMINATOR_HOOK, None
        try:
            yield
        finally:
            ops.LOSS_DENOMINATOR_HOOK = saved

    def _alloc_flat(self):
        # one persistent flat buffer per bucket, every parameter's slot 16-byte aligned; after the exchange
        # p.grad IS its slot (a view): no scatter copies, and the fused optimizer keeps its vector loads
        self._flat, self._views = [], []
        for ps in self.buckets:
            dt = ps[0].dtype
            if any(p.dtype != dt for p in ps):
                raise ValueError("a gradient bucket must hold one dtype")
            pad = max(1, 16 // ps[0].element_size())
            offs, off = [], 0
            for p in ps:
                offs.append(off)
                off += (p.numel() + pad - 1) // pad * pad
            flat = torch.zeros(off, dtype=dt, device=ps[0].device)
            self._flat.append(flat)
            self._views.append([flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, ps)])
            # ops.grad_slot: the fused decoder block writes its weight gradients straight into these slots
            for p, v in zip(ps, self._views[-1]):
                p.mgx_grad_slot = v

    def _hook(self, p):
        p.mgx_slot_claimed = False                       # ops.grad_slot: the slot may be handed out again next backward
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _exchange(self, b):
        ps, flat = self.buckets[b], self._flat[b]
        # gradients that were produced in place (ops.grad_slot) need no copy
        todo = [(v, p.grad) for v, p in zip(self._views[b], ps) if p.grad.data_ptr() != v.data_ptr()]
        if todo:
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])      # one multi-tensor launch
        if self.avg:
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        flat.div_(self.world)
        return dist.all_reduce(flat, group=self.group, async_op=True)

    def _exchange_inplace(self, b):
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM          # gloo: sum here, divide after the wait
        with dist._coalescing_manager(group=self.group, async_ops=True) as cm:
            for p in self.buckets[b]:
                dist.all_reduce(p.grad, op=op, group=self.group)
        return cm

    def _launch(self, b):
        if self.inplace:
            # c10d orders its RCCL stream after the current stream (the gradients' producer) at the call and the
            # caller's stream after the collective at wait(): no side stream, no copies.  Errors propagate.
            self._work.append((b, self._exchange_inplace(b)))
            return
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                work = self._exchange(b)
                for p in self.buckets[b]:
                    p.grad.record_stream(self.stream)
        else:
            work = self._exchange(b)
        self._work.append((b, work))

    def finish(self):
        """Wait for every bucket; p.grad becomes the parameter's slot of the averaged flat buffer."""
        for b, n in enumerate(self._pending):
            if n != 0 and n != len(self.buckets[b]):
                raise RuntimeError("some parameters of a bucket received no gradient")
            if n == len(self.buckets[b]) and any(p.grad is not None for p in self.buckets[b]):
                raise RuntimeError("bucket bookkeeping out of sync")
        if self.inplace:
            for b, work in self._work:
                work.wait()                              # orders the current stream after the collective
                if not self.avg:
                    torch._foreach_div_([p.grad for p in self.buckets[b]], self.world)
        else:
            for b, work in self._work:
                ctx = torch.cuda.stream(self.stream) if self.cuda else _null()
                with ctx:
                    work.wait()                          # orders the side stream after the collective
                for p, v in zip(self.buckets[b], self._views[b]):
                    p.grad = v
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
        self._work.clear()
        self._pending = [len(b) for b in self.buckets]


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
