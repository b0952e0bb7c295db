"""Fused AdamW on the mgx_adamw kernel with torch.optim.AdamW semantics and
state_dict layout (state keys step / exp_avg / exp_avg_sq), so checkpoints
written by either optimizer load into the other (train_utils.py:94, 'optmizer')."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            batch = {}                                   # step count -> ([p], [g], [m], [v]) for the multi-tensor launch
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype == torch.float32 and p.is_contiguous():
                    lists = batch.setdefault(int(st["step"]), ([], [], [], []))
                    for lst, t in zip(lists, (p, g, st["exp_avg"], st["exp_avg_sq"])):
                        lst.append(t)
                else:
                    ops.adamw_step(p, g, st["exp_avg"], st["exp_avg_sq"], None, float(group["lr"]), b1, b2,
                                   group["eps"], group["weight_decay"], int(st["step"]))
            for step, (ps, gs, ms, vs) in batch.items():
                ops.adamw_step_multi(ps, gs, ms, vs, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                     step)
            # the kernels wrote the parameters behind autograd's back: advance their version counters so that
            # everything keyed on them (the models' compute-dtype weight shadows, saved-tensor checks) sees the update
            for p in group["params"]:
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)
        return loss
