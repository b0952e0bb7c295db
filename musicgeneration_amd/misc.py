"""EasyDict and the Noam scheduler of the reference's misc.py (misc.py:4-33)."""
import math


class EasyDict(dict):
    """dict with attribute access; `**args` splats and missing names raise AttributeError (misc.py:4-16)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"{name} not in dictionary") from None

    def __setattr__(self, name, value):
        self[name] = value

    def search_common_naming(self, name, seperator="_"):
        key = name + seperator
        return {k.replace(key, ""): v for k, v in self.items() if key in k}

    def get_copy(self):
        return EasyDict(self.copy())


class CustomScheduler:
    """lr = d_model^-1/2 * min(step^-1/2, step * warmup^-3/2)  (misc.py:19-33)."""

    def __init__(self, optimizer, d_model, warmup_steps, param_groups=(0,)):
        self.opt = optimizer
        self.d_model = d_model
        self.warmup_steps = warmup_steps
        self.param_groups = param_groups
        self.const = 1 / math.sqrt(d_model)

    def step(self, step):
        step += 1
        lr = self.const * min(1 / math.sqrt(step), step * self.warmup_steps ** -1.5)
        for g in self.param_groups:
            self.opt.param_groups[g]["lr"] = lr
