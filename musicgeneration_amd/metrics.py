"""In-process replacement for the torchmetrics objects the reference's Metrics
classes own (alibi_unified.py:166-183): ``Accuracy(mdmc_reduce='samplewise')``
on arg-maxed predictions = mean over samples of the per-sample token accuracy."""
import torch


class SamplewiseAccuracy:
    def __init__(self):
        self.reset()

    def update(self, preds: torch.Tensor, target: torch.Tensor):
        preds = torch.as_tensor(preds)
        target = torch.as_tensor(target)
        if preds.shape != target.shape:
            raise ValueError(f"preds {tuple(preds.shape)} vs target {tuple(target.shape)}")
        per_sample = (preds == target).flatten(1).float().mean(1)
        self.total += float(per_sample.sum())
        self.count += per_sample.numel()

    def compute(self):
        return torch.tensor(self.total / max(self.count, 1))

    def reset(self):
        self.total = 0.0
        self.count = 0
