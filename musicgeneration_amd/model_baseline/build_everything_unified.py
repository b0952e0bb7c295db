"""Hyper-parameters of the reference's unified run
(model_baseline/build_everything_unified.py:18-28,38-45,57-69,97-104) and a
`build()` that assembles model / criterion / optimizer / scheduler from them.
Unlike the reference module, importing this file has no side effects (no
dataset download, no Ray actors); data comes from musicgeneration_amd.synthetic
or any object with the same get()/__len__ duck type."""
import torch

from ..misc import EasyDict
from .alibi_unified import Metrics, build_model_and_criterion

m_args = EasyDict()
m_args.vocab = 89 + 127 + 200 * 2 + 2      # notes + velocities + duration bins + shift bins + bos + eos
m_args.embed_dim = 1024
m_args.n_layers = 6
m_args.n_heads = 8
m_args.max_sequence = 1024
m_args.proj_forward = 4096
m_args.dropout = 0.1
m_args.activation = "gelu"
m_args.pad_index = 1                        # also the eos token

opt_args = EasyDict()
opt_args.lr = 1e-3
opt_args.weight_decay = 1e-4
lr_drop = 50

d_args = EasyDict()
d_args.batch_size = 8
d_args.seq_len = 1025                       # one extra token: the loss is auto-regressive
d_args.duration_bins = 200
d_args.eval_batch_size = 2

train_args = EasyDict()
train_args.epochs = 120
train_args.log_scalar_metric_step = 100 * d_args.batch_size
train_args.checkpoint_step = 500 * d_args.batch_size
train_args.batch_size = d_args.batch_size
train_args.mixed_precision = False


def build(model_args=None, optimizer_cls=torch.optim.AdamW, device="cuda"):
    args = (model_args or m_args).get_copy()
    model, criterion = build_model_and_criterion(args)
    model, criterion = model.to(device), criterion.to(device)
    optimizer = optimizer_cls(model.parameters(), **opt_args)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, lr_drop)
    return model, criterion, optimizer, lr_scheduler, Metrics()
