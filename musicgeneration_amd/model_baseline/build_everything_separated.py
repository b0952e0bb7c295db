"""Hyper-parameters of the reference's separated run
(model_baseline/build_everything_separated.py:18-32,41-48,61-71,90-97) plus a
side-effect-free `build()`; see build_everything_unified.py."""
import torch

from ..misc import EasyDict
from .alibi_separated import Metrics, build_model_and_criterion

m_args = EasyDict()
m_args.vocab_note = 89 + 3
m_args.vocab_velocity = 127 + 3
m_args.vocab_duration = 200 + 3
m_args.embed_note = 256
m_args.embed_velocity = 256
m_args.embed_duration = 512
m_args.n_layers = 10
m_args.n_heads = 8
m_args.max_sequence = 1024
m_args.proj_forward = 4096
m_args.dropout = 0.1
m_args.activation = "gelu"
m_args.loss_weights = {"loss_notes": 1, "loss_velocity": 1, "loss_duration": 1}

opt_args = EasyDict()
opt_args.lr = 1e-4
opt_args.weight_decay = 1e-4
lr_drop = 50

d_args = EasyDict()
d_args.batch_size = 8
d_args.seq_len = 1025

train_args = EasyDict()
train_args.epochs = 120
train_args.log_scalar_metric_step = 100 * d_args.batch_size
train_args.checkpoint_step = 500 * d_args.batch_size
train_args.batch_size = d_args.batch_size
train_args.mixed_precision = True


def build(model_args=None, optimizer_cls=torch.optim.AdamW, device="cuda"):
    args = (model_args or m_args).get_copy()
    model, criterion = build_model_and_criterion(args)
    model, criterion = model.to(device), criterion.to(device)
    optimizer = optimizer_cls(model.parameters(), **opt_args)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, lr_drop)
    return model, criterion, optimizer, lr_scheduler, Metrics()
