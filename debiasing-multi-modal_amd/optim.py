"""SGD-with-momentum on the multi-tensor HIP kernel, plus the reference's optimiser / LR
helpers (/root/reference/demo/util.py:70-136) with the same signatures.

`SGD` is a torch.optim.Optimizer, so `param_group['lr']` edits by the schedule helpers,
`zero_grad()`, `state_dict()` and deepcopy behave as with torch.optim.SGD; only `step()` is
replaced: all parameters of a group are updated by ONE launch of dbmm_sgd_momentum
(g += wd*p; buf = g | mu*buf + g; p -= lr*buf -- dampening 0, no nesterov, which is how
the reference configures it).
"""
import math

import numpy as np
import torch

from . import ops


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            fresh, seen = ([], [], []), ([], [], [])
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                dst = fresh if first else seen
                dst[0].append(p); dst[1].append(p.grad.contiguous()); dst[2].append(st["momentum_buffer"])
            for (ps, gs, bs), first in ((fresh, True), (seen, False)):
                if ps:
                    ops.sgd_momentum(ps, gs, bs, group["lr"], group["momentum"], group["weight_decay"], first)
        return loss


def set_optimizer(opt, model):
    """demo/util.py:118-123."""
    return SGD(model.parameters(), lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay)


def set_optimizer_reg(opt, model, freeze_old=True):
    """demo/util.py:125-136: parameters whose *name* contains "old_cls" are not stepped."""
    params = [p for n, p in model.named_parameters() if "old_cls" not in n] if freeze_old else model.parameters()
    return SGD(params, lr=opt.learning_rate_reg, momentum=opt.momentum, weight_decay=opt.weight_decay)


def _set_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr


def _decayed(base_lr, args, epoch, span):
    if args.cosine:
        eta_min = base_lr * (args.lr_decay_rate ** 3)
        return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / span)) / 2
    steps = np.sum(epoch > np.asarray(args.lr_decay_epochs))
    return base_lr * (args.lr_decay_rate ** steps) if steps > 0 else base_lr


def adjust_learning_rate(args, optimizer, epoch):
    """demo/util.py:70-81 (step decay counts the global epoch index)."""
    _set_lr(optimizer, _decayed(args.learning_rate, args, epoch, args.epochs))


def adjust_learning_rate_reg(args, optimizer, epoch):
    """demo/util.py:83-96.  The reference's cosine branch reads a misspelt attribute
    (`epochs_feature_laerning`, :89) and raises AttributeError; that behaviour is kept."""
    span = (args.epochs - args.epochs_feature_laerning) if args.cosine else None
    _set_lr(optimizer, _decayed(args.learning_rate_reg, args, epoch, span))


def warmup_learning_rate(args, epoch, batch_id, total_batches, optimizer):
    """demo/util.py:99-106."""
    if args.warm and epoch <= args.warm_epochs:
        p = (batch_id + (epoch - 1) * total_batches) / (args.warm_epochs * total_batches)
        _set_lr(optimizer, args.warmup_from + p * (args.warmup_to - args.warmup_from))


def warmup_learning_rate_reg(args, epoch, batch_id, total_batches, optimizer):
    """demo/util.py:108-115."""
    if args.warm_reg and epoch <= args.warm_epochs_reg:
        p = (batch_id + (epoch - 1) * total_batches) / (args.warm_epochs_reg * total_batches)
        _set_lr(optimizer, args.warmup_from_reg + p * (args.warmup_to_reg - args.warmup_from_reg))


def get_lr(optimizer):
    for group in optimizer.param_groups:
        return group["lr"]


def set_seed(seed):
    """demo/util.py:61-68."""
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)
