"""Optimiser factory and threshold schedule (reference: utils/train_helper.py:21-53,340-349)."""
import numpy as np

from . import optimizer


def get_optimizer(param_groups, args):
    """train_helper.py:21-53: groups 0/1 at lr, groups 2/3 at 10*lr, weight decay on every group."""
    return getattr(optimizer, args.optimizer)(
        params=[
            {"params": param_groups[0], "lr": args.lr, "weight_decay": args.wt_decay},
            {"params": param_groups[1], "lr": args.lr, "weight_decay": args.wt_decay},
            {"params": param_groups[2], "lr": args.lr * 10, "weight_decay": args.wt_decay},
            {"params": param_groups[3], "lr": args.lr * 10, "weight_decay": args.wt_decay},
        ],
        lr=args.lr, weight_decay=args.wt_decay, betas=args.betas, warmup_iter=args.warmup_iters,
        max_iter=args.max_iters, warmup_ratio=args.warmup_lr, power=args.power)


def cosine_descent(max_thres, min_thres, step, num_steps):
    """train_helper.py:340-349."""
    if step < 0:
        return max_thres
    if step >= num_steps:
        return min_thres
    interpolation_factor = step / (num_steps - 1)
    return max_thres + (min_thres - max_thres) * (1 - np.cos(np.pi * interpolation_factor)) / 2
