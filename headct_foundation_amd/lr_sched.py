"""Learning-rate schedule of the MAE path, stepped once per iteration.

Contract (reference: src/utils/lr_sched.py:18-55 for the curve, :127-139 for the factory): the rate climbs linearly from 0
to the optimizer's initial rate over `num_warmup_steps` iterations, then follows `lr_end + (lr0 - lr_end) * (1 + cos(2*pi*
num_cycles * t)) / 2` with t = fraction of the post-warm-up iterations done, never below 0.  The object handed back is a
`torch.optim.lr_scheduler.LambdaLR`, so its `state_dict()` is what reference checkpoints carry under "scheduler".
Pure host arithmetic: one Python float per step.
"""
import math

from torch.optim.lr_scheduler import LambdaLR


class WarmupCosine:
    """Multiplicative factor lr(step) / lr0.  A callable object rather than a closure: LambdaLR stores its attributes
    (five numbers) in `state_dict()["lr_lambdas"]`; a reference checkpoint holds `[None]` there, which LambdaLR skips."""

    def __init__(self, lr0: float, lr_end: float, warmup: int, total: int, cycles: float):
        self.lr0, self.lr_end = float(lr0), float(lr_end)
        self.warmup, self.total, self.cycles = int(warmup), int(total), float(cycles)

    def __call__(self, step: int) -> float:
        if step < self.warmup:
            return step / max(1, self.warmup)
        done = (step - self.warmup) / max(1, self.total - self.warmup)
        wave = 0.5 * (1.0 + math.cos(2.0 * math.pi * self.cycles * done))
        return max(0.0, (self.lr_end + (self.lr0 - self.lr_end) * wave) / self.lr0)


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    lr_end: float = 1e-6, last_epoch: int = -1):
    lr0 = optimizer.defaults["lr"]
    if lr_end >= lr0:
        raise ValueError(f"cosine schedule needs lr_end < initial lr, got lr_end={lr_end} and lr={lr0}")
    return LambdaLR(optimizer, WarmupCosine(lr0, lr_end, num_warmup_steps, num_training_steps, num_cycles), last_epoch)


def get_lr_scheduler(config, optimizer, num_warmup_steps, total_steps, min_lr):
    """TRAIN.SCHEDULER -> scheduler.  The MAE recipe uses "cosine"; the reference's other branches (poly, and a constant
    branch that cannot run) are outside the hot path and rejected here."""
    kind = config.TRAIN.SCHEDULER
    if kind != "cosine":
        raise ValueError(f"Scheduler {kind} not supported")
    return get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, total_steps, lr_end=min_lr)
