"""Learning-rate schedule of the MAE path, stepped once per iteration.

Contract (reference: src/utils/lr_sched.py:18-55 for the curve, :127-139 for the factory): the rate climbs linearly from 0
to the optimizer's initial rate over `num_warmup_steps` iterations, then follows `lr_end + (lr0 - lr_end) * (1 + cos(2*pi*
num_cycles * t)) / 2` with t = fraction of the post-warm-up iterations done, never below 0.  The object handed back is a
`torch.optim.lr_scheduler.LambdaLR`, so its `state_dict()` is what reference checkpoints carry under "scheduler".
Pure host arithmetic: one Python float per step.
"""
import math

from torch.optim.lr_scheduler import LambdaLR


def warmup_cosine(lr0: float, lr_end: float, warmup: int, total: int, cycles: float):
    """Multiplicative factor lr(step) / lr0 as a plain closure: LambdaLR.state_dict() stores `None` for functions (only callable
    OBJECTS have their attributes saved), exactly what a reference checkpoint holds under scheduler["lr_lambdas"] -- so a resume
    with a changed MAX_EPOCHS / PER_WARMUP / LR follows the NEW run's curve, as the reference's does, instead of silently
    restoring the old one's constants."""
    lr0, lr_end, warmup, total, cycles = float(lr0), float(lr_end), int(warmup), int(total), float(cycles)

    def factor(step: int) -> float:
        if step < warmup:
            return step / max(1, warmup)
        done = (step - warmup) / max(1, total - warmup)
        wave = 0.5 * (1.0 + math.cos(2.0 * math.pi * cycles * done))
        return max(0.0, (lr_end + (lr0 - lr_end) * wave) / lr0)

    return factor


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    lr_end: float = 1e-6, last_epoch: int = -1):
    lr0 = optimizer.defaults["lr"]
    if lr_end >= lr0:
        raise ValueError(f"cosine schedule needs lr_end < initial lr, got lr_end={lr_end} and lr={lr0}")
    return LambdaLR(optimizer, warmup_cosine(lr0, lr_end, num_warmup_steps, num_training_steps, num_cycles), last_epoch)


def get_lr_scheduler(config, optimizer, num_warmup_steps, total_steps, min_lr):
    """TRAIN.SCHEDULER -> scheduler.  The MAE recipe uses "cosine"; the reference's other branches (poly, and a constant
    branch that cannot run) are outside the hot path and rejected here."""
    kind = config.TRAIN.SCHEDULER
    if kind != "cosine":
        raise ValueError(f"Scheduler {kind} not supported")
    return get_cosine_schedule_with_warmup(optimizer, num_warmup_steps, total_steps, lr_end=min_lr)
