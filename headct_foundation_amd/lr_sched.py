"""Per-iteration LR schedules of the MAE path (src/utils/lr_sched.py:18-55,127-139). Host scalar math only."""
import math

from torch.optim.lr_scheduler import LambdaLR


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    lr_end: float = 1e-6, last_epoch: int = -1):
    """Linear warm-up to the optimizer's initial lr, then cosine decay to `lr_end` (lr_sched.py:18-55)."""
    lr_init = optimizer.defaults["lr"]
    if not (lr_init > lr_end):
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        lr_range = lr_init - lr_end
        progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        lr_new = lr_end + lr_range * 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress))
        lr_new /= lr_init
        return max(0.0, lr_new)

    return LambdaLR(optimizer, lr_lambda, last_epoch)


def get_lr_scheduler(config, optimizer, num_warmup_steps, total_steps, min_lr):
    """lr_sched.py:127-139; only the cosine branch is on the MAE hot path (poly/constant are out of scope;
    the reference's constant branch is itself broken)."""
    if config.TRAIN.SCHEDULER == "cosine":
        return get_cosine_schedule_with_warmup(optimizer, num_warmup_steps=num_warmup_steps,
                                               num_training_steps=total_steps, lr_end=min_lr)
    raise ValueError(f"Scheduler {config.TRAIN.SCHEDULER} not supported")
