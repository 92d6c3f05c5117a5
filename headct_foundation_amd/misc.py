"""Host-side helpers of the MAE entry points: checkpoint file, running meters, scalar reductions, process group.

Written against the behaviour the reference's callers rely on (src/utils/misc.py): the checkpoint file layout
(:35-52), what a resume restores (:55-69), the meter line "median (global average)" over a 20-value window (:140-196,
:199-236), the loss mean over ranks (:287-299) and the env:// process-group start (:325-332).  None of the hot path's
arithmetic lives here.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

CHECKPOINT_KEYS = ("epoch", "best_loss", "state_dict", "momentum_model_state_dict", "optimizer", "scheduler")


# ---- process group -------------------------------------------------------------------------------------------------
def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def _world() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def _collective_device() -> torch.device:
    # RCCL reduces device tensors, gloo host tensors
    if is_dist_avail_and_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def init_distributed_mode(args=None) -> None:
    """One process per GPU, rendezvous from the environment torchrun prepares (defaults make a bare `python main_...`
    a world of one).  Backend "nccl" is RCCL on ROCm; without a GPU the group is gloo (host-only plumbing runs)."""
    for key, default in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(key, default)
    if not torch.cuda.is_available():
        dist.init_process_group("gloo")
        return
    local = int(os.environ.get("LOCAL_RANK", getattr(args, "local_rank", 0) or 0))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))


def cleanup() -> None:
    if is_dist_avail_and_initialized():
        dist.destroy_process_group()


def all_reduce_mean(x):
    """Mean of a scalar over the ranks as a Python float; in a world of one the argument comes back untouched (the
    reference's callers then call `.item()` on it themselves)."""
    n = _world()
    if n == 1:
        return x
    buf = torch.as_tensor(x).detach().to(device=_collective_device(), dtype=torch.float32).clone()
    dist.all_reduce(buf)
    return buf.item() / n


# ---- checkpoint ----------------------------------------------------------------------------------------------------
def save_checkpoint(model, momentum_model, epoch, optimizer, scheduler, filename="model.pt", best_loss=0, dir_add=None,
                    logger=None):
    """One `torch.save` of a dict with exactly CHECKPOINT_KEYS.  `state_dict` comes from the object the caller holds, i.e.
    the data-parallel wrapper, so its keys start with `module.`; the MAE path has no momentum model (None)."""
    payload = dict.fromkeys(CHECKPOINT_KEYS)
    payload.update(epoch=epoch, best_loss=best_loss, state_dict=model.state_dict(), optimizer=optimizer.state_dict(),
                   scheduler=scheduler.state_dict())
    if momentum_model is not None:
        payload["momentum_model_state_dict"] = momentum_model.state_dict()
    os.makedirs(dir_add, exist_ok=True)
    path = os.path.join(dir_add, filename)
    torch.save(payload, path)
    if logger is not None:
        logger.info(f"Saving checkpoint {path}")


def load_optimizer(optimizer, scheduler, loaded_state_dict, logger=None):
    """Resume: optimizer and scheduler take their saved state when the file has it; training restarts at the saved
    epoch index (0 when absent).  Returns (optimizer, scheduler, start_epoch)."""
    say = logger.info if logger is not None else (lambda _msg: None)
    for key, target in (("optimizer", optimizer), ("scheduler", scheduler)):
        if key in loaded_state_dict:
            say(f"Loaded {key} state: {target.load_state_dict(loaded_state_dict[key])}")
    start = loaded_state_dict.get("epoch", 0)
    if "epoch" in loaded_state_dict:
        say(f"Loaded epoch: {start}")
    return optimizer, scheduler, start


# ---- meters --------------------------------------------------------------------------------------------------------
class SmoothedValue:
    """A scalar series: the last `window_size` values for the median / window mean, and a running sum for the global
    average.  `synchronize_between_processes` makes count and sum global (the window stays local)."""

    def __init__(self, window_size: int = 20, fmt: Optional[str] = None):
        self.window_size = window_size
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.recent = []
        self.count = 0
        self.total = 0.0

    def update(self, value, n: int = 1) -> None:
        self.recent.append(value)
        del self.recent[:-self.window_size]
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self) -> None:
        if _world() == 1:
            return
        pair = torch.tensor([float(self.count), self.total], dtype=torch.float64, device=_collective_device())
        dist.barrier()
        dist.all_reduce(pair)
        self.count, self.total = int(pair[0].item()), pair[1].item()

    @property
    def median(self) -> float:
        ordered = sorted(self.recent)
        return float(ordered[(len(ordered) - 1) // 2])  # lower median, as torch.median

    @property
    def avg(self) -> float:
        return float(sum(self.recent) / len(self.recent))

    @property
    def global_avg(self) -> float:
        return self.total / self.count

    @property
    def max(self):
        return max(self.recent)

    @property
    def value(self):
        return self.recent[-1]

    def __str__(self) -> str:
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    """Named SmoothedValues, created on first `update(name=value)`; `str()` joins "name: meter" with the delimiter."""

    def __init__(self, delimiter: str = "\t", logger=None):
        self.meters: Dict[str, SmoothedValue] = {}
        self.delimiter = delimiter
        self.logger = logger

    def update(self, **scalars) -> None:
        for name, v in scalars.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            if not isinstance(v, (int, float)):
                raise TypeError(f"meter {name}: expected a number, got {type(v).__name__}")
            self.meters.setdefault(name, SmoothedValue()).update(v)

    def add_meter(self, name: str, meter: SmoothedValue) -> None:
        self.meters[name] = meter

    def __getattr__(self, name):
        meters = self.__dict__.get("meters", {})
        if name in meters:
            return meters[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def synchronize_between_processes(self) -> None:
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def __str__(self) -> str:
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())
