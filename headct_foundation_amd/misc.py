"""Host glue of the MAE path (subset of src/utils/misc.py): checkpoint save/resume, meters, loss all-reduce,
process-group init.  No arithmetic of the hot path lives here."""
from __future__ import annotations

import os
from collections import defaultdict, deque

import torch
import torch.distributed as dist


def save_checkpoint(model, momentum_model, epoch, optimizer, scheduler, filename="model.pt", best_loss=0, dir_add=None,
                    logger=None):
    """Same file dict as the reference (misc.py:35-52): {epoch, best_loss, state_dict, momentum_model_state_dict,
    optimizer, scheduler}; `state_dict` is taken from the (DDP-wrapped) model, so keys carry `module.`."""
    save_dict = {"epoch": epoch, "best_loss": best_loss, "state_dict": model.state_dict(),
                 "momentum_model_state_dict": momentum_model.state_dict() if momentum_model is not None else None,
                 "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict()}
    os.makedirs(dir_add, exist_ok=True)
    filename = os.path.join(dir_add, filename)
    torch.save(save_dict, filename)
    if logger is not None:
        logger.info(f"Saving checkpoint {filename}")


def load_optimizer(optimizer, scheduler, loaded_state_dict, logger=None):
    """misc.py:55-69: restore optimizer / scheduler state, start_epoch = saved epoch."""
    epoch = 0
    if 'optimizer' in loaded_state_dict.keys():
        msg = optimizer.load_state_dict(loaded_state_dict['optimizer'])
        if logger:
            logger.info(f"Loaded optimizer state: {msg}")
    if 'scheduler' in loaded_state_dict.keys():
        msg = scheduler.load_state_dict(loaded_state_dict['scheduler'])
        if logger:
            logger.info(f"Loaded scheduler state: {msg}")
    if 'epoch' in loaded_state_dict.keys():
        epoch = loaded_state_dict['epoch']
        if logger:
            logger.info(f"Loaded epoch: {epoch}")
    return optimizer, scheduler, epoch


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def _reduce_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class SmoothedValue(object):
    """Window-20 median + global average (misc.py:140-196)."""

    def __init__(self, window_size=20, fmt=None):
        if fmt is None:
            fmt = "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=_reduce_device())
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count = int(t[0])
        self.total = t[1]

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t", logger=None):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter
        self.logger = logger

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(name, str(meter)) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter


def all_reduce_mean(x):
    """misc.py:287-299: mean of a scalar over ranks, returned as a Python float (the tensor itself if world == 1)."""
    world_size = dist.get_world_size() if is_dist_avail_and_initialized() else 1
    if world_size > 1:
        x_reduce = torch.as_tensor(x).detach().clone().to(_reduce_device(), dtype=torch.float32)
        dist.all_reduce(x_reduce)
        x_reduce /= world_size
        return x_reduce.item()
    return x


def init_distributed_mode(args=None):
    """misc.py:325-332 without the vestigial fairscale init: one process per GPU, env:// rendezvous from torchrun;
    backend "nccl" is RCCL on ROCm, "gloo" when no GPU is present (the CPU plumbing run of BASELINE config #1)."""
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")


def cleanup():
    if is_dist_avail_and_initialized():
        dist.destroy_process_group()
