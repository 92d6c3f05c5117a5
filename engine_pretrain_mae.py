"""MAE pre-training engine on the HIP hot path.

The four public functions keep the reference's names, argument lists and return values (engine_pretrain_mae.py:
train_one_epoch :14-27 -> {'loss','lr'}, val_one_epoch :93-103 -> {'loss'}, trainer :149-162 -> best validation loss,
tester :268-275 -> test loss), the per-iteration order (zero_grad, forward, backward, per-parameter clip, optimizer step,
scheduler step, loss mean over ranks, log line) and the log wording, so shell scripts and log parsers written for the
reference keep working.  What differs underneath:

  * the model computes in bf16 with fp32 accumulation and fp32 master weights inside libheadct_hip.so; there is no fp16
    autocast region and therefore nothing to scale: `use_amp` is accepted and ignored, a GradScaler passed in is still
    honoured step by step (scale / unscale_ / step / update) so foreign callers see the behaviour they asked for;
  * gradient clipping and AdamW are single fused launches without host round trips (headct_foundation_amd/optim.py);
  * the reference synchronises the device and reads the loss back in every iteration (engine_pretrain_mae.py:73-74) because it
    logs every iteration.  Here the log line of iteration i is written while iteration i + 1 is already queued: the loss goes to
    a pinned host buffer by an asynchronous copy and is read one iteration later (`_LossTap`), so the device never idles
    waiting for the host between steps; same lines, same order, same values, the finite-loss check one iteration late.
    HCT_SYNC_LOSS=1 restores the reference's per-iteration synchronisation.
"""
import math
import os
import sys
import time
from collections import deque
from typing import Any, Dict, Iterable, Optional

import torch

from headct_foundation_amd.misc import MetricLogger, all_reduce_mean, get_rank, save_checkpoint
from headct_foundation_amd.optim import clip_gradients


def _loss_of(config, model, batch, device) -> torch.Tensor:
    if config.MODEL.NAME != 'mae':
        raise NotImplementedError(f"Unknown model: {config.MODEL.NAME}")
    return model(batch.to(device))[0]


def _as_float(loss) -> float:
    v = all_reduce_mean(loss)
    return v.item() if isinstance(v, torch.Tensor) else float(v)


def _drain() -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class _LossTap:
    """Loss values without a per-iteration device synchronisation: `push` queues a device -> pinned-host copy of the (rank-mean)
    loss behind the step that produced it, `pop` hands back the entries whose copy has completed, waiting for the oldest only
    when more than `depth` iterations are in flight (or at the end of the epoch)."""

    def __init__(self, depth: int = 1):
        self.depth, self.q = depth, deque()

    def push(self, loss: torch.Tensor, meta) -> None:
        v = all_reduce_mean(loss)
        v = v if isinstance(v, torch.Tensor) else torch.as_tensor(float(v))
        if v.is_cuda:
            host = torch.empty((), dtype=torch.float32, pin_memory=True)
            host.copy_(v.detach().float().reshape(()), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = v.detach().float().reshape(()).clone(), None
        self.q.append((host, ev, meta))

    def pop(self, flush: bool = False):
        out = []
        while self.q and (flush or len(self.q) > self.depth or self.q[0][1] is None or self.q[0][1].query()):
            host, ev, meta = self.q.popleft()
            if ev is not None:
                ev.synchronize()
            out.append((float(host), meta))
        return out


def _finish_epoch(meters: MetricLogger, logger) -> Dict[str, float]:
    meters.synchronize_between_processes()
    logger.info(f"Averaged stats: {meters}")
    return {name: m.global_avg for name, m in meters.meters.items()}


def train_one_epoch(config: Any, model: torch.nn.Module, loader: Iterable, optimizer: torch.optim.Optimizer, scheduler,
                    epoch: int, max_epoch: int, logger=None, device: Optional[torch.device] = None, use_amp: bool = False,
                    scaler=None, wandb_run: Optional[Any] = None) -> Dict[str, float]:
    model.train()
    meters = MetricLogger(delimiter="  ", logger=logger)
    clip = config.TRAIN.GRAD_CLIP
    n_iter = len(loader)
    sync_every_step = os.environ.get("HCT_SYNC_LOSS") == "1"
    tap = _LossTap(depth=0 if sync_every_step else 1)

    def report(flush: bool) -> None:
        for value, (it_, lr_) in tap.pop(flush):
            if not math.isfinite(value):
                logger.info(f"Loss is {value}, stopping training")
                sys.exit(1)
            meters.update(loss=value, lr=lr_)
            logger.info(f"Epoch {epoch+1}/{max_epoch} [{it_}/{n_iter}]  Loss: {value:.4f}")
            if wandb_run is not None and get_rank() == 0:
                wandb_run.log({'Training Loss': value, 'Training lr': lr_})

    for it, batch in enumerate(loader, start=1):
        optimizer.zero_grad()
        loss = _loss_of(config, model, batch, device)
        if scaler is None:
            loss.backward()
        else:
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
        if clip:
            clip_gradients(model, clip)
        if scaler is None:
            optimizer.step()
        else:
            scaler.step(optimizer)
            scaler.update()
        scheduler.step()
        if sync_every_step:
            _drain()
        # (the rate logged beside it is read after scheduler.step(), as the reference does)
        tap.push(loss, (it, optimizer.param_groups[0]["lr"]))
        report(flush=sync_every_step)
    report(flush=True)
    return _finish_epoch(meters, logger)


def val_one_epoch(config: Any, model: torch.nn.Module, loader: Iterable, epoch: int, max_epoch: int, logger=None,
                  device: Optional[torch.device] = None, use_amp: bool = False, scaler=None) -> Dict[str, float]:
    """Validation is the training forward without a backward: a fresh random mask per batch, as in the reference."""
    model.eval()
    meters = MetricLogger(delimiter="  ", logger=logger)
    n_iter = len(loader)
    with torch.no_grad():
        for it, batch in enumerate(loader, start=1):
            value = _as_float(_loss_of(config, model, batch, device))
            if not math.isfinite(value):
                logger.info(f"Loss is {value}, ignored")
            _drain()
            meters.update(loss=value)
            logger.info(f"Epoch {epoch+1}/{max_epoch} [{it}/{n_iter}]  Loss: {value:.4f}")
    return _finish_epoch(meters, logger)


def trainer(config: Any, model: torch.nn.Module, train_loader, val_loader, optimizer: torch.optim.Optimizer, scheduler,
            start_epoch: int = 0, max_epochs: int = 100, val_every: int = 10, logger=None,
            device: Optional[torch.device] = None, wandb_run: Optional[Any] = None) -> float:
    """Epoch loop: train, write `latest_<SAVE_NAME>` on rank 0, every `val_every` epochs (never after the very first)
    validate and write `best_<SAVE_NAME>` when the validation loss improved."""
    best = float("inf")
    save_name, ckpt_dir = config.MODEL.SAVE_NAME, config.MODEL.DIR

    def checkpoint(tag: str, epoch: int) -> None:
        if get_rank() == 0:
            save_checkpoint(model, None, epoch, optimizer, scheduler, best_loss=best, dir_add=ckpt_dir,
                            filename=f"{tag}_{save_name}", logger=logger)

    for epoch in range(start_epoch, max_epochs):
        logger.info(f"Epoch: {epoch+1}")
        t0 = time.time()
        stats = train_one_epoch(config, model, train_loader, optimizer, scheduler, epoch, max_epochs, logger=logger,
                                device=device, use_amp=config.AMP_ENABLE, scaler=None, wandb_run=wandb_run)
        logger.info(f"Final training  {epoch+1}/{max_epochs}, loss: {stats['loss']}, time {time.time() - t0}s")
        checkpoint("latest", epoch)
        if epoch == 0 or (epoch + 1) % val_every:
            continue
        t0 = time.time()
        val = val_one_epoch(config, model, val_loader, epoch, max_epochs, logger=logger, device=device,
                            use_amp=config.AMP_ENABLE, scaler=None)['loss']
        logger.info(f"Final validation {epoch+1}/{max_epochs} loss: {val}, time {time.time() - t0}s")
        if wandb_run is not None and get_rank() == 0:
            wandb_run.log({'Validation Loss': float(val)})
        if val < best:
            logger.info(f"new best ({best} --> {val}). ")
            best = val
            checkpoint("best", epoch)
    logger.info(f"Training Finished !, Best Loss: {best}")
    return best


def tester(config: Any, model: torch.nn.Module, test_loader, logger=None, device: Optional[torch.device] = None,
           wandb_run: Optional[Any] = None) -> float:
    t0 = time.time()
    loss = val_one_epoch(config, model, test_loader, 0, 1, logger=logger, device=device, use_amp=config.AMP_ENABLE,
                         scaler=None)['loss']
    logger.info(f"Final test loss: {loss}, time {time.time() - t0}s")
    if wandb_run is not None and get_rank() == 0:
        wandb_run.log({'Test Loss': loss})
    return loss
