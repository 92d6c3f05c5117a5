"""DINO pre-training engine on the HIP path (BASELINE config #5).

Public functions keep the reference's names, argument lists and return values (engine_pretrain_dino.py: train_one_epoch :14-31
-> {'loss','lr','wd'}, val_one_epoch :132-145 -> {'loss'}, trainer :208-225 -> best validation loss, tester :345-354 -> test
loss) and the order of an iteration: weight decay of the first parameter group from the schedule, zero_grad, teacher forward on
the two global crops, student forward on all crops, DINO loss (which also moves the centre), backward, optional per-parameter
clip, last-layer gradients cancelled while epoch < DINO.FREEZE_LAST_LAYER, optimizer step, LR step, momentum-teacher update with
`momentum_scheduler[idx]` (the reference indexes it with the iteration INSIDE the epoch, :104), loss mean over ranks, log line.

Underneath: backbone and head compute in bf16 storage + MFMA with fp32 master weights (no fp16 autocast, nothing to scale:
`use_amp` / `scaler` are accepted and ignored), loss / clip / AdamW / EMA are fused launches on flat buffers.
"""
import math
import sys
import time
from typing import Any, Dict, Iterable, Optional

import torch

from headct_foundation_amd.dino import update_momentum_encoder
from headct_foundation_amd.misc import MetricLogger, all_reduce_mean, get_rank, save_checkpoint
from headct_foundation_amd.optim import clip_gradients


def _unwrapped(model):
    return model.module if hasattr(model, "module") else model


def cancel_gradients_last_layer(epoch: int, model, freeze_last_layer: int) -> None:
    """While epoch < freeze_last_layer the prototype layer is not trained: its gradients are dropped before the step."""
    if epoch >= freeze_last_layer:
        return
    for name, p in model.named_parameters():
        if "last_layer" in name:
            p.grad = None


def _forward_loss(model, momentum_model, crops, criterion, epoch, device):
    images = [im.to(device, non_blocking=True) for im in crops]
    with torch.no_grad():
        teacher = momentum_model(images[:2])['dino_output']
    student = model(images)['dino_output']
    return criterion(student.float(), teacher.float(), epoch)


def _scalar(loss) -> float:
    v = all_reduce_mean(loss)
    return v.item() if isinstance(v, torch.Tensor) else float(v)


def train_one_epoch(config: Any, model, loader: Iterable, optimizer, lr_scheduler, wd_scheduler, momentum_scheduler, epoch: int,
                    max_epoch: int, dino_criterion, momentum_model=None, logger=None, device: Optional[torch.device] = None,
                    use_amp: bool = False, scaler=None, wandb_run: Optional[Any] = None) -> Dict[str, float]:
    model.train()
    momentum_model.train()
    meters = MetricLogger(delimiter="  ", logger=logger)
    n_iter = len(loader)
    student, teacher = _unwrapped(model), _unwrapped(momentum_model)
    for idx, crops in enumerate(loader):
        optimizer.param_groups[0]["weight_decay"] = float(wd_scheduler[n_iter * epoch + idx])  # only the first group is regularised
        # (a Python float: a numpy scalar here would end up pickled inside the checkpoint's optimizer state)
        optimizer.zero_grad()
        loss = _forward_loss(model, momentum_model, crops, dino_criterion, epoch, device)
        loss.backward()
        if hasattr(model, "reduce_head_gradients"):
            model.reduce_head_gradients()  # data parallel: the head's flat gradient (the backbone's is reduced in buckets during its backward)
        if config.TRAIN.GRAD_CLIP:
            clip_gradients(student.backbone, config.TRAIN.GRAD_CLIP)
            clip_gradients(student.head, config.TRAIN.GRAD_CLIP)
        cancel_gradients_last_layer(epoch, student, config.DINO.FREEZE_LAST_LAYER)
        optimizer.step()
        lr_scheduler.step()
        m = float(momentum_scheduler[idx])
        update_momentum_encoder(student.backbone, teacher.backbone, m)
        update_momentum_encoder(student.head, teacher.head, m)
        torch.cuda.synchronize()
        value = _scalar(loss)
        if not math.isfinite(value):
            logger.info(f"Loss is {value}, stopping training")
            sys.exit(1)
        lr, wd = optimizer.param_groups[0]["lr"], optimizer.param_groups[0]["weight_decay"]
        meters.update(loss=value, lr=lr, wd=wd)
        logger.info(f"Epoch {epoch+1}/{max_epoch} [{idx+1}/{n_iter}]  Loss: {value:.4f}")
        if wandb_run is not None and get_rank() == 0:
            wandb_run.log({'Training Loss': value, 'Training lr': lr, 'Training wd': wd})
    meters.synchronize_between_processes()
    logger.info(f"Averaged stats: {meters}")
    return {k: m_.global_avg for k, m_ in meters.meters.items()}


def val_one_epoch(config: Any, model, loader: Iterable, epoch: int, max_epoch: int, dino_criterion, momentum_model=None, logger=None,
                  device: Optional[torch.device] = None, use_amp: bool = False, scaler=None, wandb_run: Optional[Any] = None) -> Dict[str, float]:
    """The training forward without a backward (the criterion still moves its centre, as the reference's does)."""
    model.eval()
    momentum_model.eval()
    meters = MetricLogger(delimiter="  ", logger=logger)
    n_iter = len(loader)
    with torch.no_grad():
        for idx, crops in enumerate(loader):
            loss = _forward_loss(model, momentum_model, crops, dino_criterion, epoch, device)
            torch.cuda.synchronize()
            value = _scalar(loss)
            if not math.isfinite(value):
                logger.info(f"Loss is {value}, ignored")
            meters.update(loss=value)
            logger.info(f"Epoch {epoch+1}/{max_epoch} [{idx+1}/{n_iter}]  Loss: {value:.4f}")
    meters.synchronize_between_processes()
    logger.info(f"Averaged stats: {meters}")
    return {k: m_.global_avg for k, m_ in meters.meters.items()}


def trainer(config: Any, model, train_loader, val_loader, optimizer, lr_scheduler, wd_scheduler, momentum_scheduler, dino_criterion,
            start_epoch: int = 0, max_epochs: int = 100, val_every: int = 10, momentum_model=None, logger=None,
            device: Optional[torch.device] = None, wandb_run: Optional[Any] = None) -> float:
    best = float("inf")

    def checkpoint(tag: str, epoch: int) -> None:
        if get_rank() == 0:
            save_checkpoint(model, momentum_model, epoch, optimizer, scheduler=lr_scheduler, filename=f"{tag}_{config.MODEL.SAVE_NAME}",
                            best_loss=best, dir_add=config.MODEL.DIR, logger=logger)

    for epoch in range(start_epoch, max_epochs):
        logger.info(f"Epoch: {epoch+1}")
        t0 = time.time()
        stats = train_one_epoch(config, model, train_loader, optimizer, lr_scheduler, wd_scheduler, momentum_scheduler, epoch, max_epochs,
                                dino_criterion, momentum_model=momentum_model, logger=logger, device=device, use_amp=config.AMP_ENABLE,
                                scaler=None, wandb_run=wandb_run)
        logger.info(f"Final training  {epoch+1}/{max_epochs}, loss: {stats['loss']}, time {time.time() - t0}s")
        checkpoint("last", epoch)
        if epoch == 0 or (epoch + 1) % val_every:
            continue
        t0 = time.time()
        val = val_one_epoch(config, model, val_loader, epoch, max_epochs, dino_criterion, momentum_model=momentum_model, logger=logger,
                            device=device, use_amp=config.AMP_ENABLE, scaler=None, wandb_run=wandb_run)['loss']
        logger.info(f"Final validation {epoch+1}/{max_epochs} loss: {val}, time {time.time() - t0}s")
        if wandb_run is not None and get_rank() == 0:
            wandb_run.log({'Validation Loss': float(val)})
        if val < best:
            logger.info(f"new best ({best} --> {val}). ")
            best = val
            checkpoint("best", epoch)
    logger.info(f"Training Finished !, Best Loss: {best}")
    return best


def tester(config: Any, model, test_loader, dino_criterion, momentum_model=None, logger=None, device: Optional[torch.device] = None,
           wandb_run: Optional[Any] = None) -> float:
    t0 = time.time()
    loss = val_one_epoch(config, model, test_loader, 0, 1, dino_criterion, momentum_model=momentum_model, logger=logger, device=device,
                         use_amp=config.AMP_ENABLE, scaler=None, wandb_run=wandb_run)['loss']
    logger.info(f"Final test loss: {loss}, time {time.time() - t0}s")
    if wandb_run is not None and get_rank() == 0:
        wandb_run.log({'Test Loss': loss})
    return loss
