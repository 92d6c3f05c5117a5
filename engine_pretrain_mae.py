"""MAE pre-training engine: same functions / signatures / step order as the reference's engine_pretrain_mae.py
(train_one_epoch :14-90, val_one_epoch :93-146, trainer :149-265, tester :268-314), driving the HIP hot path.

Differences that are not visible to callers:
  * the model's arithmetic is bf16 storage + MFMA with fp32 accumulation and fp32 master weights inside the HIP path, so
    `use_amp` / GradScaler are accepted and honoured as no-ops (bf16 needs no loss scaling; the reference's fp16 autocast
    :57 is replaced by the model's own compute dtype);
  * `torch.cuda.synchronize()` + loss `.item()` every step (:73-74) are kept because the reference logs every step, but
    run only when a GPU is present (the reference cannot run on CPU at all).
"""
import logging
import math
import sys
import time
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from headct_foundation_amd.misc import MetricLogger, all_reduce_mean, get_rank, save_checkpoint
from headct_foundation_amd.optim import clip_gradients


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def train_one_epoch(config: Any, model: torch.nn.Module, loader, optimizer: torch.optim.Optimizer, scheduler, epoch: int,
                    max_epoch: int, logger: Optional[logging.Logger] = None, device: Optional[torch.device] = None,
                    use_amp: bool = False, scaler=None, wandb_run: Optional[Any] = None) -> Dict[str, float]:
    model_name = config.MODEL.NAME
    model.train()
    metric_logger = MetricLogger(delimiter="  ", logger=logger)
    for idx, batch_data in enumerate(loader):
        optimizer.zero_grad()
        if model_name == 'mae':
            data = batch_data.to(device)
            loss, _, _ = model(data)
        else:
            raise NotImplementedError(f"Unknown model: {model_name}")
        if scaler is not None:
            scaler.scale(loss).backward()
            scaler.unscale_(optimizer)
        else:
            loss.backward()
        if config.TRAIN.GRAD_CLIP:
            clip_gradients(model, config.TRAIN.GRAD_CLIP)
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()
        scheduler.step()
        _sync()
        loss_value = all_reduce_mean(loss)
        if isinstance(loss_value, torch.Tensor):
            loss_value = loss_value.item()
        if not math.isfinite(loss_value):
            logger.info(f"Loss is {loss_value}, stopping training")
            sys.exit(1)
        metric_logger.update(loss=loss_value)
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        logger.info(f"Epoch {epoch+1}/{max_epoch} [{idx+1}/{len(loader)}]  Loss: {loss_value:.4f}")
        if wandb_run is not None and get_rank() == 0:
            wandb_run.log({'Training Loss': float(loss_value), 'Training lr': lr})
    metric_logger.synchronize_between_processes()
    logger.info(f"Averaged stats: {metric_logger}")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def val_one_epoch(config: Any, model: torch.nn.Module, loader, epoch: int, max_epoch: int,
                  logger: Optional[logging.Logger] = None, device: Optional[torch.device] = None, use_amp: bool = False,
                  scaler=None) -> Dict[str, float]:
    """Same forward as training, INCLUDING a freshly drawn random mask (the reference has no deterministic eval mask)."""
    model_name = config.MODEL.NAME
    model.eval()
    metric_logger = MetricLogger(delimiter="  ", logger=logger)
    with torch.no_grad():
        for idx, batch_data in enumerate(loader):
            if model_name == 'mae':
                data = batch_data.to(device)
                loss, _, _ = model(data)
            else:
                raise NotImplementedError(f"Unknown model: {model_name}")
            loss_value = all_reduce_mean(loss)
            if isinstance(loss_value, torch.Tensor):
                loss_value = loss_value.item()
            if not math.isfinite(loss_value):
                logger.info(f"Loss is {loss_value}, ignored")
            _sync()
            metric_logger.update(loss=loss_value)
            logger.info(f"Epoch {epoch+1}/{max_epoch} [{idx+1}/{len(loader)}]  Loss: {loss_value:.4f}")
    metric_logger.synchronize_between_processes()
    logger.info(f"Averaged stats: {metric_logger}")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def trainer(config: Any, model: torch.nn.Module, train_loader, val_loader, optimizer: torch.optim.Optimizer, scheduler,
            start_epoch: int = 0, max_epochs: int = 100, val_every: int = 10, logger: Optional[logging.Logger] = None,
            device: Optional[torch.device] = None, wandb_run: Optional[Any] = None) -> float:
    use_amp = config.AMP_ENABLE
    val_loss_min = float("inf")
    val_losses = []
    scaler = None  # bf16 HIP path: no loss scaling (the reference builds a GradScaler for fp16 autocast, :186)
    for epoch in range(start_epoch, max_epochs):
        logger.info(f"Epoch: {epoch+1}")
        epoch_time = time.time()
        train_stats = train_one_epoch(config, model, train_loader, optimizer, scheduler, epoch, max_epochs, logger=logger,
                                      device=device, use_amp=use_amp, scaler=scaler, wandb_run=wandb_run)
        logger.info(f"Final training  {epoch+1}/{max_epochs}, loss: {train_stats['loss']}, time {time.time() - epoch_time}s")
        if get_rank() == 0:
            save_checkpoint(model, None, epoch, optimizer, scheduler, best_loss=val_loss_min, dir_add=config.MODEL.DIR,
                            filename='latest_' + config.MODEL.SAVE_NAME, logger=logger)
        if (epoch + 1) % val_every == 0 and epoch != 0:
            epoch_time = time.time()
            val_stats = val_one_epoch(config, model, val_loader, epoch, max_epochs, logger=logger, device=device,
                                      use_amp=use_amp, scaler=scaler)
            logger.info(f"Final validation {epoch+1}/{max_epochs} loss: {val_stats['loss']}, time {time.time() - epoch_time}s")
            if wandb_run is not None and get_rank() == 0:
                wandb_run.log({'Validation Loss': float(val_stats['loss'])})
            val_losses.append(val_stats['loss'])
            if val_stats['loss'] < val_loss_min:
                logger.info(f"new best ({val_loss_min} --> {val_stats['loss']}). ")
                val_loss_min = val_stats['loss']
                if get_rank() == 0:
                    save_checkpoint(model, None, epoch, optimizer, scheduler, best_loss=val_loss_min, dir_add=config.MODEL.DIR,
                                    filename='best_' + config.MODEL.SAVE_NAME, logger=logger)
    logger.info(f"Training Finished !, Best Loss: {val_loss_min}")
    return val_loss_min


def tester(config: Any, model: torch.nn.Module, test_loader, logger: Optional[logging.Logger] = None,
           device: Optional[torch.device] = None, wandb_run: Optional[Any] = None) -> float:
    epoch_time = time.time()
    test_stats = val_one_epoch(config, model, test_loader, 0, 1, logger=logger, device=device, use_amp=config.AMP_ENABLE,
                               scaler=None)
    logger.info(f"Final test loss: {test_stats['loss']}, time {time.time() - epoch_time}s")
    if wandb_run is not None and get_rank() == 0:
        wandb_run.log({'Test Loss': test_stats['loss']})
    return test_stats['loss']
