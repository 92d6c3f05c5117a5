"""MAE pre-training entry point: same CLI flags and flow as the reference's main_pretrain_mae.py (:33-76 flags, :79-197
main, :199-239 seeding / launch), on the HIP hot path.

  torchrun --nnodes 1 --nproc_per_node N --master-addr 127.0.0.1 main_pretrain_mae.py --local_rank 0 \
      --model_name mae --batch_size 256 --max_epochs 400 --base_lr 1.5e-4 --cfg configs/mae/mae_HeadCT.yaml \
      --optimizer AdamW --scheduler cosine --weight_decay 5e-3 --grad_clip 3.0
"""
import argparse
import json
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between ranks (before HIP initialises)

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from config import get_config
from engine_pretrain_mae import tester, trainer
from headct_foundation_amd import MaskedAutoencoderViT, interpolate_pos_embed
from headct_foundation_amd.data import get_pretrain_dataloaders
from headct_foundation_amd.ddp import DistributedDataParallel
from headct_foundation_amd.lr_sched import get_lr_scheduler
from headct_foundation_amd.misc import cleanup, init_distributed_mode, load_optimizer
from headct_foundation_amd.optim import get_optimizer
from logger import create_logger


def parse_option():
    parser = argparse.ArgumentParser('HIP MAE training and evaluation script', add_help=False)
    parser.add_argument('--cfg', type=str, required=True, metavar="FILE", help='path to config file')
    parser.add_argument("--opts", help="Modify config options using the command-line", default=None, nargs='+')
    # distributed training
    parser.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)), help='local rank')
    parser.add_argument('--dist-backend', default='nccl', help='parsed and ignored, like the reference')
    parser.add_argument('--dist-url', default='env://', help='parsed and ignored, like the reference')
    parser.add_argument("--seed", type=int, help='seed')
    parser.add_argument("--use_amp", action='store_true')
    # wandb configs
    parser.add_argument("--use_wandb", action='store_true')
    parser.add_argument("--wandb_project", type=str, default="monai-test")
    # model parameters
    parser.add_argument("--model_name", type=str, help='model name')
    parser.add_argument("--model_load_path", type=str, help='path to trained model')
    parser.add_argument("--optimizer", type=str, help='training optimizer')
    parser.add_argument("--scheduler", type=str, help='learning rate scheduler')
    parser.add_argument("--base_lr", type=float, help='base learning rate')
    parser.add_argument("--min_lr", type=float, help='minimum learning rate')
    parser.add_argument("--weight_decay", type=float, help='weight decay')
    parser.add_argument("--grad_clip", type=float, help='gradient clipping')
    parser.add_argument("--batch_size", type=int, help='batch size')
    parser.add_argument("--num_workers", type=int, help='number of workers for dataloader')
    parser.add_argument("--max_epochs", type=int, help='max epoch')
    # dataset parameters
    parser.add_argument('--train_csv_path', type=str, help='path to train csv file')
    parser.add_argument('--val_csv_path', type=str, help='path to val csv file')
    parser.add_argument('--test_csv_path', type=str, help='path to test csv file')
    args, unparsed = parser.parse_known_args()
    config = get_config(args)
    return args, config


def build_model(config, device):
    """MaskedAutoencoderViT from the MAE.* block of the config (every constructor argument of mae.py:22-42 has a key)."""
    mae = config.MAE
    if mae.NORM_LAYER != 'layernorm':
        raise ValueError("MAE.NORM_LAYER must be 'layernorm' on the HIP path (RMSNorm is outside the hot path)")
    kwargs = {key.lower(): getattr(mae, key) for key in (
        "INPUT_SIZE", "PATCH_SIZE", "MASK_RATIO", "IN_CHANS", "DROPOUT_RATE", "SPATIAL_DIMS", "PATCH_EMBED", "POS_EMBED",
        "ENCODER_DEPTH", "ENCODER_EMBED_DIM", "ENCODER_MLP_DIM", "ENCODER_NUM_HEADS", "DECODER_DEPTH", "DECODER_EMBED_DIM",
        "DECODER_MLP_DIM", "DECODER_NUM_HEADS", "NORM_PIX_LOSS", "USE_BIAS")}
    return MaskedAutoencoderViT(norm_layer=nn.LayerNorm, compute_dtype=mae.COMPUTE_DTYPE, **kwargs).to(device)


def load_pretrained(config, model, logger):
    """MODEL.PRETRAINED -> the checkpoint dict (or None).  Weights go into `model` non-strictly after the `module.` prefix
    is dropped; a learnable position table saved at another resolution is resized first (main_pretrain_mae.py:125-134)."""
    path = config.MODEL.PRETRAINED
    if not path:
        return None
    # the file holds tensors and python scalars only, so the restricted unpickler is enough
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    weights = {name.replace("module.", ""): t for name, t in ckpt['state_dict'].items()}
    interpolate_pos_embed(model, weights)
    have = model.state_dict()["decoder_pos_embed"].shape
    want = weights.get("decoder_pos_embed")
    if want is not None and want.shape != have:
        # only the encoder table is resized (as in the reference, whose load_state_dict then raises on this tensor)
        raise SystemExit(f"size mismatch for decoder_pos_embed: checkpoint {tuple(want.shape)} vs model {tuple(have)}")
    report = model.load_state_dict(weights, strict=False)
    logger.info(f"Load Pretrained Model: {report} for Architecture: {config.MODEL.NAME}")
    return ckpt


def scale_learning_rate(config, world_size, steps_per_epoch, logger):
    """Linear LR scaling rule of the recipe (main_pretrain_mae.py:141-154): BASE_LR *= global batch / 256, MIN_LR =
    BASE_LR / 1000; returns (warm-up steps, total steps)."""
    global_batch = config.DATA.BATCH_SIZE * world_size
    total = steps_per_epoch * config.TRAIN.MAX_EPOCHS
    warmup = int(config.TRAIN.PER_WARMUP * total)
    config.defrost()
    config.TRAIN.BASE_LR = config.TRAIN.BASE_LR * global_batch / 256
    config.TRAIN.MIN_LR = config.TRAIN.BASE_LR * 1e-3
    config.freeze()
    logger.info(f"Effective Learning Rate: {config.TRAIN.BASE_LR}, Effective Batch Size: {global_batch}, Max Epochs: {config.TRAIN.MAX_EPOCHS}")
    logger.info(f"Number of Warmup Steps: {warmup}, Total Steps: {total}")
    return warmup, total


def main(config, wandb_run, logger):
    if config.MODEL.NAME != "mae":
        raise ValueError(f"Model {config.MODEL.NAME} not supported")
    if not torch.cuda.is_available():
        raise SystemExit("main_pretrain_mae.py (HIP) needs an MI355X: the MAE hot path has no CPU fallback")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device())
    train_loader, val_loader, test_loader = get_pretrain_dataloaders(config, device, rank, world)

    bare = build_model(config, device)
    ckpt = load_pretrained(config, bare, logger)
    # one replica per GPU; gradients are averaged over RCCL in buckets while the backward is still running
    model = DistributedDataParallel(bare, device_ids=[device], broadcast_buffers=False, find_unused_parameters=True)

    warmup, total = scale_learning_rate(config, world, len(train_loader), logger)
    optimizer = get_optimizer(config, config.TRAIN.BASE_LR, [model])
    scheduler = get_lr_scheduler(config, optimizer, warmup, total, config.TRAIN.MIN_LR)
    first_epoch = 0
    if ckpt is not None:
        optimizer, scheduler, first_epoch = load_optimizer(optimizer, scheduler, ckpt, logger)

    best = trainer(config=config, model=model, train_loader=train_loader, val_loader=val_loader, optimizer=optimizer,
                   scheduler=scheduler, start_epoch=first_epoch, max_epochs=config.TRAIN.MAX_EPOCHS,
                   val_every=config.TRAIN.VAL_EVERY, logger=logger, device=device, wandb_run=wandb_run)
    logger.info(f"Train completed, best train reconstruction loss: {best:.4f}")
    held_out = tester(config=config, model=model, test_loader=test_loader, logger=logger, device=device, wandb_run=wandb_run)
    logger.info(f"Test completed, best test reconstruction loss: {held_out:.4f}")
    cleanup()


def init_seed(seed):
    """Seed every generator the run draws from (torch CPU + all GPUs, numpy, random)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def start_wandb(config, logger):
    if not config.WANDB.WANDB_ENABLE or dist.get_rank() != 0:
        return None
    try:
        import wandb
    except ImportError:
        logger.info("wandb is not installed; continuing without it")
        return None
    summary = {"learning_rate": config.TRAIN.BASE_LR, "batch_size": config.DATA.BATCH_SIZE, "epochs": config.TRAIN.MAX_EPOCHS,
               "backbone": config.MODEL.NAME}
    return wandb.init(name=config.LOG.FILENAME, project=config.WANDB.PROJECT, config=summary)


if __name__ == "__main__":
    args, config = parse_option()
    init_distributed_mode(args)
    rank = dist.get_rank()
    init_seed(config.SEED + rank)  # a different mask / augmentation stream per rank (main_pretrain_mae.py:213)
    logger = create_logger(output_dir=config.LOG.OUTPUT_DIR, dist_rank=rank, name=config.LOG.FILENAME)
    if rank == 0 and config.OUTPUT:
        os.makedirs(config.OUTPUT, exist_ok=True)
        dump_path = os.path.join(config.OUTPUT, f"{config.LOG.FILENAME}.json")
        with open(dump_path, "w") as f:
            f.write(config.dump())
        logger.info(f"Full config saved to {dump_path}")
    logger.info(config.dump())
    logger.info(json.dumps(vars(args)))
    main(config, start_wandb(config, logger), logger)
