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
import warnings

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


def main(config, wandb_run, logger):
    max_epochs = config.TRAIN.MAX_EPOCHS
    val_every = config.TRAIN.VAL_EVERY
    if config.MODEL.NAME != "mae":
        raise ValueError(f"Model {config.MODEL.NAME} not supported")
    if not torch.cuda.is_available():
        raise SystemExit("main_pretrain_mae.py (HIP) needs an MI355X: the MAE hot path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    train_loader, val_loader, test_loader = get_pretrain_dataloaders(config, device, dist.get_rank(), dist.get_world_size())

    if config.MAE.NORM_LAYER != 'layernorm':
        raise ValueError("MAE.NORM_LAYER must be 'layernorm' on the HIP path (RMSNorm is outside the hot path)")
    model = MaskedAutoencoderViT(
        input_size=config.MAE.INPUT_SIZE, patch_size=config.MAE.PATCH_SIZE, mask_ratio=config.MAE.MASK_RATIO,
        in_chans=config.MAE.IN_CHANS, dropout_rate=config.MAE.DROPOUT_RATE, spatial_dims=config.MAE.SPATIAL_DIMS,
        patch_embed=config.MAE.PATCH_EMBED, pos_embed=config.MAE.POS_EMBED, encoder_depth=config.MAE.ENCODER_DEPTH,
        encoder_embed_dim=config.MAE.ENCODER_EMBED_DIM, encoder_mlp_dim=config.MAE.ENCODER_MLP_DIM,
        encoder_num_heads=config.MAE.ENCODER_NUM_HEADS, decoder_depth=config.MAE.DECODER_DEPTH,
        decoder_embed_dim=config.MAE.DECODER_EMBED_DIM, decoder_mlp_dim=config.MAE.DECODER_MLP_DIM,
        decoder_num_heads=config.MAE.DECODER_NUM_HEADS, norm_pix_loss=config.MAE.NORM_PIX_LOSS, use_bias=config.MAE.USE_BIAS,
        norm_layer=nn.LayerNorm, compute_dtype=config.MAE.COMPUTE_DTYPE,
    ).to(device)

    loaded_state_dict = None
    if config.MODEL.PRETRAINED:
        # tensors only (weights_only=True): reference checkpoints hold plain tensors / python scalars
        loaded_state_dict = torch.load(config.MODEL.PRETRAINED, map_location=torch.device('cpu'), weights_only=True)
        new_sd = {k.replace("module.", ""): v for k, v in loaded_state_dict['state_dict'].items()}
        # resuming at another resolution: resize the learnable position table (main_pretrain_mae.py:132)
        interpolate_pos_embed(model, new_sd)
        own = model.state_dict()
        if "decoder_pos_embed" in new_sd and new_sd["decoder_pos_embed"].shape != own["decoder_pos_embed"].shape:
            # the reference stops here as well: interpolate_pos_embed only treats the encoder table, and load_state_dict
            # raises on a size mismatch even with strict=False
            raise SystemExit(f"size mismatch for decoder_pos_embed: checkpoint {tuple(new_sd['decoder_pos_embed'].shape)} vs "
                             f"model {tuple(own['decoder_pos_embed'].shape)}")
        msg = model.load_state_dict(new_sd, strict=False)
        logger.info(f"Load Pretrained Model: {msg} for Architecture: {config.MODEL.NAME}")

    model = DistributedDataParallel(model, device_ids=[device], broadcast_buffers=False, find_unused_parameters=True)

    world_size = dist.get_world_size()
    effective_batch_size = config.DATA.BATCH_SIZE * world_size
    total_steps = len(train_loader) * config.TRAIN.MAX_EPOCHS
    num_warmup_steps = int(config.TRAIN.PER_WARMUP * total_steps)
    config.defrost()
    config.TRAIN.BASE_LR = config.TRAIN.BASE_LR * effective_batch_size / 256  # main_pretrain_mae.py:149-151
    config.TRAIN.MIN_LR = config.TRAIN.BASE_LR * 1e-3
    config.freeze()
    logger.info(f"Effective Learning Rate: {config.TRAIN.BASE_LR}, Effective Batch Size: {effective_batch_size}, Max Epochs: {config.TRAIN.MAX_EPOCHS}")
    logger.info(f"Number of Warmup Steps: {num_warmup_steps}, Total Steps: {total_steps}")

    optimizer = get_optimizer(config, config.TRAIN.BASE_LR, [model])
    scheduler = get_lr_scheduler(config, optimizer, num_warmup_steps, total_steps, config.TRAIN.MIN_LR)
    start_epoch = 0
    if loaded_state_dict is not None:
        optimizer, scheduler, start_epoch = load_optimizer(optimizer, scheduler, loaded_state_dict, logger)

    train_loss = trainer(config=config, model=model, train_loader=train_loader, val_loader=val_loader, optimizer=optimizer,
                         scheduler=scheduler, start_epoch=start_epoch, max_epochs=max_epochs, val_every=val_every, logger=logger,
                         device=device, wandb_run=wandb_run)
    logger.info(f"Train completed, best train reconstruction loss: {train_loss:.4f}")
    test_loss = tester(config=config, model=model, test_loader=test_loader, logger=logger, device=device, wandb_run=wandb_run)
    logger.info(f"Test completed, best test reconstruction loss: {test_loss:.4f}")
    cleanup()


def init_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


if __name__ == "__main__":
    warnings.filterwarnings("ignore", message="You are using `torch.load` with `weights_only=False`")
    args, config = parse_option()
    init_distributed_mode(args)
    seed = config.SEED + dist.get_rank()  # each rank draws different masks (main_pretrain_mae.py:213)
    init_seed(seed)
    logger = create_logger(output_dir=config.LOG.OUTPUT_DIR, dist_rank=dist.get_rank(), name=config.LOG.FILENAME)
    if dist.get_rank() == 0 and config.OUTPUT:
        os.makedirs(config.OUTPUT, exist_ok=True)
        path = os.path.join(config.OUTPUT, f"{config.LOG.FILENAME}.json")
        with open(path, "w") as f:
            f.write(config.dump())
        logger.info(f"Full config saved to {path}")
    logger.info(config.dump())
    logger.info(json.dumps(vars(args)))
    wandb_run = None
    if config.WANDB.WANDB_ENABLE and dist.get_rank() == 0:
        try:
            import wandb
            wandb_run = wandb.init(name=config.LOG.FILENAME, project=config.WANDB.PROJECT,
                                   config={"learning_rate": config.TRAIN.BASE_LR, "batch_size": config.DATA.BATCH_SIZE,
                                           "epochs": config.TRAIN.MAX_EPOCHS, "backbone": config.MODEL.NAME})
        except ImportError:
            logger.info("wandb is not installed; continuing without it")
    main(config, wandb_run, logger)
