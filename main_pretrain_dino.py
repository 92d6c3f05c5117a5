"""DINO pre-training entry point on the HIP path: the reference's CLI flags (main_pretrain_dino.py:36-75) and flow (:78-290).

  torchrun --nnodes 1 --nproc_per_node N --master-addr 127.0.0.1 main_pretrain_dino.py --local_rank 0 --model_name dino \
      --batch_size 8 --max_epochs 200 --base_lr 5e-4 --cfg configs/dino/dino_HeadCT.yaml --optimizer AdamW --scheduler cosine

Student and momentum teacher are MultiCropWrapper(ViTBackbone, DINOHead) pairs; the teacher starts as a copy of the student
and is only ever written by the momentum update.  Data: synthetic multi-crop batches (2 global + DINO.LOCAL_CROP_NUM local crops,
all at VIT.INPUT_SIZE as the reference resizes them) unless a real loader is plugged in.
"""
import argparse
import json
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between ranks (before HIP initialises)

import numpy as np
import torch
import torch.distributed as dist

from config import get_config
from engine_pretrain_dino import tester, trainer
from headct_foundation_amd.dino import DINOLoss, DinoDataParallel, DinoOptimizer, SyntheticCrops, get_wd_scheduler, wd_cosine_scheduler
from headct_foundation_amd.dino_model import DINOHead, MultiCropWrapper, ViTBackbone
from headct_foundation_amd.lr_sched import get_lr_scheduler
from headct_foundation_amd.misc import cleanup, init_distributed_mode
from logger import create_logger


def parse_option():
    parser = argparse.ArgumentParser('HIP DINO training and evaluation script', add_help=False)
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
    args, _ = parser.parse_known_args()
    return args, get_config(args)


def build_pair(config, device):
    """MultiCropWrapper(ViTBackbone, DINOHead) from the VIT.* / DINO.* blocks (main_pretrain_dino.py:107-172)."""
    v, d = config.VIT, config.DINO
    backbone = ViTBackbone(in_chans=v.IN_CHANS, img_size=v.INPUT_SIZE, patch_size=v.PATCH_SIZE, hidden_size=v.HIDDEN_SIZE, mlp_dim=v.MLP_DIM,
                           num_layers=v.NUM_LAYERS, num_heads=v.NUM_HEADS, patch_embed=v.PATCH_EMBED, pos_embed=v.POS_EMBED,
                           classification=v.CLASSIFICATION, num_classes=config.DATA.NUM_CLASSES, dropout_rate=v.DROPOUT_RATE,
                           spatial_dims=v.SPATIAL_DIMS, num_register_tokens=v.NUM_REGISTER_TOKENS, qkv_bias=v.USE_BIAS,
                           compute_dtype=config.MAE.COMPUTE_DTYPE)
    head = DINOHead(in_dim=v.HIDDEN_SIZE, out_dim=d.HEAD_N_PROTOTYPES, hidden_dim=d.HEAD_HIDDEN_DIM, bottleneck_dim=d.BOTTLENECK_DIM,
                    nlayers=d.HEAD_N_LAYERS, use_bn=d.USE_BN, norm_last_layer=d.NORM_LAST_LAYER, compute_dtype=config.MAE.COMPUTE_DTYPE)
    return MultiCropWrapper(backbone, head).to(device)


def load_pretrained(config, model, momentum_model, logger):
    """MODEL.PRETRAINED -> checkpoint dict (or None); `module.` / `_orig_mod.` prefixes dropped, non-strict (misc.py:72-96)."""
    if not config.MODEL.PRETRAINED:
        return None
    ckpt = torch.load(config.MODEL.PRETRAINED, map_location="cpu", weights_only=True)
    strip = lambda sd: {k.replace("module.", "").replace("_orig_mod.", ""): t for k, t in sd.items()}
    logger.info(f"Load Pretrained Model: {model.load_state_dict(strip(ckpt['state_dict']), strict=False)} for Architecture: {config.MODEL.NAME}")
    if ckpt.get('momentum_model_state_dict') is not None:
        report = momentum_model.load_state_dict(strip(ckpt['momentum_model_state_dict']), strict=False)
        logger.info(f"Load Pretrained Momentum Model: {report} for Architecture: {config.MODEL.NAME}")
    return ckpt


def main(config, wandb_run, logger):
    if config.MODEL.NAME != "dino":
        raise ValueError(f"Model {config.MODEL.NAME} not supported")
    if not torch.cuda.is_available():
        raise SystemExit("main_pretrain_dino.py (HIP) needs an MI355X: the DINO path has no CPU fallback")
    if not config.DATA.SYNTHETIC:
        raise NotImplementedError("the MONAI/NIfTI multi-crop data path is outside this build; set DATA.SYNTHETIC True")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device())
    n_crops = 2 + config.DINO.LOCAL_CROP_NUM
    bs = config.DATA.BATCH_SIZE
    nb = max(1, config.DATA.SYNTHETIC_SAMPLES // max(1, world) // bs)
    loaders = [SyntheticCrops(k, bs, n_crops, config.VIT.IN_CHANS, config.VIT.INPUT_SIZE, device, config.SEED + rank + salt)
               for k, salt in ((nb, 0), (max(1, nb // 4), 1000), (max(1, nb // 4), 2000))]
    train_loader, val_loader, test_loader = loaders

    student, teacher = build_pair(config, device), build_pair(config, device)
    teacher.load_state_dict(student.state_dict())  # the momentum teacher starts from the student's weights
    ckpt = load_pretrained(config, student, teacher, logger)
    for p in teacher.parameters():
        p.requires_grad_(False)
    model = DinoDataParallel(student, device_ids=[device])
    momentum_model = DinoDataParallel(teacher, device_ids=[device])

    total = len(train_loader) * config.TRAIN.MAX_EPOCHS
    warmup = int(config.TRAIN.PER_WARMUP * total)
    config.defrost()
    config.TRAIN.BASE_LR = config.TRAIN.BASE_LR * bs * world / 256  # linear scaling rule, main_pretrain_dino.py:209-212
    config.TRAIN.MIN_LR = config.TRAIN.BASE_LR * 1e-3
    config.freeze()
    logger.info(f"Effective Learning Rate: {config.TRAIN.BASE_LR}, Effective Batch Size: {bs * world}, Max Epochs: {config.TRAIN.MAX_EPOCHS}")
    logger.info(f"Number of Warmup Steps: {warmup}, Total Steps: {total}")
    if config.TRAIN.OPTIMIZER != 'AdamW':
        raise NotImplementedError(f"Unknown optimizer for the HIP DINO path: {config.TRAIN.OPTIMIZER}")
    optimizer = DinoOptimizer(model, lr=config.TRAIN.BASE_LR, betas=(config.TRAIN.BETA1, config.TRAIN.BETA2), weight_decay=config.TRAIN.WEIGHT_DECAY)
    lr_scheduler = get_lr_scheduler(config, optimizer.primary, warmup, total, config.TRAIN.MIN_LR)
    wd_scheduler = get_wd_scheduler(config, len(train_loader))
    momentum_scheduler = wd_cosine_scheduler(config.DINO.MOMENTUM_TEACHER, config.DINO.MOMENTUM_TEACHER_END, config.TRAIN.MAX_EPOCHS, len(train_loader))
    first_epoch = 0
    if ckpt is not None:
        if 'optimizer' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])
        if 'scheduler' in ckpt:
            lr_scheduler.load_state_dict(ckpt['scheduler'])
        first_epoch = ckpt.get('epoch', 0)
    criterion = DINOLoss(out_dim=config.DINO.HEAD_N_PROTOTYPES, ncrops=n_crops, warmup_teacher_temp=config.DINO.WARMUP_TEACHER_TEMP,
                         teacher_temp=config.DINO.TEACHER_TEMP, warmup_teacher_temp_epochs=config.DINO.WARMUP_TEACHER_EPOCHS,
                         nepochs=config.TRAIN.MAX_EPOCHS).to(device)
    best = trainer(config=config, model=model, momentum_model=momentum_model, train_loader=train_loader, val_loader=val_loader,
                   optimizer=optimizer, lr_scheduler=lr_scheduler, wd_scheduler=wd_scheduler, momentum_scheduler=momentum_scheduler,
                   dino_criterion=criterion, start_epoch=first_epoch, max_epochs=config.TRAIN.MAX_EPOCHS, val_every=config.TRAIN.VAL_EVERY,
                   logger=logger, device=device, wandb_run=wandb_run)
    logger.info(f"train completed, best train dino loss: {best:.4f}")
    held_out = tester(config=config, model=model, test_loader=test_loader, dino_criterion=criterion, momentum_model=momentum_model,
                      logger=logger, device=device, wandb_run=wandb_run)
    logger.info(f"test completed, best test dino loss: {held_out:.4f}")
    cleanup()


if __name__ == "__main__":
    args, config = parse_option()
    init_distributed_mode(args)
    rank = dist.get_rank()
    seed = config.SEED + rank
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    logger = create_logger(output_dir=config.LOG.OUTPUT_DIR, dist_rank=rank, name=config.LOG.FILENAME)
    if rank == 0 and config.OUTPUT:
        os.makedirs(config.OUTPUT, exist_ok=True)
        with open(os.path.join(config.OUTPUT, f"{config.LOG.FILENAME}.json"), "w") as f:
            f.write(config.dump())
    logger.info(config.dump())
    logger.info(json.dumps(vars(args)))
    main(config, None, logger)
