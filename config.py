"""Configuration surface of the reference (config.py:6-273), MAE-relevant tree kept key-for-key so the reference's
yaml files and `--opts` overrides load unchanged.  Built on headct_foundation_amd.cfgnode.CfgNode (yacs subset)."""
import os

import yaml

from headct_foundation_amd.cfgnode import CfgNode as CN

_C = CN()
_C.BASE = ['']

_C.DATA = CN()
_C.DATA.BATCH_SIZE = 64
_C.DATA.BASE_PATH = '<path-to>/datasets'
_C.DATA.TRAIN_CSV_PATH = '<path-to>/datasets/train.csv'
_C.DATA.VAL_CSV_PATH = '<path-to>/datasets/val.csv'
_C.DATA.TEST_CSV_PATH = '<path-to>/datasets/test.csv'
_C.DATA.PIN_MEMORY = True
_C.DATA.NUM_WORKERS = 4
_C.DATA.CACHE_NUM = -1
_C.DATA.CACHE_RATE = 1.0
_C.DATA.CACHE_DIR = '<path-to>/cache_dir'
_C.DATA.DATASET = 'nyu'
_C.DATA.FEW_SHOTS = -1
_C.DATA.NUM_CLASSES = 2
# additions of this build (not in the reference): synthetic volumes when no dataset is mounted
_C.DATA.SYNTHETIC = False
_C.DATA.SYNTHETIC_SAMPLES = 8

_C.MODEL = CN()
_C.MODEL.NAME = 'mae'
_C.MODEL.PRETRAINED = None
_C.MODEL.DIR = '<path-to>/model_saved'
_C.MODEL.SAVE_NAME = 'debug.pt'
_C.MODEL.ROI = [96, 96, 96]
_C.MODEL.IN_CHANS = 3

_C.MAE = CN()
_C.MAE.INPUT_SIZE = 96
_C.MAE.PATCH_SIZE = 16
_C.MAE.MASK_RATIO = 0.75
_C.MAE.IN_CHANS = 3
_C.MAE.DROPOUT_RATE = 0.0
_C.MAE.PATCH_EMBED = 'conv'
_C.MAE.POS_EMBED = 'sincos'
_C.MAE.NORM_LAYER = 'layernorm'
_C.MAE.SPATIAL_DIMS = 3
_C.MAE.NORM_PIX_LOSS = False
_C.MAE.RETURN_IMAGE = False
_C.MAE.ENCODER_EMBED_DIM = 768
_C.MAE.ENCODER_DEPTH = 12
_C.MAE.ENCODER_MLP_DIM = 3072
_C.MAE.ENCODER_NUM_HEADS = 12
_C.MAE.DECODER_EMBED_DIM = 768
_C.MAE.DECODER_DEPTH = 8
_C.MAE.DECODER_MLP_DIM = 2048
_C.MAE.DECODER_NUM_HEADS = 16
_C.MAE.USE_BIAS = False
# addition of this build: arithmetic of the HIP path ('bf16' = bf16 storage + MFMA, 'fp32' = parity mode)
_C.MAE.COMPUTE_DTYPE = 'bf16'

# DINO / VIT trees are kept so the reference's yaml files still merge; those paths are out of scope here.
_C.DINO = CN()
_C.DINO.GLOBAL_CROP_SIZE = [112, 112, 112]
_C.DINO.GLOBAL_CROP_NUM = 2
_C.DINO.LOCAL_CROP_SIZE = [64, 64, 64]
_C.DINO.LOCAL_CROP_NUM = 2
_C.DINO.HEAD_N_LAYERS = 3
_C.DINO.HEAD_N_PROTOTYPES = 65536
_C.DINO.BOTTLENECK_DIM = 256
_C.DINO.HEAD_HIDDEN_DIM = 2048
_C.DINO.MOMENTUM_TEACHER = 0.994
_C.DINO.MOMENTUM_TEACHER_END = 1.0
_C.DINO.WARMUP_TEACHER_TEMP = 0.04
_C.DINO.TEACHER_TEMP = 0.07
_C.DINO.WARMUP_TEACHER_EPOCHS = 30
_C.DINO.DINO_LOSS_WEIGHT = 1.0
_C.DINO.USE_BN = True
_C.DINO.NORM_LAST_LAYER = True
_C.DINO.FREEZE_LAST_LAYER = 1

_C.VIT = CN()
_C.VIT.INPUT_SIZE = 96
_C.VIT.PATCH_SIZE = 12
_C.VIT.IN_CHANS = 3
_C.VIT.DROPOUT_RATE = 0.0
_C.VIT.PATCH_EMBED = 'conv'
_C.VIT.POS_EMBED = 'sincos'
_C.VIT.NORM_LAYER = 'layernorm'
_C.VIT.SPATIAL_DIMS = 3
_C.VIT.NUM_LAYERS = 12
_C.VIT.NUM_HEADS = 12
_C.VIT.HIDDEN_SIZE = 768
_C.VIT.MLP_DIM = 3072
_C.VIT.NUM_REGISTER_TOKENS = 0
_C.VIT.PATCHES_OVERLAP = 0.2
_C.VIT.POOLING = 'cls'
_C.VIT.CLASSIFICATION = False
_C.VIT.USE_BIAS = False

_C.TRAIN = CN()
_C.TRAIN.MAX_EPOCHS = 100
_C.TRAIN.VAL_EVERY = 10
_C.TRAIN.BASE_LR = 1.5e-3
_C.TRAIN.MIN_LR = 1.5e-7
_C.TRAIN.WEIGHT_DECAY = 0.04
_C.TRAIN.WEIGHT_DECAY_END = 0.4
_C.TRAIN.BETA1 = 0.9
_C.TRAIN.BETA2 = 0.95
_C.TRAIN.MOMENTUM = 0.9
_C.TRAIN.LOSS = 'l1'
_C.TRAIN.TEMPERATURE = 0.5
_C.TRAIN.OPTIMIZER = 'AdamW'
_C.TRAIN.SCHEDULER = 'cosine'
_C.TRAIN.PER_WARMUP = 0.05
_C.TRAIN.GRAD_CLIP = 1.0
_C.TRAIN.LOCK = False
_C.TRAIN.LORA = False
_C.TRAIN.CLASSIFIER = 'linear'
_C.TRAIN.LABEL_NAME = 'cancer'

_C.LOG = CN()
_C.LOG.OUTPUT_DIR = '<path-to>/headCT_foundation/log'
_C.LOG.FILENAME = 'headCT_foundation'

_C.WANDB = CN()
_C.WANDB.WANDB_ENABLE = False
_C.WANDB.PROJECT = 'headCT_foundation'

_C.SEED = 42
_C.AMP_ENABLE = False
_C.LOCAL_RANK = 0
_C.OUTPUT = ''
_C.TAG = 'default'
_C.PREDS_SAVE_NAME = 'None'


def _update_config_from_file(config, cfg_file):
    """yaml merge with `BASE:` inheritance (config.py:163-180)."""
    config.defrost()
    with open(cfg_file, 'r') as f:
        yaml_cfg = yaml.safe_load(f) or {}
    for cfg in yaml_cfg.setdefault('BASE', ['']):
        if cfg:
            _update_config_from_file(config, os.path.join(os.path.dirname(cfg_file), cfg))
    print(f'=> merge config from {cfg_file}')
    config.merge_from_file(cfg_file)
    config.freeze()


_ARG_TO_KEY = [  # CLI flag -> config key (config.py:199-251)
    ('preds_save_name', 'PREDS_SAVE_NAME'), ('dataset', 'DATA.DATASET'), ('batch_size', 'DATA.BATCH_SIZE'),
    ('few_shots', 'DATA.FEW_SHOTS'), ('num_workers', 'DATA.NUM_WORKERS'), ('train_csv_path', 'DATA.TRAIN_CSV_PATH'),
    ('val_csv_path', 'DATA.VAL_CSV_PATH'), ('test_csv_path', 'DATA.TEST_CSV_PATH'), ('optimizer', 'TRAIN.OPTIMIZER'),
    ('scheduler', 'TRAIN.SCHEDULER'), ('max_epochs', 'TRAIN.MAX_EPOCHS'), ('grad_clip', 'TRAIN.GRAD_CLIP'),
    ('base_lr', 'TRAIN.BASE_LR'), ('min_lr', 'TRAIN.MIN_LR'), ('weight_decay', 'TRAIN.WEIGHT_DECAY'), ('lock', 'TRAIN.LOCK'),
    ('pooling', 'VIT.POOLING'), ('seed', 'SEED'), ('use_amp', 'AMP_ENABLE'), ('use_wandb', 'WANDB.WANDB_ENABLE'),
    ('wandb_project', 'WANDB.PROJECT'), ('model_name', 'MODEL.NAME'), ('model_load_path', 'MODEL.PRETRAINED'),
    ('label_name', 'TRAIN.LABEL_NAME'), ('classifier', 'TRAIN.CLASSIFIER'), ('filename', 'LOG.FILENAME'),
]


def update_config(config, args):
    _update_config_from_file(config, args.cfg)
    config.defrost()
    if getattr(args, 'opts', None):
        config.merge_from_list(args.opts)
    for flag, key in _ARG_TO_KEY:
        val = getattr(args, flag, None)
        if val:  # same truthiness rule as the reference's _check_args (config.py:196-197)
            node = config
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = val
    config.LOCAL_RANK = args.local_rank
    config.OUTPUT = os.path.join(config.OUTPUT)
    config.freeze()


def get_config(args):
    """Clone of the defaults, updated from `args.cfg`, `args.opts` and the named CLI flags (config.py:261-273)."""
    config = _C.clone()
    update_config(config, args)
    return config
