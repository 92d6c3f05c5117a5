"""CPU: the C-ABI library loads and exports every symbol include/headct_hip.h declares; host-side mirror of the
reference interface (names/shapes/state_dict, config, LR schedule, checkpoints, loud failure without a GPU)."""
import argparse
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import mae_oracle as O
from tests.util import GOLDEN, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "headct_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hct_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hct_mae_plan", "hct_gemm_args", "hct_mae_config", "hct_param_info"}
    assert len(declared) > 40
    from headct_foundation_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(raw, s)]
    assert not missing, missing
    assert set(_lib.exported_symbols()) <= declared | {"hct_debug_force_simple_attention", "hct_debug_set_gemm_variant", "hct_debug_set_gemm_stagger"}
    assert lib.hct_version() >= 100 and lib.hct_has_mfma_kernels() == 1


def test_gemm_workspace_sizes_host_side(lib):
    """hct_gemm_workspace_bytes / hct_gemm_nt_flags_offset are host arithmetic (no device call): a forward / dgrad product whose
    remainder round of tiles would be shared out by K range (K >= 512, enough stage pairs to gain) asks for the stream-K region
    (64 MiB of slabs + a 4 KiB head) behind its column-sum partials, other shapes do not, and the flags sit at a 256-byte aligned
    offset from the END of whatever workspace is passed."""
    from headct_foundation_amd._lib import HCT_BF16, GemmArgs
    sk = 256 * 262144 + 4096

    def nt(M, N, K, colsum=False):
        a = GemmArgs()
        a.M, a.N, a.K = M, N, K
        a.A, a.a_dtype, a.lda, a.transA = 256, HCT_BF16, K, 0  # (alignment probes: nothing is dereferenced)
        a.B, a.b_dtype, a.ldb, a.transB = 256, HCT_BF16, K, 1
        a.C, a.c_dtype, a.ldc = 256, HCT_BF16, N
        a.alpha = 1.0
        if colsum:
            a.colsum_out = 256
        return lib.hct_gemm_workspace_bytes(C.byref(a))

    assert lib.hct_gemm_nt_stream_k_bytes() == sk
    assert nt(55552, 768, 3072) == sk  # 651 tiles on 256 CUs (no GPU here: the library assumes 256): 139 remainder tiles, 22 pairs saved
    assert nt(1000, 768, 512) == 0 and nt(14080, 768, 3072) == 0  # too little to gain (threshold: 20 stage pairs per CU)
    assert nt(55552, 3072, 256) == 0 and nt(55552, 768, 448) == 0  # K < 512, or not a multiple of 64 on the persistent kernel
    with_cs = nt(55552, 3072, 768, colsum=True)
    assert with_cs >= 217 * 4 * 3072 * 4 and with_cs < sk  # column-sum partials only: 2604 tiles leave too short a remainder
    none = 2 ** 64 - 1
    assert lib.hct_gemm_nt_flags_offset(0) == none and lib.hct_gemm_nt_flags_offset(sk - 1) == none
    assert lib.hct_gemm_nt_flags_offset(sk) == 0
    for extra in (1, 255, 256, 1000, 12345678):
        off = lib.hct_gemm_nt_flags_offset(sk + extra)
        assert off % 256 == 0 and off <= extra and extra - off < 256


@pytest.mark.parametrize("name", ["micro", "yaml_cut", "tiny", "vitb_cut"])
def test_module_mirrors_reference_state_dict(lib, name):
    """Keys, order, shapes and dtypes of state_dict() equal the manifest dumped from the reference model."""
    from headct_foundation_amd import MaskedAutoencoderViT
    batch, seed = (2, 1) if name == "yaml_cut" else (2, 0)
    fx = load_golden(f"{name}_b{batch}_s{seed}")
    cfg = O.CONFIGS[name]
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    assert got == fx["state_dict_manifest"]
    assert [n for n, _ in m.named_parameters()] == [n for n, _, _ in O.param_shapes(cfg)]
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert frozen == ["decoder_pos_embed"]  # mae.py:92
    # load / round-trip through the flat buffer
    params = O.make_params(cfg, seed)
    m.load_state_dict(params, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, params[k]), k
    # every parameter is a view of one flat buffer laid out by the native plan
    base = m._flat.data_ptr()
    for n, p in m.named_parameters():
        assert base <= p.data_ptr() < base + m._flat.numel() * 4


def test_reference_init_statistics(lib):
    from headct_foundation_amd import MaskedAutoencoderViT, build_sincos_position_embedding
    torch.manual_seed(0)
    cfg = O.CONFIGS["tiny"]
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    sd = m.state_dict()
    assert torch.equal(sd["decoder_pos_embed"], O.build_sincos_position_embedding_3d(cfg.grid, cfg.decoder_embed_dim))
    assert torch.equal(sd["patch_embedding.position_embeddings"], build_sincos_position_embedding([cfg.grid] * 3, cfg.encoder_embed_dim))
    w = sd["blocks.0.attn.qkv.weight"]
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5  # xavier_uniform on the fused [3D, D] weight (mae.py:144)
    assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.95 * bound
    assert float(sd["blocks.3.mlp.linear1.bias"].abs().max()) == 0.0
    assert torch.equal(sd["norm.weight"], torch.ones_like(sd["norm.weight"]))
    cw = sd["patch_embedding.patch_embeddings.weight"]
    assert float(cw.abs().max()) <= 1.0 / (cfg.patch_dim ** 0.5) + 1e-7  # Conv3d default init kept (mae.py:140-148)
    assert 0.005 < float(sd["cls_token"].std()) < 0.04


def test_forward_fails_loudly_without_gpu(lib):
    from headct_foundation_amd import HctError, MaskedAutoencoderViT
    from headct_foundation_amd.optim import HipAdamW, clip_gradients
    cfg = O.CONFIGS["micro"]
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    with pytest.raises(HctError):
        m(O.make_volume(cfg, 2, 0))
    with pytest.raises(HctError):
        clip_gradients(m, 3.0)
    opt = HipAdamW(m, lr=1e-3)
    with pytest.raises(HctError):
        opt.step()


def test_lr_scheduler_resume_follows_the_new_run(lib):
    """LambdaLR saves the attributes of a callable OBJECT and restores them on load; a closure is saved as None, which is what a
    reference checkpoint holds (lr_sched.py:127-139 builds a closure): a resume with another total / warm-up / final rate then
    follows the NEW curve from the restored step count (the base rate itself is part of LambdaLR's saved state, `base_lrs`, in the
    reference as here)."""
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    mk = lambda lr: torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
    old_opt = mk(1e-3)
    old = get_cosine_schedule_with_warmup(old_opt, 5, 100, lr_end=1e-6)
    for _ in range(20):
        old_opt.step(); old.step()
    sd = old.state_dict()
    assert sd["lr_lambdas"] == [None]
    new_opt = mk(1e-3)
    new = get_cosine_schedule_with_warmup(new_opt, 10, 400, lr_end=2e-6)
    new.load_state_dict(sd)
    fresh_opt = mk(1e-3)
    fresh = get_cosine_schedule_with_warmup(fresh_opt, 10, 400, lr_end=2e-6)
    for _ in range(20):
        fresh_opt.step(); fresh.step()
    for _ in range(30):
        new_opt.step(); new.step(); fresh_opt.step(); fresh.step()
        assert abs(new_opt.param_groups[0]["lr"] - fresh_opt.param_groups[0]["lr"]) < 1e-15


def test_dino_optimizer_state_dict_has_the_reference_layout():
    """DinoOptimizer.state_dict() merges / splits the two fused optimizers' dicts at the number of backbone parameters: the result is
    what ONE torch.optim.AdamW over [backbone parameters, head parameters] carries (main_pretrain_dino.py:219), so reference DINO
    checkpoints load and ours are readable by the reference's load_optimizer (misc.py:55-69).  Host logic only: stand-ins for the
    two fused optimizers."""
    from headct_foundation_amd.dino import DinoOptimizer

    class _Half:
        def __init__(self, n, base):
            self.param_groups = [{"lr": 0.1, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.04, "params": list(range(n))}]
            self.state = {i: {"step": torch.tensor(3.0), "exp_avg": torch.full((2,), float(base + i)), "exp_avg_sq": torch.zeros(2)} for i in range(n)}

        def state_dict(self):
            return {"state": dict(self.state), "param_groups": [dict(self.param_groups[0])]}

        def load_state_dict(self, sd):
            self.loaded = sd

    opt = DinoOptimizer.__new__(DinoOptimizer)
    opt.primary, opt.secondary = _Half(5, 0), _Half(3, 100)
    sd = opt.state_dict()
    assert sorted(sd["state"]) == list(range(8)) and len(sd["param_groups"]) == 1 and sd["param_groups"][0]["params"] == list(range(8))
    assert float(sd["state"][5]["exp_avg"][0]) == 100.0 and float(sd["state"][4]["exp_avg"][0]) == 4.0
    # the same layout as torch.optim.AdamW over the concatenated parameter list
    ps = [torch.nn.Parameter(torch.zeros(2)) for _ in range(8)]
    ref = torch.optim.AdamW(ps, lr=0.1, weight_decay=0.04)
    for p in ps:
        p.grad = torch.ones(2)
    ref.step()
    rsd = ref.state_dict()
    assert sorted(rsd["state"]) == sorted(sd["state"]) and rsd["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    assert set(rsd["state"][0]) == set(sd["state"][0])
    opt.load_state_dict(rsd)  # a reference-style dict splits at 5
    assert sorted(opt.primary.loaded["state"]) == list(range(5)) and sorted(opt.secondary.loaded["state"]) == list(range(3))
    assert opt.primary.loaded["param_groups"][0]["params"] == list(range(5)) and opt.secondary.loaded["param_groups"][0]["params"] == list(range(3))
    opt.load_state_dict({"backbone": {"x": 1}, "head": {"y": 2}})  # the split form of older checkpoints still loads
    assert opt.primary.loaded == {"x": 1} and opt.secondary.loaded == {"y": 2}
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"params": list(range(7))}]})


def test_lr_scheduler_matches_reference_values(lib):
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    with open(os.path.join(GOLDEN, "lr_schedule.json")) as f:
        fx = json.load(f)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=fx["base_lr"])
    sched = get_cosine_schedule_with_warmup(opt, fx["warmup"], fx["total"], lr_end=fx["min_lr"])
    vals = []
    for _ in range(len(fx["lrs"])):
        vals.append(opt.param_groups[0]["lr"])
        opt.step(); sched.step()
    assert np.allclose(vals, fx["lrs"], rtol=1e-12)
    with pytest.raises(ValueError):
        get_cosine_schedule_with_warmup(opt, 1, 10, lr_end=1.0)


def test_config_surface(tmp_path):
    import config as cfgmod
    a = argparse.Namespace(cfg=os.path.join(ROOT, "configs/mae/mae_tiny_plumbing.yaml"),
                           opts=["TRAIN.GRAD_CLIP", "3.0", "MODEL.PRETRAINED", "None", "MAE.MASK_RATIO", "0.5"],
                           local_rank=3, batch_size=4, model_name="mae", base_lr=1e-3, use_amp=False, seed=7)
    c = cfgmod.get_config(a)
    assert c.MAE.ENCODER_EMBED_DIM == 192 and c.MAE.DECODER_DEPTH == 2  # BASE inheritance + override
    assert c.DATA.BATCH_SIZE == 4 and c.TRAIN.BASE_LR == 1e-3 and c.SEED == 7 and c.LOCAL_RANK == 3
    assert c.TRAIN.GRAD_CLIP == 3.0 and c.MODEL.PRETRAINED is None and c.MAE.MASK_RATIO == 0.5
    assert c.is_frozen()
    with pytest.raises(AttributeError):
        c.TRAIN.BASE_LR = 1.0
    c.defrost(); c.TRAIN.BASE_LR = 2.0; c.freeze()
    assert "BASE_LR: 2.0" in c.dump()
    bad = argparse.Namespace(cfg=a.cfg, opts=["NOPE.KEY", "1"], local_rank=0)
    with pytest.raises(KeyError):
        cfgmod.get_config(bad)
    # the reference's own yaml keys merge unchanged (same tree)
    ref_like = tmp_path / "ref.yaml"
    ref_like.write_text("MODEL:\n  NAME: vit\n  PRETRAINED: None\nMAE:\n  PATCH_SIZE: 12\n  IN_CHANS: 3\n  USE_BIAS: True\nTRAIN:\n  LOSS: L1\n")
    c2 = cfgmod.get_config(argparse.Namespace(cfg=str(ref_like), opts=None, local_rank=0, model_name="mae"))
    assert c2.MODEL.NAME == "mae" and c2.MAE.PATCH_SIZE == 12 and c2.MODEL.PRETRAINED is None


def test_optimizer_state_dict_is_torch_adamw_compatible(lib):
    """HipAdamW.state_dict() has torch.optim.AdamW's layout (golden: keys dumped from the reference run)."""
    from headct_foundation_amd import MaskedAutoencoderViT
    from headct_foundation_amd.optim import HipAdamW
    fx = load_golden("micro_b2_s0")
    cfg = O.CONFIGS["micro"]
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    opt = HipAdamW(m, lr=1e-3, weight_decay=5e-3, betas=(0.9, 0.95))
    sd = opt.state_dict()
    assert sorted(sd.keys()) == fx["train"]["opt_state_keys"] == ["param_groups", "state"]
    ref = torch.optim.AdamW(list(m.parameters()), lr=1e-3, weight_decay=5e-3, betas=(0.9, 0.95))
    g, gr = sd["param_groups"][0], ref.state_dict()["param_groups"][0]
    assert g["params"] == gr["params"] and g["lr"] == gr["lr"] and g["betas"] == gr["betas"] and g["weight_decay"] == gr["weight_decay"]
    assert g["eps"] == gr["eps"] == 1e-8


@pytest.mark.parametrize("input_size,patch,ratio", [(80, 8, 0.6), (80, 8, 0.9), (40, 8, 0.6), (96, 16, 0.75), (64, 16, 0.5),
                                                    (80, 8, 0.3), (80, 8, 0.7), (48, 8, 0.9)])
def test_plan_keeps_as_many_patches_as_python_does(lib, input_size, patch, ratio):
    """int(L * (1 - mask_ratio)) is evaluated in double on both sides of the C ABI (a float32 ratio gives 100 instead of
    99 kept patches at L=1000, ratio 0.9, and 399 instead of 400 at 0.6)."""
    from headct_foundation_amd import MaskedAutoencoderViT
    m = MaskedAutoencoderViT(input_size=input_size, patch_size=patch, mask_ratio=ratio, encoder_depth=1, encoder_embed_dim=48,
                             encoder_mlp_dim=96, encoder_num_heads=2, decoder_depth=1, decoder_embed_dim=48, decoder_mlp_dim=96,
                             decoder_num_heads=2, pos_embed="sincos")
    L = (input_size // patch) ** 3
    assert m.len_keep == int(L * (1 - ratio))
    h = lib.hct_mae_plan_create(C.byref(m._ccfg), 2, m._dt)
    try:
        assert lib.hct_mae_plan_len_keep(h) == m.len_keep
    finally:
        lib.hct_mae_plan_destroy(h)
