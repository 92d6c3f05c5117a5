"""Same-process A/B of the full training step (bench.py's workload) under a debug hook of the library.

Between boxes the step time moves by +-0.5 ms, on one box by +-0.03 ms: small kernel changes are judged here, alternating
blocks of steps with the hook off / on.   usage: python scripts/ab_step.py stagger0 | w4auto | <none>
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from headct_foundation_amd import MaskedAutoencoderViT, _lib
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients

if os.environ.get('HCT_LIB_TAG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ['HCT_LIB_TAG']}.so")
lib = _lib.load()
HOOKS = {
    "stagger0": (lambda: lib.hct_debug_set_gemm_stagger(0), lambda: lib.hct_debug_set_gemm_stagger(-1)),
    "w4auto": (lambda: lib.hct_debug_set_gemm_variant(-4), lambda: lib.hct_debug_set_gemm_variant(-5)),
    "tile256": (lambda: lib.hct_debug_set_gemm_variant(256), lambda: lib.hct_debug_set_gemm_variant(0)),  # on = 256-row tiles only
    "attn_online": (lambda: lib.hct_debug_force_simple_attention(2), lambda: lib.hct_debug_force_simple_attention(3)),  # on = online-softmax forward
    "attn_bwd1": (lambda: lib.hct_debug_force_simple_attention(14), lambda: lib.hct_debug_force_simple_attention(10)),  # on = single-phase backward
    "attn_bwd4w": (lambda: lib.hct_debug_force_simple_attention(18), lambda: lib.hct_debug_force_simple_attention(10)),  # on = 4-wave two-phase backward
    "fusedfold": (lambda: lib.hct_debug_set_gemm_variant(-7), lambda: lib.hct_debug_set_gemm_variant(-6)),  # on = wgrad splits folded inside the launch
    "bwd3enc": (lambda: lib.hct_debug_force_simple_attention(101206), lambda: lib.hct_debug_force_simple_attention(100020)),  # on = key-owner backward for the encoder
    "bwd3k2": (lambda: lib.hct_debug_force_simple_attention(100022), lambda: lib.hct_debug_force_simple_attention(101206)),  # on = encoder backward as two waves x two key tiles
    "bwd3off": (lambda: lib.hct_debug_force_simple_attention(42), lambda: lib.hct_debug_force_simple_attention(10)),  # on = two-phase backward everywhere
    "bwd4off": (lambda: lib.hct_debug_force_simple_attention(100002), lambda: lib.hct_debug_force_simple_attention(101206)),  # on = two-phase backward for the decoder instead of the persistent key-owner kernel
    "bwd4k2": (lambda: lib.hct_debug_force_simple_attention(100038), lambda: lib.hct_debug_force_simple_attention(101206)),  # on = bwd4 with 8 waves x two key tiles
    "fwd4": (lambda: lib.hct_debug_force_simple_attention(100062), lambda: lib.hct_debug_force_simple_attention(101206)),  # on = persistent forward for the decoder
    "stagger2": (lambda: lib.hct_debug_set_gemm_stagger(2), lambda: lib.hct_debug_set_gemm_stagger(-1)),
    "stagger4": (lambda: lib.hct_debug_set_gemm_stagger(4), lambda: lib.hct_debug_set_gemm_stagger(-1)),
    "skoff": (lambda: lib.hct_debug_set_gemm_variant(-1000 - (1 << 24)), lambda: lib.hct_debug_set_gemm_variant(-1000 - 512)),  # on = whole tiles only (no stream-K remainder round)
    "sk1536": (lambda: lib.hct_debug_set_gemm_variant(-1000 - 1536), lambda: lib.hct_debug_set_gemm_variant(-1000 - 512)),  # on = stream-K only for K >= 1536
    "skgain8": (lambda: lib.hct_debug_set_gemm_variant(-100 - 8), lambda: lib.hct_debug_set_gemm_variant(-100 - 20)),  # on = stream-K where it saves >= 8 pairs per CU (default 20)
    # on = stream-K where it saves >= 20 / 16 stage pairs per CU, off = whole tiles only
    "sk20": (lambda: (lib.hct_debug_set_gemm_variant(-1000 - 512), lib.hct_debug_set_gemm_variant(-100 - 20)), lambda: lib.hct_debug_set_gemm_variant(-1000 - (1 << 24))),
    "sk16": (lambda: (lib.hct_debug_set_gemm_variant(-1000 - 512), lib.hct_debug_set_gemm_variant(-100 - 16)), lambda: lib.hct_debug_set_gemm_variant(-1000 - (1 << 24))),
    # on = weight gradients queued and run in grouped launches (the default), off = every weight gradient inside its stage (split-K + fold)
    "wgdefer": (lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 1, 0) for pl in model._plans.values()],
                lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 0, 0) for pl in model._plans.values()]),
    "wggroup4": (lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 1, 4) for pl in model._plans.values()],
                 lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 1, 0) for pl in model._plans.values()]),
    # on = the module default (decoder tail on the masked patches' rows only), off = every row through the whole decoder
    "tail": (lambda: setattr(model, "full_pred", False), lambda: setattr(model, "full_pred", True)),
    # on = the persistent GEMM grids of the BACKWARD leave 16 CUs free (what the data-parallel wrapper does while gradient buckets are
    # in flight, ddp.py): its cost at world size 1 is the floor of the scaling loss
    "reserve16": (lambda: globals().__setitem__("BWD_RESERVE", 16), lambda: globals().__setitem__("BWD_RESERVE", 0)),
    "wggroup6": (lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 1, 6) for pl in model._plans.values()],
                 lambda: [lib.hct_mae_plan_set_wgrad_defer(pl.handle, 1, 0) for pl in model._plans.values()]),
    # on = two workgroups per CU (256 x 128 tiles) for the shapes with less than one round of 256 x 256 tiles (the encoder's 165)
    "w4small": (lambda: lib.hct_debug_set_gemm_variant(-10), lambda: lib.hct_debug_set_gemm_variant(-11)),
    # on = the first decoder block's LayerNorm1 / qkv on kept rows + one row per patch position (the default), off = on every row
    "dec0": (lambda: [lib.hct_mae_plan_set_dec0(pl.handle, 1) for pl in model._plans.values()],
             lambda: [lib.hct_mae_plan_set_dec0(pl.handle, 0) for pl in model._plans.values()]),
    "evenrounds": (lambda: lib.hct_debug_set_gemm_variant(-12), lambda: lib.hct_debug_set_gemm_variant(-13)),
    "skgain12": (lambda: lib.hct_debug_set_gemm_variant(-100 - 12), lambda: lib.hct_debug_set_gemm_variant(-100 - 20)),
    "skgain16": (lambda: lib.hct_debug_set_gemm_variant(-100 - 16), lambda: lib.hct_debug_set_gemm_variant(-100 - 20)),
    # on = 192-row tiles for the plain / +residual shapes with less than one round of 256-row tiles (the encoder's N = 768 products; the default)
    "mt3": (lambda: lib.hct_debug_set_gemm_variant(-14), lambda: lib.hct_debug_set_gemm_variant(-15)),
    "none": (lambda: None, lambda: None),
}
BWD_RESERVE = 0
name = sys.argv[1] if len(sys.argv) > 1 else "none"
on, off = HOOKS[name]
dev = torch.device("cuda", 0)
torch.manual_seed(42)
B = 256
model = MaskedAutoencoderViT(**bench.VITB, compute_dtype="bf16").to(dev)
opt = HipAdamW(model, lr=1.5e-4, weight_decay=5e-3, betas=(0.9, 0.95))
sched = get_cosine_schedule_with_warmup(opt, 50, 1000, lr_end=1.5e-7)
gen = torch.Generator(device=dev); gen.manual_seed(42)
pool = [torch.rand(B, 1, 96, 96, 96, device=dev, generator=gen) for _ in range(2)]
noises = [torch.rand(B, model.num_patches, device=dev, generator=gen) for _ in range(2)]


def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        opt.zero_grad()
        loss, _, _ = model(pool[i % 2], noise=noises[i % 2])
        if BWD_RESERVE:
            lib.hct_set_cu_reserve(BWD_RESERVE)
        loss.backward()
        if BWD_RESERVE:
            lib.hct_set_cu_reserve(0)
        clip_gradients(model, 3.0)
        opt.step(); sched.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


block(5)
res = {"off": [], "on": []}
for rep in range(4):
    off(); res["off"].append(block(10))
    on(); res["on"].append(block(10))
off()
print(name, "off:", " ".join(f"{v:.3f}" for v in res["off"]), "| on:", " ".join(f"{v:.3f}" for v in res["on"]),
      f"| mean off {sum(res['off'])/4:.3f} on {sum(res['on'])/4:.3f} ms/step")
