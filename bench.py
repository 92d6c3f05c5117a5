#!/usr/bin/env python3
"""Benchmark: CT-volumes/s of one full MAE pre-training step (BASELINE.json metric / config #2).

    python bench.py --gpus N --steps K --warmup W

A "step" = zero_grad -> forward -> backward (+ bucketed RCCL gradient all-reduce for N > 1) -> per-parameter
clip (3.0) -> AdamW -> cosine-warmup LR step, on one batch of synthetic 96^3 x 1ch volumes already resident in
HBM (engine_pretrain_mae.py:52-71).  ViT-B/16^3, mask 0.75, bf16 storage + MFMA with fp32 accumulation, fp32
master weights, B = 256 per GPU (weak scaling).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- dominant kernel (the bf16 MFMA "NT" GEMM): algorithmic FLOPs of its launches / their summed
                  duration, from HIP events recorded on the launch stream inside the timed region
                  (csrc/prof.hip); peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline -- the CPU oracle's train step (oracle/mae_oracle.py, plain PyTorch fp32) timed on this host's
                  cores on a bounded sample (ViT-B, B=8, a few steps), rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL's buffer exchange between ranks needs this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

VITB = dict(input_size=96, patch_size=16, mask_ratio=0.75, in_chans=1, dropout_rate=0.0, spatial_dims=3, patch_embed="conv",
            pos_embed="sincos", encoder_depth=12, encoder_embed_dim=768, encoder_mlp_dim=3072, encoder_num_heads=12,
            decoder_depth=8, decoder_embed_dim=768, decoder_mlp_dim=3072, decoder_num_heads=16, norm_pix_loss=False, use_bias=False)
# BASELINE config #4 (`--config vitl`): ViT-L/16^3 encoder on 128^3 volumes, learnable position table (the reference's default
# POS_EMBED), decoder 768 x 8 layers x 16 heads; 513-token decoder sequences = the attention-tile LDS stress case
VITL = dict(VITB, input_size=128, pos_embed="learnable", encoder_depth=24, encoder_embed_dim=1024, encoder_mlp_dim=4096,
            encoder_num_heads=16)
WORKLOADS = {
    "vitb": (VITB, 256, "BASELINE config #2: MAE ViT-B/16^3, 96^3x1ch, mask 0.75, full train step (fwd+bwd+clip+AdamW+LR)",
             "CT-volumes/sec MAE pretrain step (96^3, ViT-B/16^3, mask 0.75)"),
    "vitl": (VITL, 96, "BASELINE config #4: MAE ViT-L/16^3, 128^3x1ch, mask 0.75, decoder 768x8x16, full train step",
             "CT-volumes/sec MAE pretrain step (128^3, ViT-L/16^3, mask 0.75)"),
}


def algorithmic_train_flops_per_volume(c) -> float:
    """multiply-add = 2; GEMMs + QK^T + PV; patch-embed on kept tokens; train step = 3 x forward."""
    L = (c["input_size"] // c["patch_size"]) ** 3
    K = int(L * (1 - c["mask_ratio"]))
    pd = c["in_chans"] * c["patch_size"] ** 3
    D, M, Dd, Md = c["encoder_embed_dim"], c["encoder_mlp_dim"], c["decoder_embed_dim"], c["decoder_mlp_dim"]
    blk = lambda N, d, m: N * (8 * d * d + 4 * d * m) + 4 * N * N * d
    fwd = 2.0 * K * pd * D + c["encoder_depth"] * blk(K + 1, D, M) + 2.0 * (K + 1) * D * Dd + c["decoder_depth"] * blk(L + 1, Dd, Md) + 2.0 * L * Dd * pd
    return 3.0 * fwd


# BASELINE config #5 (`--config dino`): ViT-B/12^3 student + momentum teacher on 96^3 x 3-channel crops (configs/dino/dino_HeadCT.yaml: 512
# patches + class + 4 register tokens = 517 tokens, qkv bias, sincos table), 2 global + 8 local crops all at 96^3 (the reference
# resizes every crop to the final size, transforms.py:75-97), projection head 768-2048-2048-256-65536.
DINO = dict(vit=dict(in_chans=3, img_size=96, patch_size=12, hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12, pos_embed="sincos",
                     num_register_tokens=4, qkv_bias=True),
            head=dict(in_dim=768, out_dim=65536, hidden_dim=2048, bottleneck_dim=256), crops=10)


def dino_train_flops_per_volume(c) -> float:
    """Algorithmic FLOPs of one DINO iteration per input volume (multiply-add = 2; GEMMs + QK^T + PV): student forward + backward
    (3 x forward) on all crops, teacher forward on the two global crops, both through backbone and head."""
    v, h = c["vit"], c["head"]
    L = (v["img_size"] // v["patch_size"]) ** 3
    N = 1 + v["num_register_tokens"] + L
    D, M = v["hidden_size"], v["mlp_dim"]
    pd = v["in_chans"] * v["patch_size"] ** 3
    backbone = 2.0 * L * pd * D + v["num_layers"] * (N * (8 * D * D + 4 * D * M) + 4 * N * N * D)
    head = 2.0 * (h["in_dim"] * h["hidden_dim"] + h["hidden_dim"] ** 2 + h["hidden_dim"] * h["bottleneck_dim"] + h["bottleneck_dim"] * h["out_dim"])
    return (3.0 * c["crops"] + 2.0) * (backbone + head)


def run_dino(args, world, rank, device):
    """One DINO training iteration per step (engine_pretrain_dino.py:59-104): weight-decay schedule, teacher forward (2 crops), student
    forward (10 crops), DINO loss + centre update, backward, AdamW on backbone + head, LR step, momentum-teacher update."""
    import torch.distributed as dist
    from headct_foundation_amd.dino import DINOLoss, DinoDataParallel, DinoOptimizer, update_momentum_encoder, wd_cosine_scheduler
    from headct_foundation_amd.dino_model import DINOHead, MultiCropWrapper, ViTBackbone
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    B, G, V = args.batch or 64, world, DINO["crops"]  # the reference's DATA.BATCH_SIZE default (config.py:15)
    torch.manual_seed(42)
    mk = lambda: MultiCropWrapper(ViTBackbone(**DINO["vit"], compute_dtype=args.dtype), DINOHead(**DINO["head"], compute_dtype=args.dtype)).to(device)
    student, teacher = mk(), mk()
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad_(False)
    model, momentum_model = DinoDataParallel(student), DinoDataParallel(teacher)
    total = max(1000, args.steps + args.warmup)
    lr = 5e-4 * B * G / 256
    opt = DinoOptimizer(model, lr=lr, betas=(0.9, 0.999), weight_decay=0.04)
    sched = get_cosine_schedule_with_warmup(opt.primary, int(0.1 * total), total, lr_end=lr * 1e-3)
    wd = wd_cosine_scheduler(0.04, 0.4, 1, total)
    mom = wd_cosine_scheduler(0.999, 1.0, 1, total)
    crit = DINOLoss(DINO["head"]["out_dim"], V, 0.04, 0.04, 30, 200).to(device)
    torch.manual_seed(42 + rank)
    S = DINO["vit"]["img_size"]
    pool = [[torch.rand(B, 3, S, S, S, device=device) for _ in range(V)] for _ in range(2)]
    losses = torch.zeros(args.steps + args.warmup, device=device)

    def step(i):
        opt.param_groups[0]["weight_decay"] = float(wd[i])
        opt.zero_grad()
        crops = pool[i % 2]
        with torch.no_grad():
            t_out = momentum_model(crops[:2])['dino_output']
        s_out = model(crops)['dino_output']
        loss = crit(s_out.float(), t_out.float(), 0)
        loss.backward()
        model.reduce_head_gradients()
        opt.step()
        sched.step()
        update_momentum_encoder(student.backbone, teacher.backbone, float(mom[i]))
        update_momentum_encoder(student.head, teacher.head, float(mom[i]))
        losses[i] = loss.detach()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from headct_foundation_amd import _lib
    lib = _lib.load()
    for i in range(args.warmup):
        step(i)
    fence()
    prof = not args.no_prof
    if prof:
        lib.hct_prof_reset()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        if prof:  # the dominant kernel's launches (the same NT GEMM instances as the MAE step) bracketed on every 4th timed step
            lib.hct_prof_enable((0x1F if args.prof_all else 0x1) if (i - args.warmup) % 4 == 0 else 0)
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if prof:
        lib.hct_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lv = losses.cpu()
    if not torch.isfinite(lv).all():
        raise SystemExit(f"non-finite loss during the benchmark: {lv.tolist()}")
    roof, extra = roofline_from_events(lib, args.steps, elapsed, "dino") if prof else (None, {})
    if rank == 0:
        fl = dino_train_flops_per_volume(DINO)
        vols = B * G * args.steps
        print(json.dumps({
            "metric": "CT-volumes/sec DINO pretrain step (ViT-B/12^3 student + teacher, 96^3x3ch, 2 global + 8 local crops)",
            "value": round(vols / elapsed, 2), "unit": "CT-volumes/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE config #5: DINO, ViT-B/12^3 (517 tokens, 4 register tokens), head 768-2048-2048-256-65536, full iteration",
                       "per_gpu_batch": B, "global_batch": B * G, "crops_per_volume": V, "parallelism": f"dp{G}",
                       "algorithmic_GFLOP_per_volume": round(fl / 1e9, 1)},
            "step_mfma_frac": round(vols / elapsed * fl / G / 1e12 / PEAK_BF16_TFLOPS, 4), "roofline": roof, "other_kernels": extra or None,
            "loss_first": round(float(lv[0]), 5), "loss_last": round(float(lv[-1]), 5)}), flush=True)


def pmc_traffic(config: str = "vitb"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary OF THIS CONFIGURATION (separate passes
    cannot run inside the timed region); None if there is none.  Files: profiles/r<NN>[_tag]_pmc_nt256[_<config>].json -- no
    config suffix = vitb (config #2); the highest round wins, then the untagged file of that round."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_nt256*.json")):
        m = re.match(r"r(\d+)(?:_([a-z0-9]+))?_pmc_nt256(?:_([a-z]+))?\.json$", os.path.basename(f))
        if not m or (m.group(3) or "vitb") != config:
            continue
        key = (int(m.group(1)), m.group(2) is None, m.group(2) or "")
        if best is None or key > best[0]:
            best = (key, f)
    if best is None:
        return None
    try:
        with open(best[1]) as f:
            d = json.load(f)
        return {"hbm_bytes_per_launch": round(d["hbm_bytes_per_launch"]), "source": os.path.relpath(best[1], ROOT),
                "note": "2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 FETCH_SIZE correction applied"}
    except Exception:
        return None


def roofline_from_events(lib, steps: int, elapsed: float, config: str):
    """`roofline` object of the JSON line from the HIP events the library recorded around the NT GEMM launches of the sampled steps
    (every 4th timed step), plus the other kernel classes if they were timed."""
    sampled = (steps + 3) // 4
    ms, n, w, nt_bytes = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    from headct_foundation_amd import _lib
    _lib.check(lib.hct_prof_read(0, C.byref(ms), C.byref(n), C.byref(w)), "hct_prof_read")
    _lib.check(lib.hct_prof_read_bytes(0, C.byref(nt_bytes)), "hct_prof_read_bytes")
    roof, extra = None, {}
    traf = pmc_traffic(config)
    if n.value and ms.value > 0:
        ach = w.value / (ms.value * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "gemm_bf16_nt256_kernel<*> (all epilogue modes)", "achieved": round(ach, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": (traf or {}).get("hbm_bytes_per_launch"), "traffic_detail": traf,
                "algorithmic_TFLOP_per_launch": round(w.value / n.value / 1e12, 4),
                "algorithmic_bytes_per_launch": round(nt_bytes.value / n.value),
                "launches_per_step": n.value // max(1, sampled), "sampled_steps": sampled, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                "time_share_of_step": round(ms.value * 1e-3 * steps / sampled / elapsed, 4)}
    for kid, name in ((1, "gemm_bf16_tn (grouped weight gradients)"), (3, "attention_fwd"), (4, "attention_bwd"), (2, "gemm_generic")):
        _lib.check(lib.hct_prof_read(kid, C.byref(ms), C.byref(n), C.byref(w)), "hct_prof_read")
        if n.value and ms.value > 0:
            extra[name] = {"TFLOP/s": round(w.value / (ms.value * 1e-3) / 1e12, 2), "time_share_of_step": round(ms.value * 1e-3 * steps / sampled / elapsed, 4)}
    lib.hct_prof_reset()
    return roof, extra


def _host_threads() -> int:
    """Threads to use for the CPU baseline: the CPUs this process may run on, capped at the GPU box's 16-CPU share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(seconds_budget: float = 25.0):
    """Oracle train step on the host cores: ViT-B, B=8 (SURVEY 8d / BASELINE.md 3).  Bounded sample (~seconds_budget)."""
    from oracle import mae_oracle as O
    threads = _host_threads()
    torch.set_num_threads(threads)
    cfg = O.CONFIGS["vitb"]
    B = 8
    st = O.TrainState(O.make_params(cfg, 42, generic=False))
    hp = dict(base_lr=1.5e-4 * B / 256, min_lr=1.5e-7, warmup=2, total=100, weight_decay=5e-3, grad_clip=3.0)
    x, noise = O.make_volume(cfg, B, 42), O.make_noise(cfg, B, 42)
    times = []
    t_start = time.time()
    for i in range(16):  # ~14 s of CPU work at 0.87 s/step on the GPU box, cut off by seconds_budget on slower hosts
        t0 = time.time()
        O.train_step(cfg, st, x, noise, **hp)
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    timed = sorted(times[1:]) if len(times) > 1 else times  # first step = warm-up unless it is all the budget allowed
    med = timed[len(timed) // 2]
    return {"value": round(B / med, 3), "unit": "CT-volumes/s", "cores": threads, "kind": "port",
            "sample": f"oracle/mae_oracle.py train_step, ViT-B/16^3 96^3x1ch fp32, B={B}, median of {len(timed)} step(s)"
                      f"{' after 1 warm-up' if len(times) > 1 else ' (no warm-up fitted the budget)'} ({med:.2f} s/step); "
                      f"{threads} torch threads, host reports {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vitb", choices=sorted(WORKLOADS) + ["dino"],
                    help="vitb = BASELINE config #2 (the headline metric), vitl = config #4, dino = config #5")
    ap.add_argument("--batch", type=int, default=0, help="volumes per GPU (default: 256 for vitb, 96 for vitl, 64 for dino)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--prof-all", action="store_true", help="also time TN GEMM / attention launches (adds event overhead)")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    args = ap.parse_args()

    import torch.distributed as dist
    from headct_foundation_amd import MaskedAutoencoderViT, _lib
    from headct_foundation_amd.ddp import DistributedDataParallel
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    from headct_foundation_amd.optim import HipAdamW, clip_gradients

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    # rehearsal of the N>1 path on a one-GPU box: HCT_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (the
    # driver's runs use one rank per GPU over RCCL, the default)
    rehearsal = os.environ.get("HCT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # HCT_BENCH_RCCL1=1 (one-GPU box): world size ONE over the real "nccl" backend (= RCCL) with the data-parallel wrapper forced on, so
    # that communicator set-up, the bucketed asynchronous all-reduce on RCCL's stream, the CU reserve and the final wait all execute
    # on hardware (the sums are those of one rank); the step-time difference to the plain run bounds the compute-side cost of the wrapper
    rccl1 = os.environ.get("HCT_BENCH_RCCL1") == "1" and world == 1 and args.config != "dino"
    if world > 1 or rccl1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rccl1:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", device_id=device, rank=0, world_size=1)
        elif rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    lib = _lib.load()
    if args.config == "dino":
        run_dino(args, world, rank, device)
        if world > 1:
            dist.destroy_process_group()
        return
    arch, default_batch, workload, metric = WORKLOADS[args.config]
    B, G = args.batch or default_batch, world
    S = arch["input_size"]
    torch.manual_seed(42)  # reference init scheme at seed 42 (same weights on every rank; DDP broadcasts rank 0's anyway)
    model = MaskedAutoencoderViT(**arch, compute_dtype=args.dtype).to(device)
    ddp = DistributedDataParallel(model, device_ids=[device], bucket_cap_mb=args.bucket_mb, force_collectives=rccl1) if (world > 1 or rccl1) else model
    total_steps = max(1000, args.steps + args.warmup)
    base_lr = 1.5e-4 * B * G / 256  # main_pretrain_mae.py:149-151
    opt = HipAdamW(ddp, lr=base_lr, weight_decay=5e-3, betas=(0.9, 0.95))
    sched = get_cosine_schedule_with_warmup(opt, int(0.05 * total_steps), total_steps, lr_end=base_lr * 1e-3)

    torch.manual_seed(42 + rank)  # main_pretrain_mae.py:213: every rank draws its own volumes and masks
    pool = [torch.rand(B, 1, S, S, S, device=device) for _ in range(4)]
    losses = torch.zeros(args.steps + args.warmup, device=device)

    def step(i):
        opt.zero_grad()
        loss, _, _ = ddp(pool[i % 4])  # the masking noise is drawn inside forward, as in the reference (mae.py:206)
        loss.backward()
        clip_gradients(ddp, 3.0)
        opt.step()
        sched.step()
        losses[i] = loss.detach()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    prof = not args.no_prof
    if prof:
        lib.hct_prof_reset()
    # the per-launch HIP events cost ~3 us a pair: bracket the dominant kernel's launches on every 4th timed step only
    prof_mask = 0x1F if args.prof_all else 0x1
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        if prof:
            lib.hct_prof_enable(prof_mask if (i - args.warmup) % 4 == 0 else 0)
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if prof:
        lib.hct_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    lv = losses.cpu()
    if not torch.isfinite(lv).all():
        raise SystemExit(f"non-finite loss during the benchmark: {lv.tolist()}")

    roof, extra = roofline_from_events(lib, args.steps, elapsed, args.config) if prof else (None, {})
    if rank == 0:
        vols = B * G * args.steps
        flops_vol = algorithmic_train_flops_per_volume(arch)
        out = {
            "metric": metric,
            "value": round(vols / elapsed, 2), "unit": "CT-volumes/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload,
                       "per_gpu_batch": B, "global_batch": B * G, "parallelism": f"dp{G}", "init": "reference init, seed 42",
                       "algorithmic_GFLOP_per_volume": round(flops_vol / 1e9, 2)},
            "step_mfma_frac": round(vols / elapsed * flops_vol / G / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roof, "loss_first": round(float(lv[0]), 5), "loss_last": round(float(lv[-1]), 5),
        }
        if prof and extra:
            out["other_kernels"] = extra
        if rccl1:
            out["config"]["parallelism"] = "dp1 over RCCL (HCT_BENCH_RCCL1: wrapper forced on at world size 1)"
            out["rccl_buckets_per_step"] = [int(e - b) * 4 for b, e in ddp.launched]
        if world == 1 and not args.no_cpu_baseline and args.config == "vitb":  # the CPU sample is the headline workload
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1 or rccl1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
