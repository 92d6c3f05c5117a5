"""Per-shape view of the GEMM launches inside the training step (bench.py's workload): HIP events around every launch
(csrc/prof.hip), keyed by (M, N, K, epilogue mode, tiles, stream-K tiles).

    python scripts/nt_shapes.py [--config vitb|vitl] [--batch B] [--steps 6] [--out profiles/r03_nt_shapes.json]

Columns: rounds = tiles / 256 CUs; us = mean launch duration; TF/s = 2MNK / duration; frac = of the 2.5 PFLOP/s dense bf16 peak.
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from headct_foundation_amd import MaskedAutoencoderViT, _lib
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="vitb")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--out", default="")
args = ap.parse_args()
if os.environ.get('HCT_LIB_TAG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ['HCT_LIB_TAG']}.so")
lib = _lib.load()
arch, default_batch, workload, _ = bench.WORKLOADS[args.config]
B = args.batch or default_batch
dev = torch.device("cuda", 0)
torch.manual_seed(42)
model = MaskedAutoencoderViT(**arch, compute_dtype="bf16").to(dev)
opt = HipAdamW(model, lr=1.5e-4, weight_decay=5e-3, betas=(0.9, 0.95))
sched = get_cosine_schedule_with_warmup(opt, 50, 1000, lr_end=1.5e-7)
S = arch["input_size"]
pool = [torch.rand(B, 1, S, S, S, device=dev) for _ in range(2)]


def step(i):
    opt.zero_grad()
    loss, _, _ = model(pool[i % 2])
    loss.backward()
    clip_gradients(model, 3.0)
    opt.step(); sched.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
lib.hct_prof_reset()
lib.hct_prof_enable(0x3)
t0 = time.perf_counter()
for i in range(args.steps):
    step(i)
torch.cuda.synchronize()
ms_step = (time.perf_counter() - t0) / args.steps * 1e3
lib.hct_prof_enable(0)
MODES = {0: "generic", 1: "plain", 2: "+res f32", 3: "GELU", 4: "xGELU'", 5: "plain f32", 6: "xGELU'+cs"}
out = {"workload": workload, "batch": B, "ms_per_step_with_events": round(ms_step, 3), "nt": [], "tn": []}
for cls, key in ((0, "nt"), (1, "tn")):
    rows = _lib.prof_shapes(lib, cls)
    tot = 0.0
    print(f"--- {key.upper()} GEMM launches per step ({'mode' if cls == 0 else 'splits'}) ---")
    print(f"{'M':>6} {'N':>5} {'K':>6} {'mode':>10} {'tiles':>6} {'rounds':>6} {'sk':>4} {'n/step':>6} {'us':>8} {'ms/step':>8} {'TF/s':>7} {'frac':>6}")
    for r in sorted(rows, key=lambda r: -r.total_ms):
        n = r.launches / args.steps
        us = r.total_ms / r.launches * 1e3
        tf = r.work / (r.total_ms * 1e-3) / 1e12
        tot += r.total_ms / args.steps
        mode = MODES.get(r.mode, str(r.mode)) if cls == 0 else str(r.mode)
        print(f"{r.M:>6} {r.N:>5} {r.K:>6} {mode:>10} {r.tiles:>6} {r.tiles / 256:>6.2f} {r.sk_tiles:>4} {n:>6.1f} {us:>8.1f} {r.total_ms / args.steps:>8.3f} {tf:>7.0f} {tf / 2500:>6.3f}")
        out[key].append(dict(M=r.M, N=r.N, K=r.K, mode=mode, tiles=r.tiles, rounds=round(r.tiles / 256, 2), sk_tiles=r.sk_tiles, launches_per_step=n,
                             avg_us=round(us, 1), ms_per_step=round(r.total_ms / args.steps, 3), tflops=round(tf), frac=round(tf / 2500, 3),
                             alg_MB=round(r.bytes / r.launches / 1e6, 1)))
    print(f"total {tot:.2f} ms/step")
    out[key + "_ms_per_step"] = round(tot, 3)
print(f"step (with events) {ms_step:.2f} ms")
if args.out:
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
