"""Rate a user of the drop-in entry point gets: engine_pretrain_mae.train_one_epoch on a synthetic B = 256 loader (batches resident
on the device), against bench.py's bare step loop.  Two runs: asynchronous loss read-back (the default) and HCT_SYNC_LOSS=1 (the
reference's per-iteration synchronize + .item(), engine_pretrain_mae.py:73-74).

    python scripts/bench_engine.py [--iters 24]
"""
import argparse, logging, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import engine_pretrain_mae as E
from headct_foundation_amd import MaskedAutoencoderViT
from headct_foundation_amd.cfgnode import CfgNode
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=24)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B = 256
torch.manual_seed(42)
model = MaskedAutoencoderViT(**bench.VITB, compute_dtype="bf16").to(dev)
opt = HipAdamW(model, lr=1.5e-4, weight_decay=5e-3, betas=(0.9, 0.95))
sched = get_cosine_schedule_with_warmup(opt, 50, 1000, lr_end=1.5e-7)
pool = [torch.rand(B, 1, 96, 96, 96, device=dev) for _ in range(4)]


class Loader:
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def __iter__(self): return (pool[i % 4] for i in range(self.n))


cfg = CfgNode()
cfg.MODEL = CfgNode(); cfg.MODEL.NAME = "mae"
cfg.TRAIN = CfgNode(); cfg.TRAIN.GRAD_CLIP = 3.0
log = logging.getLogger("bench_engine")
log.addHandler(logging.NullHandler()); log.propagate = False


def bare(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        opt.zero_grad()
        loss, _, _ = model(pool[i % 4])
        loss.backward()
        clip_gradients(model, 3.0)
        opt.step(); sched.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def engine(n, sync):
    os.environ["HCT_SYNC_LOSS"] = "1" if sync else "0"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E.train_one_epoch(cfg, model, Loader(n), opt, sched, 0, 1, logger=log, device=dev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


bare(5)
res = {"bare": [], "engine_async": [], "engine_sync": []}
for rep in range(3):
    res["bare"].append(bare(args.iters))
    res["engine_async"].append(engine(args.iters, False))
    res["engine_sync"].append(engine(args.iters, True))
for k, v in res.items():
    print(f"{k:>13}: " + " ".join(f"{x:.3f}" for x in v) + f" | mean {sum(v) / len(v):.3f} ms/step = {B / (sum(v) / len(v)) * 1e3:.0f} volumes/s")
