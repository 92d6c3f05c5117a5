"""Step time of BASELINE config #4 (ViT-L/16^3 on 128^3, learnable position table, decoder 768x8x16), bf16, one GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mae_oracle as O  # configuration table only
from headct_foundation_amd import MaskedAutoencoderViT
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients

cfg = O.CONFIGS["vitl"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = MaskedAutoencoderViT(**cfg.ctor_kwargs(), compute_dtype="bf16").to(dev)
opt = HipAdamW(model, lr=1e-4, weight_decay=5e-3, betas=(0.9, 0.95))
sched = get_cosine_schedule_with_warmup(opt, 10, 1000, lr_end=1e-7)
x = torch.rand(B, 1, 128, 128, 128, device=dev)
noise = torch.rand(B, model.num_patches, device=dev)


def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        opt.zero_grad()
        loss, _, _ = model(x, noise=noise)
        loss.backward()
        clip_gradients(model, 3.0)
        opt.step(); sched.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


block(3)
dt = block(8)
fl = O.algorithmic_flops_per_volume(cfg)
tot = fl["train_step"] if isinstance(fl, dict) and "train_step" in fl else 445.83e9
print(f"ViT-L/128^3 B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} volumes/s, {B*tot/dt/1e12:.0f} TFLOP/s algorithmic")
