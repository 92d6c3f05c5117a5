"""Where do the small device-to-device copies of a training step come from?  (torch profiler, python stacks)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from headct_foundation_amd import MaskedAutoencoderViT, _lib
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(42)
B = 64
model = MaskedAutoencoderViT(**bench.VITB, compute_dtype="bf16").to(dev)
opt = HipAdamW(model, lr=1.5e-4, weight_decay=5e-3, betas=(0.9, 0.95))
sched = get_cosine_schedule_with_warmup(opt, 50, 1000, lr_end=1.5e-7)
x = torch.rand(B, 1, 96, 96, 96, device=dev)
losses = torch.zeros(8, device=dev)
def step(i):
    opt.zero_grad()
    loss, _, _ = model(x)
    loss.backward()
    clip_gradients(model, 3.0)
    opt.step(); sched.step()
    losses[i] = loss.detach()
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
from collections import Counter
c = Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "copy_" in n or "emset" in n or "fill_" in n or "aten::zero_" in n or "aten::clone" in n or "aten::to" == n:
        st = [s for s in (e.stack or []) if "headct" in s or "bench" in s or "find_copies" in s or "optim" in s]
        c[(n, st[0] if st else "?")] += 1
for k, v in c.most_common(40):
    print(v, k)
