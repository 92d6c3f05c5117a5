"""Throughput of the forward-only ViT encoder (feature extraction), ViT-B/16^3 on 96^3 volumes, bf16."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import ViT

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = ViT(in_chans=1, img_size=96, patch_size=16, hidden_size=768, mlp_dim=3072, num_layers=12, num_heads=12, pos_embed="sincos",
        num_register_tokens=0, compute_dtype="bf16").to(dev)
for B in (64, 256):
    x = torch.rand(B, 1, 96, 96, 96, device=dev)
    for _ in range(3):
        m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out, hidden = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    flops = B * 217 * 12 * (8 * 768 * 768 + 4 * 768 * 3072) * 1.0 + B * 12 * 4 * 217 * 217 * 768
    print(f"B={B}: {dt*1e3:.2f} ms per batch, {B/dt:.0f} volumes/s, {flops/dt/1e12:.0f} TFLOP/s (blocks only)")
