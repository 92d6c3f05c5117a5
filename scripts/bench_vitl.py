"""Step time of BASELINE config #4 (ViT-L/16^3 on 128^3, learnable position table, decoder 768x8x16), bf16, one GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from headct_foundation_amd import MaskedAutoencoderViT
from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
from headct_foundation_amd.optim import HipAdamW, clip_gradients

# BASELINE config #4: ViT-L/16^3 encoder on 128^3, decoder 768 x 8 layers x 16 heads
VITL = dict(input_size=128, patch_size=16, mask_ratio=0.75, in_chans=1, dropout_rate=0.0, spatial_dims=3, patch_embed="conv",
            pos_embed="learnable", encoder_depth=24, encoder_embed_dim=1024, encoder_mlp_dim=4096, encoder_num_heads=16,
            decoder_depth=8, decoder_embed_dim=768, decoder_mlp_dim=3072, decoder_num_heads=16, norm_pix_loss=False, use_bias=False)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = MaskedAutoencoderViT(**VITL, compute_dtype="bf16").to(dev)
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
tot = bench.algorithmic_train_flops_per_volume(VITL)  # 445.8 GFLOP per volume
print(f"ViT-L/128^3 B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} volumes/s, {B*tot/dt/1e12:.0f} TFLOP/s algorithmic")
