"""Micro-benchmark of hct_attention_fwd/bwd on the MAE shapes (B=256)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import _lib
if os.environ.get('HCT_LIB_TAG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ['HCT_LIB_TAG']}.so")
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream

def run(B, N, H, dh, mode, reps=10):
    lib.hct_debug_force_simple_attention(mode)
    qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16()
    d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
    o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=dev)
    dqkv = torch.empty_like(qkv)
    f = lambda: lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    b = lambda: lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dqkv.data_ptr(), st)
    out = []
    for fn, fl in ((f, 4.0), (b, 10.0)):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        out.append((us, fl * B * H * N * N * dh / us / 1e6))
    lib.hct_debug_force_simple_attention(0)
    return out

for tag, N, H, dh in (("decoder", 217, 16, 48), ("encoder", 55, 12, 64)):
    for mode, nm in ((3, "full-row"), (2, "online")):
        (fu, ft), (bu, bt) = run(256, N, H, dh, mode)
        print(f"{tag} N={N} H={H} dh={dh} [{nm:8s}] fwd {fu:7.1f} us {ft:6.1f} TF | bwd {bu:7.1f} us {bt:6.1f} TF")

for dbg, nm in ((0, "key-owner five-product (bwd3)"), (32, "two-phase 8 waves"), (8, "two-phase 4 waves"), (4, "single-phase 112 KB")):
    lib.hct_debug_force_simple_attention(10 + dbg)
    for tag, N, H, dh in (("decoder", 217, 16, 48), ("encoder", 55, 12, 64)):
        (fu, ft), (bu, bt) = run(256, N, H, dh, 3)
        print(f"{tag} bwd [{nm}]: {bu:7.1f} us {bt:6.1f} TF")
lib.hct_debug_force_simple_attention(10)

# ViT-L/128^3 decoder (config #4): 513 tokens, 16 heads x 48 -- MFMA kernels vs the fp32-math fallback it used before
for mode, nm in ((0, "mfma"), (1, "fp32-math kernels")):
    (fu, ft), (bu, bt) = run(32, 513, 16, 48, mode, reps=3)
    print(f"ViT-L decoder N=513 B=32 [{nm}] fwd {fu:8.1f} us {ft:6.1f} TF | bwd {bu:8.1f} us {bt:6.1f} TF")
