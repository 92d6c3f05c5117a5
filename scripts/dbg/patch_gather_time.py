"""hct_patch_gather on DINO's crops (64 x 3 x 96^3, 12^3 patches): per-patch kernel (identity index table) vs pencil kernel (no table)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for (B, C, S, P, xdt) in ((64, 3, 96, 12, torch.float32), (64, 3, 96, 12, torch.float16), (256, 1, 96, 16, torch.float32)):
    g = S // P; L = g ** 3
    x = torch.rand(B, C, S, S, S, device=dev).to(xdt)
    rows = torch.empty(B * L, C * P ** 3, dtype=torch.bfloat16, device=dev)
    ids = torch.arange(L, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
    xd = _lib.HCT_F16 if xdt == torch.float16 else _lib.HCT_F32
    for nm, ip in (("per-patch", ids.data_ptr()), ("pencil", None)):
        f = lambda: lib.hct_patch_gather(x.data_ptr(), xd, ip, B, C, S, P, L, L, rows.data_ptr(), _lib.HCT_BF16, st)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        gb = (x.numel() * x.element_size() + rows.numel() * 2) / 1e9
        print(f"B={B} C={C} S={S} P={P} {str(xdt):14s} [{nm:9s}]: {us:7.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s", flush=True)
