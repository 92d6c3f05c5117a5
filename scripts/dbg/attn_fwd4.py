"""fwd4 (persistent prefetching attention forward) against the one-workgroup-per-head kernel: results on multi-item grids and
time on the MAE decoder shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream


def fwd(B, N, H, dh, mask, reps=0, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    qkv = torch.randn(B, N, 3 * H * dh, device=dev, generator=g).bfloat16()
    o = torch.full((B, N, H * dh), float("nan"), device=dev, dtype=torch.bfloat16)
    lse = torch.full((B, H, N), float("nan"), device=dev)
    lib.hct_debug_force_simple_attention(100000 + mask)
    call = lambda: lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    rc = call()
    assert rc == 0, rc
    us = None
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
    torch.cuda.synchronize()
    lib.hct_debug_force_simple_attention(101206)
    return o, lse, us


for B, N, H in ((2, 217, 16), (20, 217, 16), (40, 200, 8), (70, 193, 8), (35, 224, 8)):
    o0, l0, _ = fwd(B, N, H, 48, 0)
    o1, l1, _ = fwd(B, N, H, 48, 8)
    fin = bool(torch.isfinite(o1.float()).all() and torch.isfinite(l1).all())
    err = ((o0.float() - o1.float()).norm() / o0.float().norm()).item()
    print(f"B={B} N={N} H={H}: finite={fin} rel diff of o {err:.3e}, max |lse diff| {(l0 - l1).abs().max().item():.3e}", flush=True)
    assert fin and err < 5e-3 and (l0 - l1).abs().max().item() < 1e-3
for mask, nm in ((0, "one workgroup per head"), (8, "fwd4")):
    _, _, us = fwd(256, 217, 16, 48, mask, reps=20)
    print(f"decoder B=256 N=217 H=16 dh=48 forward [{nm}]: {us:7.1f} us", flush=True)
for dbg, nm in ((0x100, "image traffic only (no compute, no stores)"), (0x200, "compute + stores on stale images (no DMA after the first item)")):
    lib.hct_debug_force_simple_attention(10 + dbg)
    _, _, us = fwd(256, 217, 16, 48, 8, reps=20)
    lib.hct_debug_force_simple_attention(10)
    print(f"fwd4 ablation [{nm}]: {us:7.1f} us", flush=True)
