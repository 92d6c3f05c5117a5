"""bwd4 (persistent prefetching key-owner attention backward) against the two-phase kernel: equality of results on a
multi-item grid (more (batch, head) items than CUs) and time on the MAE decoder shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
if os.environ.get('HCT_LIB_TAG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ['HCT_LIB_TAG']}.so")
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream


def bwd(B, N, H, dh, mask, reps=0, seed=0):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    qkv = torch.randn(B, N, 3 * H * dh, device=dev, generator=g).bfloat16()
    d_o = torch.randn(B, N, H * dh, device=dev, generator=g).bfloat16()
    o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=dev)
    lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    dqkv = torch.full_like(qkv, float("nan"))
    lib.hct_debug_force_simple_attention(100000 + mask)
    call = lambda: lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dqkv.data_ptr(), st)
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
    return dqkv, us


for B, N, H in ((2, 217, 16), (20, 217, 16), (40, 200, 8), (33, 256, 9), (70, 193, 8)):
    a, _ = bwd(B, N, H, 48, 0)
    b, _ = bwd(B, N, H, 48, 4)
    c, _ = bwd(B, N, H, 48, 20)
    assert bool(torch.isfinite(c.float()).all()) and ((a.float() - c.float()).norm() / a.float().norm()).item() < 1e-2
    fin = bool(torch.isfinite(b.float()).all())
    err = ((a.float() - b.float()).norm() / a.float().norm()).item()
    print(f"B={B} N={N} H={H}: finite={fin} rel diff vs two-phase {err:.3e}", flush=True)
    assert fin and err < 1e-2
for mask, nm in ((0, "two-phase"), (1, "bwd3"), (4, "bwd4, 8 waves x 2 key tiles"), (20, "bwd4, 16 waves x 1 key tile")):
    _, us = bwd(256, 217, 16, 48, mask, reps=20)
    print(f"decoder B=256 N=217 H=16 dh=48 [{nm}]: {us:7.1f} us", flush=True)
for dbg, nm in ((0x100, "no main part"), (0x200, "no dQ part"), (0x300, "loads, delta, barriers and stores only")):
    lib.hct_debug_force_simple_attention(10 + dbg)
    for mask in (4, 20):
        lib.hct_debug_force_simple_attention(10 + dbg)
        _, us = bwd(256, 217, 16, 48, mask, reps=20)
        lib.hct_debug_force_simple_attention(10)
        print(f"bwd4 ablation ({'8 waves x 2 tiles' if mask == 4 else '16 waves x 1 tile'}) [{nm}]: {us:7.1f} us", flush=True)
