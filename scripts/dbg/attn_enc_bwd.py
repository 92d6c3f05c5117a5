"""Encoder-shape attention backward (N = 55, head dim 64): the bwd3 instances against the two-phase kernel (results + time)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
DEFAULT = 101206


def bwd(B, N, H, dh, mask, reps=0):
    g = torch.Generator(device=dev); g.manual_seed(0)
    qkv = torch.randn(B, N, 3 * H * dh, device=dev, generator=g).bfloat16()
    d_o = torch.randn(B, N, H * dh, device=dev, generator=g).bfloat16()
    o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev)
    lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    dqkv = torch.full_like(qkv, float("nan"))
    lib.hct_debug_force_simple_attention(100000 + mask)
    call = lambda: lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dqkv.data_ptr(), st)
    assert call() == 0
    us = None
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
    torch.cuda.synchronize()
    lib.hct_debug_force_simple_attention(DEFAULT)
    return dqkv, us


for B, N, H in ((3, 55, 12), (40, 55, 12), (7, 33, 5), (5, 64, 3)):
    a, _ = bwd(B, N, H, 64, 0)
    for mask in (2, 34):
        b, _ = bwd(B, N, H, 64, mask)
        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
        assert bool(torch.isfinite(b.float()).all()) and err < 1e-2, (B, N, H, mask, err)
    print(f"B={B} N={N} H={H}: bwd3 instances agree with the two-phase kernel", flush=True)
for mask, nm in ((0, "two-phase"), (2, "bwd3, 2 waves x 2 key tiles"), (34, "bwd3, 4 waves x 1 key tile")):
    _, us = bwd(256, 55, 12, 64, mask, reps=30)
    print(f"encoder B=256 N=55 H=12 dh=64 [{nm}]: {us:6.1f} us", flush=True)
for B, N, H in ((3, 129, 16), (5, 160, 3), (4, 100, 2)):
    a, _ = bwd(B, N, H, 64, 0)
    for mask in (3,):
        b, _ = bwd(B, N, H, 64, mask)
        err = ((a.float() - b.float()).norm() / a.float().norm()).item()
        assert bool(torch.isfinite(b.float()).all()) and err < 1e-2, (B, N, H, mask, err)
    print(f"B={B} N={N} H={H}: bwd3 instances agree with the two-phase kernel", flush=True)
for mask, nm in ((0, "two-phase"), (3, "bwd3, 4 waves x 3 key tiles"), (54, "default")):
    _, us = bwd(64, 129, 16, 64, mask, reps=30)
    print(f"ViT-L encoder B=64 N=129 H=16 dh=64 [{nm}]: {us:6.1f} us", flush=True)
