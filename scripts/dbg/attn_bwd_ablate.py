"""Timing-only ablations of the key-owner attention backward (outputs are wrong with any bit set)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
def t(B, N, H, dh, dbg, reps=10):
    lib.hct_debug_force_simple_attention(10 + dbg)
    qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16(); d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
    o = torch.randn(B, N, H * dh, device=dev).bfloat16(); lse = torch.randn(B, H, N, device=dev); dq = torch.empty_like(qkv)
    f = lambda: lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    lib.hct_debug_force_simple_attention(10)
    return e0.elapsed_time(e1) / reps * 1e3
for dbg, nm in ((0, "one wave per SIMD, hoisted"), (64, "two waves per SIMD")):
    print(f"N=217 {nm:30s}: B=256 {t(256, 217, 16, 48, dbg):8.1f} us   B=16 {t(16, 217, 16, 48, dbg, reps=20):8.1f} us")
print(f"N=55: {t(256, 55, 12, 64, 0):8.1f} us")
