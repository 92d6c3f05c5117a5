"""bwd5: time against the item count and the block count (fixed cost per item vs cost per query block)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for (B, N, H, dh) in ((16, 513, 16, 48), (32, 513, 16, 48), (96, 513, 16, 48), (96, 289, 16, 48), (96, 385, 16, 48), (96, 576, 16, 48)):
    qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16(); d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
    o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev); dq = torch.empty_like(qkv)
    lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    for mode, nm in ((101206, "bwd5"), (100054, "two-phase")):
        lib.hct_debug_force_simple_attention(mode)
        for _ in range(2):
            lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        print(f"B*H={B*H:5d} N={N} ({(N+31)//32} blocks) [{nm:9s}]: {us:8.1f} us = {us / max(1, B * H / 256):7.1f} us per round of 256 items", flush=True)
lib.hct_debug_force_simple_attention(101206)
