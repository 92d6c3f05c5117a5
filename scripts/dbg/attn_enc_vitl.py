"""ViT-L encoder attention backward (129 tokens, head dim 64): two-phase vs the persistent key-owner kernel (bit 10 of the hook)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
B, N, H, dh = 96, 129, 16, 64
qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16(); d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev)
lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
outs = {}
for mode, nm in ((100182, "two-phase"), (101206, "bwd4<64,160,1>")):
    lib.hct_debug_force_simple_attention(mode)
    dq = torch.full_like(qkv, float("nan"))
    for _ in range(2):
        lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    outs[nm] = dq.float()
    print(f"{nm:22s}: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us  finite {bool(torch.isfinite(dq.float()).all())}  rel diff vs two-phase {((outs[nm] - outs['two-phase']).norm() / outs['two-phase'].norm()).item():.2e}", flush=True)
lib.hct_debug_force_simple_attention(101206)
