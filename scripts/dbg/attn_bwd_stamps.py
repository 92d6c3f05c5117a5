import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for B in (16, 256):
  for mode in (0x80, 0x80 | 64):
    N, H, dh = 217, 16, 48
    qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16(); d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
    o = torch.randn(B, N, H * dh, device=dev).bfloat16(); lse = torch.randn(B, H, N, device=dev); dq = torch.empty_like(qkv)
    lib.hct_debug_force_simple_attention(10 + mode)
    for _ in range(3):
        lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
    torch.cuda.synchronize()
    lib.hct_debug_force_simple_attention(10)
    ts = dq.view(-1)[:48].view(torch.int64).cpu().tolist()
    rel = [(t - ts[0]) / 100.0 for t in ts]  # 100 MHz -> us
    print(f"B={B} mode={mode:#x}: stamps (us from WG start): " + " ".join(f"{r:.2f}" for r in rel))
