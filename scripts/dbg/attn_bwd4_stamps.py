"""Phase stamps of the persistent attention backward (bwd4): workgroup 0, waves 0 / 4 / 7, first four items."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
B, N, H, dh = 256, 217, 16, 48
qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16(); d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
o = torch.randn(B, N, H * dh, device=dev).bfloat16(); lse = torch.randn(B, H, N, device=dev) + 5; dq = torch.empty_like(qkv)
lib.hct_debug_force_simple_attention(100000 + int(sys.argv[1]) if len(sys.argv) > 1 else 100004)
names = ["top", "own loads landed", "barrier 1", "delta + barrier 2", "DMA issued", "q-block 0", "q-block 1", "q loop done", "reg prefetch + dQ tail", "stores issued"]
for extra, nm in ((0, "full"), (0x100, "no main"), (0x200, "no dQ")):
    lib.hct_debug_force_simple_attention(10 + 0x80 + extra)
    for _ in range(3):
        lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
    torch.cuda.synchronize()
    # workgroup 0's last item: item 3840 -> bh = 480 -> b = 30, h = 0; one stamp at the start of each of its first 120 rows
    rows = dq.view(B, N, 3 * H * dh)[30, :120, :4].contiguous().view(torch.int64).view(-1).cpu().tolist()
    t0 = rows[0]
    print(f"--- {nm}: us from the workgroup's first stamp (rows: wave 0 / 4 / 7; items 0..3)")
    for w, wn in enumerate((0, 4, 7)):
        for it in range(4):
            v = [(rows[(w * 4 + it) * 10 + k] - t0) / 100.0 for k in range(10)]
            print(f"wave {wn} item {it}: " + " ".join(f"{x:7.2f}" for x in v))
print("columns: " + " | ".join(names))
lib.hct_debug_force_simple_attention(10); lib.hct_debug_force_simple_attention(101206)
