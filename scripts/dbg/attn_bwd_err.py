"""Where does hct_attention_bwd differ from autograd?  Error per output part (dQ/dK/dV) and per 16-token block."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
B, N, H, dh = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (1, 217, 1, 48)
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * H * dh, device=dev).bfloat16()
d_o = torch.randn(B, N, H * dh, device=dev).bfloat16()
qr = qkv.float().requires_grad_(True)
q, k, v = qr.view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
o_ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, H * dh)
(o_ref * d_o.float()).sum().backward()
o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev)
lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
dqkv = torch.full_like(qkv, float("nan"))
lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dqkv.data_ptr(), st)
torch.cuda.synchronize()
got = dqkv.float().view(B, N, 3, H, dh); ref = qr.grad.view(B, N, 3, H, dh)
for part, nm in enumerate("QKV"):
    for h in range(H):
        line = []
        for t0 in range(0, N, 16):
            a, b = got[0, t0:t0 + 16, part, h], ref[0, t0:t0 + 16, part, h]
            line.append(f"{float((a - b).norm() / (b.norm() + 1e-9)):.2f}")
        print(f"d{nm} h{h}: " + " ".join(line))
    for d0 in range(0, dh, 16):
        a, b = got[0, :, part, 0, d0:d0 + 16], ref[0, :, part, 0, d0:d0 + 16]
        print(f"   d{nm} h0 cols {d0}-{d0+15}: {float((a - b).norm() / (b.norm() + 1e-9)):.3f}")
