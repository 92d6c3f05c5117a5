"""bwd5 (long-sequence five-product backward): parity against an fp32 torch reference and timing against the two-phase kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
if os.environ.get('HCT_LIB_TAG'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ['HCT_LIB_TAG']}.so")
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
ON, OFF = 100000 + 54 + 128 + 256 + 512, 100000 + 54  # ON: bwd5 for both head dims (the default takes it for head dim 48 only)


def ref(qkv, d_o, B, N, H, dh):
    q, k, v = qkv.float().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q = q.detach().requires_grad_(True); k = k.detach().requires_grad_(True); v = v.detach().requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    o = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, N, H * dh)
    (o * d_o.float()).sum().backward()
    return torch.stack([q.grad, k.grad, v.grad], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * dh)


def run(B, N, H, dh, mode, scale=1.0):
    g = torch.Generator(device=dev); g.manual_seed(N * 7 + dh)
    qkv = (torch.randn(B, N, 3 * H * dh, device=dev, generator=g) * scale).bfloat16()
    d_o = torch.randn(B, N, H * dh, device=dev, generator=g).bfloat16()
    o = torch.empty(B, N, H * dh, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev)
    lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, 1, o.data_ptr(), lse.data_ptr(), st)
    dqkv = torch.full_like(qkv, float("nan"))
    lib.hct_debug_force_simple_attention(mode)
    _lib.check(lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dqkv.data_ptr(), st), "bwd")
    torch.cuda.synchronize()
    return qkv, d_o, dqkv, (o, lse)


if "check" in sys.argv or len(sys.argv) == 1:
    for (B, N, H, dh, sc) in [(1, 513, 2, 48, 1.0), (2, 517, 3, 64, 1.0), (1, 300, 2, 48, 1.0), (2, 226, 2, 48, 1.0), (2, 544, 2, 64, 1.0),
                              (1, 576, 2, 48, 1.0), (3, 529, 5, 64, 3.0), (2, 257, 4, 64, 1.0), (40, 513, 16, 48, 1.0)]:
        qkv, d_o, dq, _ = run(B, N, H, dh, ON, sc)
        _, _, dq2, _ = run(B, N, H, dh, OFF, sc)
        r = ref(qkv, d_o, B, N, H, dh) if B * H <= 64 else dq2.float()
        e = lambda a: ((a.float() - r).norm() / r.norm()).item()
        parts = dq.float().view(B, N, 3, H * dh); rp = r.view(B, N, 3, H * dh)
        pe = [((parts[:, :, i] - rp[:, :, i]).norm() / rp[:, :, i].norm()).item() for i in range(3)]
        print(f"B={B} N={N} H={H} dh={dh} x{sc}: bwd5 rel err {e(dq):.2e} (dq/dk/dv {pe[0]:.2e} {pe[1]:.2e} {pe[2]:.2e}) | two-phase {e(dq2):.2e} | finite {bool(torch.isfinite(dq.float()).all())} | differing elements {int((dq != dq2).sum())} of {dq.numel()}",
              flush=True)

if "time" in sys.argv or len(sys.argv) == 1:
    for (tag, B, N, H, dh) in (("ViT-L decoder", 96, 513, 16, 48), ("DINO global crops", 128, 517, 12, 64)):
        for mode, nm in ((ON, "bwd5"), (OFF, "two-phase")):
            qkv, d_o, dq, (o, lse) = run(B, N, H, dh, mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, 1, dq.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            print(f"{tag} B={B} N={N} H={H} dh={dh} [{nm}]: {us:8.1f} us  {10.0 * B * H * N * N * dh / us / 1e6:6.1f} TF/s", flush=True)
lib.hct_debug_force_simple_attention(101206)

