"""Repeated-call data check of every specialised NT epilogue (plain / +bias+residual / GELU / dGELU) against fp32 torch.

Outputs are pre-filled with a sentinel so a store that never lands is told apart from a store of wrong data.  This is the
check that exposed the gfx950 store-data hazard documented at HCT_STORE_GUARD in csrc/gemm.hip (single-shot parity tests
passed by luck most of the time).  Optional argv[1]: tag of a diagnostic library built with
`python -m headct_foundation_amd.build --variant TAG -D...`.
"""
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from headct_foundation_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{sys.argv[1]}.so")
from test_kernels_gpu import gemm, _rand
import torch.nn.functional as F
lib = _lib.load(); cuda = torch.device("cuda")
tot = {}
for (M, N, K) in [(165, 768, 768), (512, 512, 256), (2048, 768, 768), (2048, 3072, 768), (16384, 1536, 768), (9000, 3072, 256)]:  # the last two: more tiles than CUs (tile hand-over of the persistent kernel)
    A = _rand((M, K), cuda, torch.bfloat16, 1); B = _rand((N, K), cuda, torch.bfloat16, 2, 0.05)
    bias = _rand((N,), cuda, torch.float32, 3); res = _rand((M, N), cuda, torch.float32, 4)
    aux = _rand((M, N), cuda, torch.bfloat16, 7)
    plain = A.float() @ B.float().t()
    u = aux.float().requires_grad_(True); F.gelu(u).sum().backward()
    pre = plain + bias
    refs = {"res": plain + bias + res, "dgelu": plain * u.grad, "plain": plain, "gelu": F.gelu(pre)}
    for rep in range(20):
        outs = {}
        for tag in ("res", "dgelu", "plain", "gelu"):
            dt = torch.float32 if tag == "res" else torch.bfloat16
            t = torch.full((M, N), 777.0, dtype=dt, device=cuda); torch.cuda.synchronize(); del t
            if tag == "res": o = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res)
            elif tag == "dgelu": o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, act=2, aux=aux)
            elif tag == "plain": o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16)
            else:
                ax = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
                o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, bias=bias, act=1, aux=ax)
            o = o.float(); tol = 1e-3 if tag == "res" else 0.03
            bad = (o - refs[tag]).abs() > tol + tol * refs[tag].abs()
            k = (tag, M, N, K); tot.setdefault(k, [0, 0, 0])
            tot[k][0] += int(bad.any()); tot[k][1] += int(bad.sum()); tot[k][2] += int((o[bad] == 777.0).sum())
            del o
for k, v in tot.items():
    print(sys.argv[1:] or "main", k, "bad calls", v[0], "/20, bad elements", v[1], "of which sentinel (store never landed)", v[2])

# ---- run-to-run determinism + outlier check of the other MFMA kernels (wgrad TN GEMM, attention forward / backward) ----
from test_kernels_gpu import _attn_ref, _dt, _st


def outliers(x, ref, tol):
    return int(((x.float() - ref).abs() > tol + tol * ref.abs()).sum())


for (R, M, N) in [(55552, 768, 768), (14080, 2304, 768), (8192, 768, 3072)]:
    A = _rand((R, M), cuda, torch.bfloat16, 5); B = _rand((R, N), cuda, torch.bfloat16, 6)
    ref = A.float().t() @ B.float()
    first, nd, no = None, 0, 0
    for rep in range(10):
        o = gemm(lib, A, B, 1, 0, M, N, R)
        no += outliers(o, ref, 2e-3 * R ** 0.5)
        if first is None: first = o.clone()
        nd += int((o != first).sum())
    print("tn wgrad", (R, M, N), "outliers", no, "elements differing between repeats", nd)

for (Bb, Nt, H, dh) in [(64, 217, 16, 48), (64, 55, 12, 64)]:
    qkv = _rand((Bb, Nt, 3 * H * dh), cuda, torch.bfloat16, 11); d_o = _rand((Bb, Nt, H * dh), cuda, torch.bfloat16, 12)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, Bb, Nt, H, dh)
    (o_ref * d_o.float()).sum().backward()
    first, nd, no = None, 0, 0
    for rep in range(10):
        o = torch.empty(Bb, Nt, H * dh, dtype=torch.bfloat16, device=cuda); lse = torch.empty(Bb, H, Nt, dtype=torch.float32, device=cuda)
        _lib.check(lib.hct_attention_fwd(qkv.data_ptr(), Bb, Nt, H, dh, _dt(qkv), o.data_ptr(), lse.data_ptr(), _st()), "fwd")
        dqkv = torch.full_like(qkv, float("nan"))
        _lib.check(lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), Bb, Nt, H, dh, _dt(qkv), dqkv.data_ptr(), _st()), "bwd")
        no += outliers(o, o_ref.detach(), 0.03) + outliers(dqkv, qr.grad, 0.06)
        cur = torch.cat([o.flatten().float(), dqkv.flatten().float()])
        if first is None: first = cur.clone()
        nd += int((cur != first).sum())
    print("attention fwd+bwd", (Bb, Nt, H, dh), "outliers", no, "elements differing between repeats", nd)
