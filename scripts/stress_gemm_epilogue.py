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
for (M, N, K) in [(165, 768, 768), (512, 512, 256), (2048, 768, 768), (2048, 3072, 768)]:
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
