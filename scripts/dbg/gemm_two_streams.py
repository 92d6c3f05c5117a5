"""Do a forward / dgrad product (NT) and a wgrad product (TN) of one decoder block fill each other's partly filled last rounds
when they are launched on two streams?  Times the pair back to back on one stream against the same pair on two streams."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs

lib = _lib.load()
dev = torch.device("cuda")


def nt(M, N, K):
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Cm = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, K, 0
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, K, 1
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_BF16, N
    a.alpha = 1.0
    return a, (A, B, Cm), None


def tn(M, N, R):
    A = torch.randn(R, M, device=dev).bfloat16(); B = torch.randn(R, N, device=dev).bfloat16()
    Cm = torch.empty(M, N, dtype=torch.float32, device=dev)
    a = GemmArgs()
    a.M, a.N, a.K = M, N, R
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, M, 1
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, N, 0
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_F32, N
    a.alpha = 1.0
    ws = torch.zeros(lib.hct_gemm_workspace_bytes(C.byref(a)), dtype=torch.uint8, device=dev)
    return a, (A, B, Cm), ws


def launch(job, stream):
    a, _, ws = job
    _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream.cuda_stream), "gemm")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    Md = 256 * 217
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    for name, j1, j2 in [("fc1 dgrad (NT 651 tiles, K=3072) + fc2 wgrad", nt(Md, 768, 3072), tn(768, 3072, Md)),
                         ("fc2 dgrad (NT 2604 tiles, K=768) + fc1 wgrad", nt(Md, 3072, 768), tn(3072, 768, Md)),
                         ("proj dgrad (NT 651 tiles, K=768) + qkv wgrad", nt(Md, 768, 768), tn(2304, 768, Md))]:
        def seq():
            launch(j1, main); launch(j2, main)

        def par():
            ev = torch.cuda.Event(); ev.record(main)
            side.wait_event(ev)
            launch(j2, side)
            launch(j1, main)
            ev2 = torch.cuda.Event(); ev2.record(side)
            main.wait_event(ev2)

        t1 = timeit(lambda: launch(j1, main)); t2 = timeit(lambda: launch(j2, main))
        print(f"{name}: alone {t1:.1f} + {t2:.1f} = {t1 + t2:.1f} us | one stream {timeit(seq):.1f} us | two streams {timeit(par):.1f} us", flush=True)
