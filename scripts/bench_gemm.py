"""Micro-benchmark of hct_gemm on the MAE step's shapes (B=256): TFLOP/s per shape and variant, random operands."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs

lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream


def run(kind, M, N, K, variant=0, reps=10):
    a = GemmArgs()
    if kind == "nt":
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Cm = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        a.M, a.N, a.K = M, N, K
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, K, 0
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, K, 1
        a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_BF16, N
    else:  # tn: C[M,N] = A[R,M]^T B[R,N], R = K
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        Cm = torch.empty(M, N, dtype=torch.float32, device=dev)
        a.M, a.N, a.K = M, N, K
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, M, 1
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, N, 0
        a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_F32, N
    a.alpha = 1.0
    ws = torch.empty(max(16, lib.hct_gemm_workspace_bytes(C.byref(a))), dtype=torch.uint8, device=dev)
    lib.hct_debug_set_gemm_variant(variant)
    for _ in range(3):
        _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st)
    e1.record(); torch.cuda.synchronize()
    lib.hct_debug_set_gemm_variant(0)
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * M * N * K / ms / 1e9, ms * 1e3


def run_lib(kind, M, N, K, reps=10):
    """The same product through torch.matmul (hipBLASLt / rocBLAS) as a measuring stick only - nothing in the package calls it.
    bf16 output for both kinds (the TN product of the package writes fp32, so the library has less to store there)."""
    if kind == "nt":
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        f = lambda: torch.matmul(A, B.t())
    else:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        f = lambda: torch.matmul(A.t(), B)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * M * N * K / ms / 1e9, ms * 1e3


if __name__ == "__main__":
    with_lib = "--lib" in sys.argv
    Md, Me = 256 * 217, 256 * 55
    print("NT (forward / dgrad):  shape  128-tile TF | 256-tile TF")
    for M, N, K in [(Md, 2304, 768), (Md, 768, 768), (Md, 3072, 768), (Md, 768, 3072), (Md, 768, 2304), (Md, 4096, 768),
                    (Me, 2304, 768), (Me, 768, 768), (Me, 3072, 768), (Me, 768, 3072), (256 * 54, 768, 4096), (4096, 4096, 4096)]:
        t128, u128 = run("nt", M, N, K, 128)
        t256, u256 = run("nt", M, N, K, 256)
        extra = "" if not with_lib else " | library {:7.1f} TF ({:7.1f} us)".format(*run_lib("nt", M, N, K))
        print(f"  M={M:6d} N={N:5d} K={K:5d}   {t128:7.1f} TF ({u128:7.1f} us) | {t256:7.1f} TF ({u256:7.1f} us){extra}")
    print("TN (wgrad): C[M,N] over R rows")
    for M, N, R in [(2304, 768, Md), (768, 768, Md), (3072, 768, Md), (768, 3072, Md), (4096, 768, Md), (2304, 768, Me),
                    (3072, 768, Me), (768, 4096, 256 * 54)]:
        t, u = run("tn", M, N, R)
        extra = "" if not with_lib else " | library {:7.1f} TF ({:7.1f} us)".format(*run_lib("tn", M, N, R))
        print(f"  M={M:5d} N={N:5d} R={R:6d}   {t:7.1f} TF ({u:7.1f} us){extra}")
