"""wgrad GEMM (TN) time per launch: in-launch split fold vs separate fold kernel, decoder shapes (R = 55552 token rows)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import GemmArgs
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
R = 55552
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for M, N in ((768, 3072), (3072, 768), (768, 768), (2304, 768)):
    A = torch.randn(R, M, device=dev).bfloat16(); B = torch.randn(R, N, device=dev).bfloat16()
    Cm = torch.empty(M, N, device=dev)
    a = GemmArgs(); a.M, a.N, a.K = M, N, R
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), 1, M, 1
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), 1, N, 0
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), 0, N
    a.alpha = 1.0
    ws = torch.zeros(lib.hct_gemm_workspace_bytes(C.byref(a)), dtype=torch.uint8, device=dev)
    a.workspace_armed = 1
    res = {}
    for mode, nm in ((-7, "in-launch"), (-6, "separate")):
        lib.hct_debug_set_gemm_variant(mode)
        for cold in (0, 1):
            ts = []
            for rep in range(6):
                if cold: flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res[(nm, cold)] = sorted(ts)[len(ts) // 2]
    lib.hct_debug_set_gemm_variant(-6)
    print(f"dW[{M}x{N}]: in-launch warm {res[('in-launch',0)]:6.1f} cold {res[('in-launch',1)]:6.1f} us | separate warm {res[('separate',0)]:6.1f} cold {res[('separate',1)]:6.1f} us")
