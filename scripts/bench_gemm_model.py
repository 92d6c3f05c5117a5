"""hct_gemm on the exact calls of one decoder / encoder block (B=256), with the real epilogues."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream

def call(name, M, N, K, out_f32=False, bias=False, residual=False, act=0, tn=False, reps=8):
    a = GemmArgs()
    if tn:
        A = torch.randn(K, M, device=dev).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, M, 1
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, N, 0
    else:
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, K, 0
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, K, 1
    a.M, a.N, a.K = M, N, K
    Cm = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_F32 if out_f32 else HCT_BF16, N
    keep = [A, B, Cm]
    if bias:
        bv = torch.randn(N, device=dev); a.bias = bv.data_ptr(); keep.append(bv)
    if residual:
        r = torch.randn(M, N, device=dev); a.residual = r.data_ptr(); a.ldr = N; keep.append(r)
    if act:
        aux = torch.randn(M, N, device=dev).bfloat16(); a.aux, a.aux_dtype, a.ldaux = aux.data_ptr(), HCT_BF16, N; keep.append(aux)
    a.act = act; a.alpha = 1.0
    ws = torch.empty(max(16, lib.hct_gemm_workspace_bytes(C.byref(a))), dtype=torch.uint8, device=dev)
    for _ in range(2):
        _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"  {name:34s} M={M:6d} N={N:5d} K={K:6d}  {2.0*M*N*K/us/1e6:7.1f} TF  {us:7.1f} us")
    return us

for tag, M, var in (("decoder", 256 * 217, 256), ("decoder", 256 * 217, 0), ("decoder", 256 * 217, 4), ("encoder", 256 * 55, 256), ("encoder", 256 * 55, 0), ("encoder", 256 * 55, 4)):
    lib.hct_debug_set_gemm_variant(var)
    print(tag, "block, forward + backward GEMMs, NT variant", var, "(0 = auto: 192- or 256-row tiles)" if var == 0 else "")
    t = 0
    t += call("qkv fwd (plain bf16)", M, 2304, 768)
    t += call("proj fwd (+bias +residual, f32)", M, 768, 768, out_f32=True, bias=True, residual=True)
    t += call("fc1 fwd (GELU, aux, bias)", M, 3072, 768, bias=True, act=1)
    t += call("fc2 fwd (+bias +residual, f32)", M, 768, 3072, out_f32=True, bias=True, residual=True)
    t += call("fc2 dgrad (x gelu')", M, 3072, 768, act=2)
    t += call("fc1 dgrad", M, 768, 3072)
    t += call("proj dgrad", M, 768, 768)
    t += call("qkv dgrad", M, 768, 2304)
    t += call("fc2 wgrad", 768, 3072, M, out_f32=True, tn=True)
    t += call("fc1 wgrad", 3072, 768, M, out_f32=True, tn=True)
    t += call("proj wgrad", 768, 768, M, out_f32=True, tn=True)
    t += call("qkv wgrad", 2304, 768, M, out_f32=True, tn=True)
    print(f"  total {t/1e3:.2f} ms per block")
