"""In-kernel timeline of the persistent NT GEMM (diagnostic build: `python -m headct_foundation_amd.build --stamps`).

Wave 0 of every workgroup stamps s_memrealtime (100 MHz, chip-global) at four points of each tile:
  0 tile top | 1 first stage landed | 2 main loop done | 3 epilogue issued (all stores in flight)
Prints, per call, the chip-wide mean of each segment per tile index and the spread of phase between workgroups.
Shares only: the stamps' fences cost a little, never quote this build's run time.
"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libheadct_hip_{os.environ.get('HCT_STAMP_LIB', 'stamps')}.so")
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
lib.hct_debug_set_stamp_buffer.restype = C.c_int
lib.hct_debug_set_stamp_buffer.argtypes = [C.c_void_p, C.c_uint]
stamps = torch.zeros(256 * 64, dtype=torch.int32, device=dev)
_lib.check(lib.hct_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel()), "stamp buffer")


def call(name, M, N, K, out_f32=False, bias=False, residual=False, act=0, stagger=-1, colsum=False):
    a = GemmArgs()
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
    if colsum:
        cso = torch.zeros(N, device=dev); a.colsum_out = cso.data_ptr(); keep.append(cso)
    ws = torch.empty(max(16, lib.hct_gemm_workspace_bytes(C.byref(a))), dtype=torch.uint8, device=dev)
    lib.hct_debug_set_gemm_variant(256)
    lib.hct_debug_set_gemm_stagger(stagger)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        stamps.zero_()
        e0.record()
        _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), st), "gemm")
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    s = stamps.cpu().numpy().astype(np.int64).reshape(256, 16, 4) & 0xFFFFFFFF
    tiles = (M // 256) * ((N + 255) // 256)
    nblk = 256 if os.environ.get('HCT_NT_STREAMK_PAIRS', '20') != '1000000' else min(256, tiles)
    s = s[:nblk]
    t0 = s[:, 0, 0].min()
    print(f"{name}: M={M} N={N} K={K} stagger={stagger}: {us:.1f} us, {2.0*M*N*K/us/1e6:.0f} TF, {tiles} tiles on {nblk} WGs")
    print("   tile |  #WG | start(us) min/mean/max | wait-land | main loop | epilogue issue |   (all us, mean over WGs)")
    for i in range(16):
        live = s[:, i, 3] != 0
        if not live.any():
            break
        q = s[live, i, :].astype(np.float64)
        rel = (q[:, 0] - t0) / 100.0
        seg = (q[:, 1:] - q[:, :-1]) / 100.0
        print(f"   {i:4d} | {int(live.sum()):4d} | {rel.min():6.1f} {rel.mean():6.1f} {rel.max():6.1f} | {seg[:,0].mean():9.2f} | "
              f"{seg[:,1].mean():9.2f} | {seg[:,2].mean():9.2f}")
    last = s[:, :, 3].max()
    print(f"   kernel span by stamps: {(last - t0)/100.0:.1f} us")


Md = 256 * 217
which = [a for a in sys.argv[1:] if not a.lstrip("-").isdigit()] or ["qkv", "fc1", "dgelu", "proj"]
for stg in ([int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or (-1, 0)):
    if "qkv" in which: call("qkv fwd (plain bf16)", Md, 2304, 768, stagger=stg)
    if "fc1" in which: call("fc1 fwd (GELU, aux, bias)", Md, 3072, 768, bias=True, act=1, stagger=stg)
    if "dgelu" in which: call("fc2 dgrad (x gelu')", Md, 3072, 768, act=2, stagger=stg)
    # the forms the training step uses: forward saves gelu' (act 4), the backward multiplies by it and sums columns (act 5)
    if "fc1d" in which: call("fc1 fwd (GELU + saved gelu', bias)", Md, 3072, 768, bias=True, act=4, stagger=stg)
    if "mulaux" in which: call("fc2 dgrad (x saved gelu' + column sums)", Md, 3072, 768, act=5, stagger=stg, colsum=True)
    if "encmulaux" in which: call("encoder fc2 dgrad (x saved gelu' + column sums)", 256 * 55, 3072, 768, act=5, stagger=stg, colsum=True)
    if "encproj" in which: call("encoder proj fwd (+bias +res f32)", 256 * 55, 768, 768, out_f32=True, bias=True, residual=True, stagger=stg)
    if "proj" in which: call("proj fwd (+bias +res f32)", Md, 768, 768, out_f32=True, bias=True, residual=True, stagger=stg)
    # 651 tiles: with the stream-K remainder round (default; HCT_NT_STREAMK_PAIRS=1000000 switches it off) item 0 / 1 of a
    # workgroup are its follower / owner pieces, whose "epilogue issue" column is the slab hand-over / the fix-up + epilogue
    if "fc1dgrad" in which: call("fc1 dgrad (plain bf16, K=3072)", Md, 768, 3072, stagger=stg)
    if "encfc1dgrad" in which: call("encoder fc1 dgrad (plain bf16, K=3072, 165 tiles)", 256 * 55, 768, 3072, stagger=stg)
