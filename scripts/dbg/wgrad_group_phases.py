"""Grouped wgrad: rate of the whole-tile rounds against the stream-K remainder round (isolated launches, R = 55552).
   python scripts/dbg/wgrad_group_phases.py"""
import ctypes as C, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs
lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 55552
torch.manual_seed(0)


def run(shapes, label):
    bufs = {}
    def buf(w, k):
        if (w, k) not in bufs:
            bufs[(w, k)] = torch.randn(R, w, device=dev).to(torch.bfloat16)
        return bufs[(w, k)]
    n = len(shapes)
    jobs = (GemmArgs * n)()
    outs = []
    for i, (M, N) in enumerate(shapes):
        A, B = buf(M, i % 3), buf(N, 3 + i % 3)
        Cm = torch.empty(M, N, device=dev); outs.append(Cm)
        a = jobs[i]
        a.M, a.N, a.K = M, N, R
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, M, 1
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, N, 0
        a.C, a.c_dtype, a.ldc, a.alpha = Cm.data_ptr(), HCT_F32, N, 1.0
    nb = lib.hct_gemm_tn_group_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.hct_gemm_tn_group_prepare(C.cast(jobs, C.c_void_p), n, ws.data_ptr(), nb, st), "prepare")
    f = lambda: _lib.check(lib.hct_gemm_tn_group_run(C.cast(jobs, C.c_void_p), n, ws.data_ptr(), nb, st), "run")
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    fl = sum(2.0 * M * N * R for M, N in shapes)
    tiles = sum(((M + 255) // 256) * ((N + 255) // 256) for M, N in shapes)
    print(f"{label:>28}: {tiles:5d} tiles ({tiles / 256:.2f} rounds)  {ms:7.3f} ms  {fl / ms / 1e9:6.0f} TF/s  {ms * 1e3 / (tiles / 256):7.1f} us per round-equivalent")


blk = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
run([(768, 3072)] * 64 , "whole rounds, 3x12 jobs x64")  # 2304 tiles = 9 rounds
run(blk * 64, "whole rounds? block x64")                   # 6912 tiles = 27 rounds
run([(768, 3072)] * 4, "remainder only, 144 tiles")
run(blk * 2, "remainder only, 2 blocks")
run(blk * 8, "decoder-like: 8 blocks")
run(blk * 7 + [(768, 3072)] * 0, "7 blocks (756 tiles)")
