"""Grouped wgrad (hct_gemm_tn_group_*) against the per-product split-K launches (hct_gemm) on the MAE step's shapes.

    python scripts/bench_wgrad_group.py [decoder|encoder]
"""
import ctypes as C, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs

lib = _lib.load()
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "decoder"
if which == "decoder":
    R, blocks = 55552, 8
else:
    R, blocks = 14080, 12
d, m = 768, 3072
per_block = [(d, m), (m, d), (d, d), (3 * d, d)]  # (M, N) of fc2, fc1, proj, qkv weight gradients; A = dY [R, M], B = X [R, N]
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
# operands shared between blocks by width (as many distinct buffers as one block needs; the data is irrelevant to the timing of a bf16 MFMA loop
# only through the clock the chip holds -- random, not zeros)
bufs = {w: [torch.randn(R, w, device=dev).to(torch.bfloat16) for _ in range(2)] for w in (d, m, 3 * d)}
outs, ops = [], []
for b in range(blocks):
    for (M, N) in per_block:
        A, Bm = bufs[M][b % 2], bufs[N][(b + 1) % 2]
        Cm = torch.empty(M, N, device=dev)
        outs.append(Cm)
        ops.append((A, Bm, Cm, M, N))
n = len(ops)
jobs = (GemmArgs * n)()
for i, (A, Bm, Cm, M, N) in enumerate(ops):
    a = jobs[i]
    a.M, a.N, a.K = M, N, R
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, M, 1
    a.B, a.b_dtype, a.ldb, a.transB = Bm.data_ptr(), HCT_BF16, N, 0
    a.C, a.c_dtype, a.ldc, a.alpha = Cm.data_ptr(), HCT_F32, N, 1.0
flops = sum(2.0 * M * N * R for (_, _, _, M, N) in ops)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ws1 = torch.zeros(max(lib.hct_gemm_workspace_bytes(C.byref(jobs[i])) for i in range(n)) + 16, dtype=torch.uint8, device=dev)


def separate():
    for i in range(n):
        _lib.check(lib.hct_gemm(C.byref(jobs[i]), ws1.data_ptr(), ws1.numel(), st), "hct_gemm")


ms_sep = timed(separate)
ref = [o.clone() for o in outs]
nb = lib.hct_gemm_tn_group_workspace_bytes(n)
ws2 = torch.empty(nb, dtype=torch.uint8, device=dev)
_lib.check(lib.hct_gemm_tn_group_prepare(C.cast(jobs, C.c_void_p), n, ws2.data_ptr(), nb, st), "prepare")


def grouped():
    _lib.check(lib.hct_gemm_tn_group_run(C.cast(jobs, C.c_void_p), n, ws2.data_ptr(), nb, st), "run")


ms_grp = timed(grouped)
err = max(float((o - r).norm() / r.norm()) for o, r in zip(outs, ref))
print(f"{which}: {n} products, {flops / 1e12:.2f} TFLOP | separate split-K launches {ms_sep:.3f} ms ({flops / ms_sep / 1e9:.0f} TF/s) | "
      f"grouped {ms_grp:.3f} ms ({flops / ms_grp / 1e9:.0f} TF/s) | max rel diff {err:.2e}")
