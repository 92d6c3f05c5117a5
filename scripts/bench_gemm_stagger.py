"""Start-phase stagger sweep of the persistent 256x256 NT kernel on the decoder-block shapes (real epilogues)."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
with contextlib.redirect_stdout(io.StringIO()):
    import bench_gemm_model as g
lib = g.lib
Md = 256 * 217
shapes = [("qkv fwd", dict(M=Md, N=2304, K=768)), ("proj fwd", dict(M=Md, N=768, K=768, out_f32=True, bias=True, residual=True)),
          ("fc1 fwd", dict(M=Md, N=3072, K=768, bias=True, act=1)), ("fc2 fwd", dict(M=Md, N=768, K=3072, out_f32=True, bias=True, residual=True)),
          ("fc2 dgrad", dict(M=Md, N=3072, K=768, act=2)), ("fc1 dgrad", dict(M=Md, N=768, K=3072)), ("qkv dgrad", dict(M=Md, N=768, K=2304))]
lib.hct_debug_set_gemm_variant(256)
for name, kw in shapes:
    row = []
    for st in (0, 1, 2, 3, 4, 6, 8, -1):
        lib.hct_debug_set_gemm_stagger(st)
        with contextlib.redirect_stdout(io.StringIO()):
            us = g.call(name, kw["M"], kw["N"], kw["K"], **{k: v for k, v in kw.items() if k not in "MNK"})
        row.append(f"{st}:{us:6.1f}")
    print(f"{name:10s} " + "  ".join(row))
lib.hct_debug_set_gemm_stagger(-1)
