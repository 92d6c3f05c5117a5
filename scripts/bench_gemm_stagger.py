import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run, lib
Md, Me = 256 * 217, 256 * 55
for (M, N, K) in [(Md, 2304, 768), (Md, 768, 768), (Md, 3072, 768), (Md, 4096, 768), (Me, 2304, 768), (Me, 3072, 768), (Md, 768, 3072)]:
    out = []
    for st in (0, 4, 8, 12, 16, 24):
        lib.hct_debug_set_gemm_stagger(st)
        tf, us = run("nt", M, N, K, 4)
        out.append(f"st={st}: {tf:6.1f}TF")
    lib.hct_debug_set_gemm_stagger(-1)
    tf, us = run("nt", M, N, K, 4)
    t2, u2 = run("nt", M, N, K, 256)
    print(f"M={M} N={N} K={K}: w4 " + " | ".join(out) + f" | auto: {tf:6.1f}TF {us:6.1f}us || 256^2: {t2:6.1f}TF {u2:6.1f}us")
