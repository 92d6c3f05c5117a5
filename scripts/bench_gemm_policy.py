import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run, lib
Md, Me = 256 * 217, 256 * 55
for (M, N, K) in [(Md, 2304, 768), (Md, 768, 768), (Md, 3072, 768), (Md, 768, 3072), (Me, 2304, 768)]:
    out = []
    for pol, nm in ((0, "plain"), (1, "nt"), (2, "sc1"), (3, "sc0sc1")):
        lib.hct_debug_set_gemm_stagger(-100 - pol)
        tf, us = run("nt", M, N, K, 0)
        out.append(f"{nm}: {tf:6.1f}TF {us:6.1f}us")
    lib.hct_debug_set_gemm_stagger(-100)
    print(f"M={M} N={N} K={K}: " + " | ".join(out))
