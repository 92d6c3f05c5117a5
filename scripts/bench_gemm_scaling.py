import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import run
Md = 256 * 217
for K in (768,):
    r = [run("nt", Md, 2304, K, v) for v in (256, 1256, 2256, 3256, 4256)]
    print(f"N=2304 K={K:5d}: normal {r[0][0]:7.1f} TF {r[0][1]:8.1f} us | no-epilogue {r[1][0]:7.1f} TF {r[1][1]:8.1f} us | L2-resident operands {r[2][0]:7.1f} TF {r[2][1]:8.1f} us | dense stores {r[3][0]:7.1f} TF {r[3][1]:8.1f} us | staged+dense {r[4][1]:8.1f} us")
for M in (256 * 16, 256 * 32, 256 * 64, 256 * 217):
    r = run("nt", M, 2304, 768, 256)
    print(f"M={M:6d} N=2304 K=768: {r[0]:7.1f} TF {r[1]:8.1f} us  tiles={M//256*9}")
