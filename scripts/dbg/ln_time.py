"""LayerNorm forward / backward launch times on the step's two shapes (encoder 14 080 x 768, decoder 55 552 x 768), inputs rotated
over several buffers so that they do not simply sit in the Infinity Cache.  (Measured with a diagnostic build: two or four rows in
flight per wave 4 - 18 % slower, a grid cap of 4096 instead of 2048 workgroups 4 % faster on the decoder shape only: 5.1 - 5.3 TB/s.)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16

lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for rows, D, nbuf in [(256 * 55, 768, 8), (256 * 217, 768, 4)]:
    xs = [torch.randn(rows, D, device=dev) for _ in range(nbuf)]
    ys = [torch.empty(rows, D, dtype=torch.bfloat16, device=dev) for _ in range(nbuf)]
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)

    def fwd(i):
        _lib.check(lib.hct_layernorm_fwd(xs[i % nbuf].data_ptr(), g.data_ptr(), b.data_ptr(), rows, D, 1e-5, ys[i % nbuf].data_ptr(), HCT_BF16,
                                         mean.data_ptr(), rstd.data_ptr(), st), "ln fwd")

    for i in range(4):
        fwd(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        fwd(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    print(f"layernorm fwd rows={rows} D={D}: {us:.1f} us, {rows * D * 6 / us / 1e6:.2f} TB/s", flush=True)
