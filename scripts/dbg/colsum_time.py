"""hct_colsum on the DINO qkv-bias shape (41 360 x 2 304 bf16) and the MAE decoder shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from headct_foundation_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
for rows, cols in ((41360, 2304), (55552, 3072), (14080, 768)):
    x = torch.randn(rows, cols, device=dev).bfloat16(); out = torch.empty(cols, device=dev)
    ws = torch.empty(lib.hct_colsum_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev)
    call = lambda: lib.hct_colsum(x.data_ptr(), 1, rows, cols, cols, out.data_ptr(), ws.data_ptr(), ws.numel(), st)
    assert call() == 0
    torch.cuda.synchronize()
    want = x.float().sum(0)
    err = ((out - want).norm() / want.norm()).item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"colsum {rows} x {cols}: {us:6.1f} us ({rows * cols * 2 / us / 1e6:.2f} TB/s), rel err {err:.1e}", flush=True)
