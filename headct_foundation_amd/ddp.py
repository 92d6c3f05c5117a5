"""Data-parallel wrapper: bucketed gradient all-reduce over RCCL, overlapped with the staged backward.

Replaces `torch.nn.parallel.DistributedDataParallel(model, device_ids=[device], broadcast_buffers=False,
find_unused_parameters=True)` at main_pretrain_mae.py:139 for the HIP model:
  * construction broadcasts the flat fp32 parameter buffer from rank 0 (DDP ctor semantics);
  * the native backward runs in stages that each compute one contiguous range of the flat gradient buffer,
    from its end towards its start (csrc/mae_plan.hip); the weight gradients of a few stages run together in one
    grouped launch, after which the ranges of those stages are final (`wgrad_group_blocks`, default 4 block
    stages per launch here, against once per decoder / encoder without data parallelism).  Every range that
    became final is appended to the open bucket; once a bucket holds >= bucket_cap_mb it is all-reduced asynchronously
    (`torch.distributed` backend "nccl" == RCCL; the collective is ordered after the producing kernels on the
    compute stream and runs on the process group's own stream, so it overlaps the remaining backward);
  * the mean over ranks costs nothing: the backward is seeded with dLoss/world_size, so the SUM all-reduce
    already yields the average;
  * one process per GPU, no data-path collective other than this one (pure data parallelism).
`state_dict()` keys carry the `module.` prefix exactly like the reference's checkpoints (misc.py:38).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, broadcast_buffers: bool = False,
                 find_unused_parameters: bool = False, bucket_cap_mb: float = 64.0, process_group=None, grad_dtype: str = "fp32",
                 force_collectives: bool = False):
        super().__init__()
        for attr in ("_flat", "_flat_grad", "_bucket_hook", "_post_backward_hook"):
            if not hasattr(module, attr):
                raise TypeError("DistributedDataParallel (HIP) wraps the HIP MaskedAutoencoderViT (flat-buffer model)")
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives (rehearsal on a one-GPU box): with an initialised process group of ONE rank the wrapper still runs its whole
        # data path -- grouped weight gradients every few stages, buckets, asynchronous all-reduce on the backend's stream, the CU
        # reserve, the final wait -- so that the RCCL calls themselves execute on hardware; the sums are those of one rank.
        self._active = self.world_size > 1 or (force_collectives and dist.is_initialized())
        self.bucket_bytes = int(bucket_cap_mb * (1 << 20))
        # "bf16": a bucket crosses the links as bfloat16 (298 MB per step instead of 595 MB for ViT-B) -- cast into a staging
        # buffer, SUM all-reduce, cast back into the fp32 gradient.  Each rank's gradient is rounded to 8 significant bits before
        # the sum (the sum itself is carried out in bf16 by the collective): an accuracy / bandwidth trade that is OFF by default.
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        self._staged: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self._open: Optional[Tuple[int, int]] = None  # [begin, end) of the bucket being filled
        self._works: List = []
        self.launched: List[Tuple[int, int]] = []      # ranges reduced during the last backward (for tests)
        if self._active:
            dist.broadcast(module._flat, src=0, group=process_group)  # DDP ctor: rank 0's parameters win
            if hasattr(module, "mark_weights_updated"):
                module.mark_weights_updated()
        # Persistent GEMM workgroups fill every CU's register file and LDS: while a gradient all-reduce is in flight a few
        # CUs are left free for the RCCL kernels, otherwise the GEMM workgroups that find no CU start only when the others
        # have finished (HCT_CU_RESERVE overrides the count; 0 disables).  The reserve is switched on with the first bucket
        # of a backward and off again when the collectives have been waited for, so the forward pass and the part of
        # the backward before the first bucket keep all CUs.
        self._reserve = 0
        self._set_reserve = None
        if self._active and getattr(module, "_flat", None) is not None and module._flat.is_cuda:
            import os
            from . import _lib
            self._reserve = int(os.environ.get("HCT_CU_RESERVE", "16"))
            self._set_reserve = _lib.load().hct_set_cu_reserve
        module._grad_prescale = 1.0 / self.world_size
        if self._active and getattr(module, "wgrad_group_blocks", 0) is None:
            import os
            module.wgrad_group_blocks = int(os.environ.get("HCT_WGRAD_GROUP_BLOCKS", "4"))
        module._bucket_hook = self._on_stage
        module._post_backward_hook = self._finish

    # called by the model after backward stage `stage` has been enqueued when a gradient range became final: [begin, end)
    def _on_stage(self, stage: int, begin: int, end: int) -> None:
        if not self._active:
            return
        if end == self.module._flat_grad.numel():  # first range of a backward
            self.launched = []
        if self._open is None:
            self._open = (begin, end)
        else:
            ob, oe = self._open
            if end != ob:  # not adjacent (should not happen): flush and restart
                self._launch(ob, oe)
                self._open = (begin, end)
            else:
                self._open = (begin, oe)
        ob, oe = self._open
        if (oe - ob) * 4 >= self.bucket_bytes:
            self._launch(ob, oe)
            self._open = None

    def _launch(self, begin: int, end: int) -> None:
        if self._set_reserve is not None and not self._works and self._reserve > 0:
            self._set_reserve(self._reserve)  # GEMMs enqueued from here on run beside the collective
        view = self.module._flat_grad[begin:end]
        if self.grad_dtype == "bf16":
            stage = view.to(torch.bfloat16)
            self._staged.append((stage, view))
            view = stage
        self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True))
        self.launched.append((begin, end))

    def _finish(self) -> None:
        if not self._active:
            return
        if self._open is not None:
            self._launch(*self._open)
            self._open = None
        for w in self._works:
            w.wait()  # compute stream waits for the collective (no host block on NCCL/RCCL)
        self._works = []
        for stage, view in self._staged:  # bf16 buckets: back into the fp32 gradient
            view.copy_(stage)
        self._staged = []
        if self._set_reserve is not None and self._reserve > 0:
            self._set_reserve(0)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
