"""DINO self-distillation pieces on the HIP path (BASELINE config #5; reference engine_pretrain_dino.py:14-130).

Built: `DINOLoss` (src/losses/losses.py:46-102: same constructor, `forward(student_output, teacher_output, epoch)`, `center`
buffer and `update_center`, fused loss + gradient kernel over the K prototypes), `update_momentum_encoder`
(src/utils/misc.py:386-397) on the models' flat fp32 parameter buffers, and the cosine weight-decay / momentum schedules
(src/utils/wd_sched.py:3-23).  No CPU fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from ._lib import HCT_BF16, HCT_F32, HctError


def wd_cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """Per-iteration schedule: optional linear warm-up, then half a cosine from base_value to final_value."""
    n_warm = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, n_warm) if warmup_epochs > 0 else np.array([])
    t = np.arange(epochs * niter_per_ep - n_warm)
    body = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * t / len(t)))
    out = np.concatenate((warm, body))
    if len(out) != epochs * niter_per_ep:
        raise AssertionError("schedule length does not match epochs * iterations per epoch")
    return out


def get_wd_scheduler(config, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    return wd_cosine_scheduler(config.TRAIN.WEIGHT_DECAY, config.TRAIN.WEIGHT_DECAY_END, config.TRAIN.MAX_EPOCHS, niter_per_ep,
                               warmup_epochs, start_warmup_value)


def _st():
    return torch.cuda.current_stream().cuda_stream


class _DinoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, teacher, center, ncrops, student_temp, teacher_temp, center_sum):
        lib = _lib.load()
        if not (student.is_cuda and teacher.is_cuda):
            raise HctError("DINOLoss (HIP) needs GPU tensors; there is no CPU fallback")
        if student.dtype not in (torch.float32, torch.bfloat16) or teacher.dtype != student.dtype:
            raise HctError("DINOLoss (HIP): student / teacher logits must both be fp32 or both bf16")
        student, teacher = student.contiguous(), teacher.contiguous()
        K = student.shape[1]
        B = teacher.shape[0] // 2
        if student.shape[0] != ncrops * B or teacher.shape[0] != 2 * B or teacher.shape[1] != K:
            raise HctError(f"DINOLoss: shapes student {tuple(student.shape)} / teacher {tuple(teacher.shape)} do not match {ncrops} crops")
        dt = HCT_BF16 if student.dtype == torch.bfloat16 else HCT_F32
        ws = torch.empty(lib.hct_dino_loss_workspace_bytes(ncrops, B, K), dtype=torch.uint8, device=student.device)
        loss = torch.empty(1, dtype=torch.float32, device=student.device)
        dstudent = torch.empty_like(student) if ctx.needs_input_grad[0] else None
        _lib.check(lib.hct_dino_loss(student.data_ptr(), teacher.data_ptr(), dt, ncrops, B, K, center.data_ptr(), float(student_temp),
                                     float(teacher_temp), loss.data_ptr(), _lib.ptr(dstudent), None, _lib.ptr(center_sum), ws.data_ptr(),
                                     ws.numel(), _st()), "hct_dino_loss")
        ctx.dstudent = dstudent
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.dstudent is None:
            raise HctError("DINOLoss: the forward ran without gradient storage")
        return ctx.dstudent * g.to(ctx.dstudent.dtype), None, None, None, None, None, None


class DINOLoss(nn.Module):
    """Same interface as the reference's DINOLoss (losses.py:46-102)."""

    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs, student_temp=0.1,
                 center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
                                                     np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))

    def forward(self, student_output, teacher_output, epoch):
        temp = float(self.teacher_temp_schedule[epoch])
        csum = torch.empty(self.center.shape[1], dtype=torch.float32, device=self.center.device)
        # grad mode is read by the Function's caller side: inside Function.forward it is always off
        loss = _DinoLossFn.apply(student_output, teacher_output.detach(), self.center, self.ncrops, self.student_temp, temp, csum)
        self._apply_center(csum, teacher_output.shape[0])
        return loss

    @torch.no_grad()
    def _apply_center(self, csum, n_rows):
        world = 1
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(csum)
            world = dist.get_world_size()
        _lib.check(_lib.load().hct_dino_center_update(self.center.data_ptr(), csum.data_ptr(), csum.numel(), float(self.center_momentum),
                                                      float(n_rows * world), _st()), "hct_dino_center_update")

    @torch.no_grad()
    def update_center(self, teacher_output):
        """losses.py:93-102 as a stand-alone call (forward() already does it from the loss kernel's column sums)."""
        csum = teacher_output.float().sum(dim=0).contiguous()
        self._apply_center(csum, teacher_output.shape[0])


@torch.no_grad()
def update_momentum_encoder(model, momentum_model, m: float) -> None:
    """misc.py:386-397 for flat-buffer HIP models: one fused launch over the whole parameter buffer (k = k*m + (1-m)*q)."""
    q, k = getattr(model, "_flat", None), getattr(momentum_model, "_flat", None)
    if q is None or k is None or q.numel() != k.numel() or not q.is_cuda:
        raise HctError("update_momentum_encoder (HIP) needs two flat-buffer HIP models of the same architecture on the GPU")
    _lib.check(_lib.load().hct_ema_update(k.data_ptr(), q.data_ptr(), k.numel(), float(m), _st()), "hct_ema_update")
    if hasattr(momentum_model, "mark_weights_updated"):
        momentum_model.mark_weights_updated()


@torch.no_grad()
def ema_update_(k: torch.Tensor, q: torch.Tensor, m: float) -> None:
    """The same update on two arbitrary contiguous fp32 GPU tensors (numel % 4 == 0)."""
    _lib.check(_lib.load().hct_ema_update(k.data_ptr(), q.data_ptr(), k.numel(), float(m), _st()), "hct_ema_update")


class DinoOptimizer:
    """AdamW over the student's two flat parameter buffers (backbone plan + projection head): two fused HipAdamW launches that
    share the learning rate and weight decay of `param_groups[0]` (what the LR scheduler and the weight-decay schedule write to).
    The reference builds one torch AdamW over MultiCropWrapper.parameters() (main_pretrain_dino.py:219); the arithmetic per
    parameter is the same, and `state_dict()` has the reference's flat layout (one AdamW state dict over backbone + head parameters)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8):
        from .optim import HipAdamW
        m = model.module if hasattr(model, "module") else model
        self.primary = HipAdamW(m.backbone, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.secondary = HipAdamW(m.head, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    @property
    def param_groups(self):
        return self.primary.param_groups

    def zero_grad(self, set_to_none: bool = True):
        self.primary.zero_grad(set_to_none=set_to_none)
        self.secondary.zero_grad(set_to_none=set_to_none)

    def step(self):
        for k in ("lr", "weight_decay", "betas", "eps"):
            self.secondary.param_groups[0][k] = self.primary.param_groups[0][k]
        self.primary.step()
        self.secondary.step()

    # The checkpoint's "optimizer" entry has the reference's layout: ONE torch AdamW state dict over MultiCropWrapper.parameters()
    # (main_pretrain_dino.py:219; misc.py:55-69 loads it back), i.e. state indices 0 .. nb-1 = backbone parameters, nb .. = head
    # parameters, one param group.  The two fused optimizers' dicts are merged / split at nb.
    def _nb(self) -> int:
        return len(self.primary.param_groups[0]["params"])

    def state_dict(self):
        a, b = self.primary.state_dict(), self.secondary.state_dict()
        nb = len(a["param_groups"][0]["params"])
        state = dict(a["state"])
        state.update({nb + i: v for i, v in b["state"].items()})
        group = dict(a["param_groups"][0])
        group["params"] = list(range(nb + len(b["param_groups"][0]["params"])))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "backbone" in sd and "head" in sd:  # checkpoints written before the flat layout
            self.primary.load_state_dict(sd["backbone"])
            self.secondary.load_state_dict(sd["head"])
            return
        nb, nh = self._nb(), len(self.secondary.param_groups[0]["params"])
        group = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(group["params"]) != nb + nh:
            raise ValueError(f"optimizer state of {sum(len(g['params']) for g in sd['param_groups'])} parameters in {len(sd['param_groups'])} "
                             f"group(s) does not match backbone ({nb}) + head ({nh}) in one group")
        ga, gb = dict(group), dict(group)
        ga["params"], gb["params"] = list(range(nb)), list(range(nh))
        self.primary.load_state_dict({"state": {i: v for i, v in sd["state"].items() if i < nb}, "param_groups": [ga]})
        self.secondary.load_state_dict({"state": {i - nb: v for i, v in sd["state"].items() if i >= nb}, "param_groups": [gb]})


class DinoDataParallel(nn.Module):
    """Data-parallel wrapper of a MultiCropWrapper: the backbone's flat gradient is all-reduced in buckets while its staged backward
    is still running (ddp.DistributedDataParallel), the head's (8 % of the parameters) in one all-reduce that the engine issues
    right after `loss.backward()` (`reduce_head_gradients`).  Rank 0's parameters are broadcast at construction."""

    def __init__(self, module, device_ids=None, broadcast_buffers: bool = False, find_unused_parameters: bool = False, bucket_cap_mb: float = 64.0):
        super().__init__()
        from .ddp import DistributedDataParallel
        self.module = module
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._backbone_ddp = DistributedDataParallel(module.backbone, bucket_cap_mb=bucket_cap_mb)
        if self.world_size > 1:
            dist.broadcast(module.head._flat, src=0)
            module.head.mark_weights_updated()

    def forward(self, x):
        return self.module(x)

    def reduce_head_gradients(self) -> None:
        if self.world_size > 1:
            g = self.module.head._flat_grad
            dist.all_reduce(g)
            g.div_(self.world_size)


class SyntheticCrops:
    """`n_batches` pre-generated multi-crop batches on the device: each a list of `n_crops` tensors [B, C, S, S, S] in [0, 1) (the
    reference resizes global and local crops to one size, transforms.py:75-97, so one backbone pass serves all crops)."""

    def __init__(self, n_batches, batch_size, n_crops, in_chans, size, device, seed=0):
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.batches = [[torch.rand(batch_size, in_chans, size, size, size, device=device, generator=gen) for _ in range(n_crops)]
                        for _ in range(n_batches)]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)
