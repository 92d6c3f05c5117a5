"""Fused per-parameter gradient clip + AdamW on the flat parameter buffer (HIP).

Reference semantics:
  * clip_gradients  -- src/utils/misc.py:374-383 (PER-TENSOR L2 clip, coef = clip/(norm+1e-6) applied iff < 1)
  * get_optimizer   -- src/utils/optimizers.py:344-360 (torch.optim.AdamW, one param group, weight decay on
                       every parameter, eps 1e-8)
`HipAdamW` is a torch.optim.Optimizer whose state_dict()/load_state_dict() are interchangeable with
torch.optim.AdamW's (state = {index: {step, exp_avg, exp_avg_sq}}), so reference checkpoints resume.
"""
from __future__ import annotations

from itertools import chain
from typing import List, Optional

import torch

from . import _lib
from ._lib import HctError
from .mae import FlatPlanModule, MaskedAutoencoderViT  # noqa: F401 (MaskedAutoencoderViT re-exported for callers)


def _is_flat(m) -> bool:
    """A model whose parameters / gradients live in flat fp32 buffers laid out in 1024-element units (MAE, ViT backbone, DINO head)."""
    return isinstance(m, FlatPlanModule) or all(hasattr(m, a) for a in ("_flat", "_flat_grad", "_layout", "flat_segments"))


def unwrap(model):
    """Strip DistributedDataParallel-style wrappers (`.module`)."""
    while hasattr(model, "module") and not _is_flat(model):
        model = model.module
    return model


class _FlatState:
    """Device-side bookkeeping shared by clip and AdamW for one model."""

    def __init__(self, model):
        self.model = model
        self.flat_id = None
        self.refresh()

    def refresh(self):
        m = self.model
        dev = m._flat.device
        names, seg = m.flat_segments()
        self.names = names
        self.nseg = len(names)
        self.total = seg[-1]
        self.seg_off = torch.tensor(seg, dtype=torch.int64, device=dev)
        named = dict(m.named_parameters())
        self.skip = torch.tensor([0 if named[n].requires_grad else 1 for n in names], dtype=torch.uint8, device=dev)
        self.norms = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
        self.coef = torch.ones(self.nseg, dtype=torch.float32, device=dev)
        lib = _lib.load()
        self.ws = torch.empty(max(16, lib.hct_grad_norms_workspace_bytes(self.total)), dtype=torch.uint8, device=dev)
        self.flat_id = m._flat.data_ptr()
        self.coef_pending = False

    def ensure(self):
        if self.flat_id != self.model._flat.data_ptr():
            self.refresh()


def _state_for(model) -> _FlatState:
    st = getattr(model, "_flat_state", None)
    if st is None:
        st = _FlatState(model)
        model._flat_state = st
    st.ensure()
    return st


def clip_gradients(model, clip: float, defer_to_optimizer: Optional[bool] = None):
    """Per-parameter gradient clipping (src/utils/misc.py:374-383) in one pass, without host syncs.

    Returns the per-parameter L2 norms as a DEVICE tensor in `named_parameters()` order of the parameters
    that have a gradient (the reference returns a Python list after ~250 `.item()` syncs; call `.tolist()`
    on the result if you need that).  When the model is driven by `HipAdamW` the scaling itself is folded
    into the optimizer kernel (the clipped gradient is still written back to `.grad` there); otherwise the
    gradients are scaled in place right here.
    """
    m = unwrap(model)
    if not _is_flat(m):
        raise HctError("clip_gradients (HIP) expects a flat-buffer HIP model (MaskedAutoencoderViT, ViTBackbone, DINOHead)")
    if not m._flat.is_cuda:
        raise HctError("clip_gradients (HIP) needs the model on a GPU; there is no CPU fallback")
    st = _state_for(m)
    lib = _lib.load()
    defer = m._managed_updates if defer_to_optimizer is None else defer_to_optimizer
    _lib.check(lib.hct_grad_norms(m._flat_grad.data_ptr(), st.seg_off.data_ptr(), st.nseg, st.total, float(clip),
                                  0 if defer else 1, st.norms.data_ptr(), st.coef.data_ptr(), st.ws.data_ptr(),
                                  st.ws.numel(), _lib.stream_ptr()), "hct_grad_norms")
    st.coef_pending = bool(defer)
    # norms in named_parameters() order of the parameters that have a gradient; the gather index lives on the device
    # and is rebuilt only when the set changes (building it every call would be a blocking host->device copy, i.e. a
    # full pipeline drain per step)
    key = tuple(p.grad is not None for p in m.parameters())
    if getattr(st, "norm_key", None) != key:
        order = {n: i for i, n in enumerate(st.names)}
        idx = [order[n] for n, p in m.named_parameters() if p.grad is not None]
        st.norm_idx = torch.tensor(idx, device=st.norms.device, dtype=torch.long)
        st.norm_key = key
    return st.norms[st.norm_idx]


class HipAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics, executed as one fused HIP kernel over the model's flat buffers."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        m = unwrap(model)
        if not _is_flat(m):
            raise HctError("HipAdamW expects a flat-buffer HIP model (MaskedAutoencoderViT, ViTBackbone, DINOHead)")
        self._model = m
        params = list(m.parameters())  # registration order == torch.optim.AdamW(model.parameters()) order
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(params, defaults)
        self._step_count_fused = 0
        self._step_tensor = torch.tensor(0.0)
        self._m = self._v = None
        m._managed_updates = True

    # flat moment buffers, exposed per-parameter through self.state for state_dict() compatibility
    def _ensure_state(self):
        m = self._model
        if self._m is not None and self._m.device == m._flat.device and self._m.numel() == m._flat.numel():
            return
        old = {id(p): self.state.get(p) for p in m.parameters()}
        self._m = torch.zeros_like(m._flat)
        self._v = torch.zeros_like(m._flat)
        named = dict(m.named_parameters())
        for name, off, numel, shape, rg, _ in m._layout:
            p = named[name]
            if not p.requires_grad:
                continue
            mv, vv = self._m[off:off + numel].view(shape), self._v[off:off + numel].view(shape)
            prev = old.get(id(p))
            if prev:
                mv.copy_(prev["exp_avg"]); vv.copy_(prev["exp_avg_sq"])
            self.state[p] = {"step": self._step_tensor, "exp_avg": mv, "exp_avg_sq": vv}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # re-home the loaded moments into the flat buffers
        steps = [float(s["step"]) for s in self.state.values() if "step" in s]
        self._step_count_fused = int(max(steps)) if steps else 0
        self._step_tensor = torch.tensor(float(self._step_count_fused))
        self._m = None
        self._ensure_state()

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        self._model._grad_overwrite = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self._model
        if not m._flat.is_cuda:
            raise HctError("HipAdamW.step needs the model on a GPU; there is no CPU fallback")
        self._ensure_state()
        st = _state_for(m)
        grp = self.param_groups[0]
        self._step_count_fused += 1
        lib = _lib.load()
        coef = st.coef.data_ptr() if st.coef_pending else None
        # parameters that received no gradient this step are skipped like torch does (p.grad is None): the skip mask follows the
        # set of gradient-less parameters and is re-uploaded only when that set changes (e.g. the DINO prototype layer, whose
        # gradients are cancelled during the first epochs)
        named = getattr(m, "_named_cache", None) or dict(m.named_parameters())
        key = tuple((not named[n].requires_grad) or named[n].grad is None for n in st.names)
        if getattr(st, "skip_key", None) != key:
            st.skip = torch.tensor([1 if k else 0 for k in key], dtype=torch.uint8, device=m._flat.device)
            st.skip_key = key
        _lib.check(lib.hct_adamw_step(
            m._flat.data_ptr(), m._flat_grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), st.seg_off.data_ptr(), coef,
            st.skip.data_ptr(), st.nseg, st.total, float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]),
            float(grp["eps"]), float(grp["weight_decay"]), self._step_count_fused,
            _lib.ptr(m._flat_bf16), _lib.stream_ptr()), "hct_adamw_step")
        st.coef_pending = False
        self._step_tensor.fill_(float(self._step_count_fused))  # one shared CPU scalar referenced by every state entry
        m.mark_weights_updated(plain_bf16_fresh=m._flat_bf16 is not None)
        return loss


def get_optimizer(config, lr, models):
    """src/utils/optimizers.py:344-378: the MAE path uses AdamW; other optimizers are outside the hot path."""
    if config.TRAIN.OPTIMIZER != 'AdamW':
        raise NotImplementedError("Unknown optimizer for the HIP MAE path: {} (only AdamW is on the hot path)".format(config.TRAIN.OPTIMIZER))
    if len(models) != 1:
        raise HctError("get_optimizer (HIP) expects exactly one model")
    return HipAdamW(models[0], lr=lr, weight_decay=config.TRAIN.WEIGHT_DECAY, betas=(config.TRAIN.BETA1, config.TRAIN.BETA2))
