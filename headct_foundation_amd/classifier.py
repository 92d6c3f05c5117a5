"""Classification heads over ViT features on the HIP kernels (forward only, eval-mode arithmetic).

Mirrors of `LinearClassifier` and `AttentionClassifier` (src/models/classifier.py:7-99): same constructor arguments,
parameter and buffer names (`bn.running_mean`, `bn.running_var`, `bn.num_batches_tracked`, `linear.*`; `bn1`, `bn2`, `wkv.*`,
`cls_token`), so a head trained with the reference loads with `load_state_dict`.  The BatchNorm layers use their running
statistics - what the reference computes after `.eval()`.  `LinearClassifier` also runs in training mode on frozen
(detached) features - linear probing, engine_downstream.py:70-117 with TRAIN.LOCK: batch statistics + running-statistics
update, logits, `cross_entropy` (nn.CrossEntropyLoss()) and the gradients of `linear.weight` / `linear.bias`, all on the HIP
kernels through `torch.autograd.Function`s.  Not built: gradients with respect to the features (fine-tuning the backbone)
and the training mode of `AttentionClassifier`, whose `forward` refuses a module left in training mode.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .mae import _Affine, _Holder


class _BatchNormStats(_Holder):
    """Buffers of nn.BatchNorm1d(dim, affine=False): running_mean, running_var, num_batches_tracked."""

    def __init__(self, dim: int):
        super().__init__()
        self.eps = 1e-6
        self.register_buffer("running_mean", torch.zeros(dim))
        self.register_buffer("running_var", torch.ones(dim))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


def _init_linear(m: _Affine) -> None:  # nn.Linear defaults
    nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
    if m.bias is not None:
        bound = 1 / math.sqrt(m.weight.shape[1])
        nn.init.uniform_(m.bias, -bound, bound)


def _require_eval_cuda(mod: nn.Module, x: torch.Tensor, what: str, need_eval: bool = True) -> None:
    if mod.training and need_eval:
        raise _lib.HctError(f"{what} (HIP) computes the eval-mode forward (BatchNorm running statistics): call .eval() first; "
                            "training-mode batch statistics are not built")
    if not x.is_cuda or not mod.linear.weight.is_cuda:
        raise _lib.HctError(f"{what} (HIP) runs on the GPU: move the module and the input to 'cuda' (no CPU fallback exists)")


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


class _LinearProbeFn(torch.autograd.Function):
    """Training-mode LinearClassifier on detached features: BatchNorm1d batch statistics (+ running update, momentum 0.1),
    Linear; backward gives the parameter gradients only."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        lib = _lib.load()
        B, D = x.shape
        ncls = weight.shape[0]
        with torch.cuda.device(x.device):
            st = _stream(x.device)
            mean = torch.empty(D, dtype=torch.float32, device=x.device)
            var = torch.empty(D, dtype=torch.float32, device=x.device)
            _lib.check(lib.hct_batchnorm_stats(x.data_ptr(), B, D, 0.1, mean.data_ptr(), var.data_ptr(), bn.running_mean.data_ptr(),
                                               bn.running_var.data_ptr(), st), "hct_batchnorm_stats")
            bn.num_batches_tracked += 1
            out = torch.empty(B, ncls, dtype=torch.float32, device=x.device)
            _lib.check(lib.hct_head_linear(x.data_ptr(), D, 1, mean.data_ptr(), var.data_ptr(), bn.eps, weight.data_ptr(), bias.data_ptr(),
                                           _lib.HCT_ACT_NONE, out.data_ptr(), B, D, ncls, st), "hct_head_linear")
        ctx.save_for_backward(x, mean, var)
        ctx.eps, ctx.ncls = bn.eps, ncls
        return out

    @staticmethod
    def backward(ctx, dlogits):
        x, mean, var = ctx.saved_tensors
        lib = _lib.load()
        B, D = x.shape
        dlogits = dlogits.to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            dW = torch.empty(ctx.ncls, D, dtype=torch.float32, device=x.device)
            db = torch.empty(ctx.ncls, dtype=torch.float32, device=x.device)
            _lib.check(lib.hct_head_linear_wgrad(x.data_ptr(), mean.data_ptr(), var.data_ptr(), ctx.eps, dlogits.data_ptr(), B, D, ctx.ncls,
                                                 dW.data_ptr(), db.data_ptr(), _stream(x.device)), "hct_head_linear_wgrad")
        return None, dW, db, None


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        lib = _lib.load()
        B, ncls = logits.shape
        with torch.cuda.device(logits.device):
            loss = torch.empty((), dtype=torch.float32, device=logits.device)
            _lib.check(lib.hct_softmax_xent(logits.data_ptr(), target.data_ptr(), B, ncls, None, loss.data_ptr(), None, _stream(logits.device)),
                       "hct_softmax_xent")
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        lib = _lib.load()
        B, ncls = logits.shape
        dloss = dloss.to(torch.float32).contiguous()
        with torch.cuda.device(logits.device):
            dlogits = torch.empty_like(logits)
            _lib.check(lib.hct_softmax_xent(logits.data_ptr(), target.data_ptr(), B, ncls, dloss.data_ptr(), None, dlogits.data_ptr(),
                                            _stream(logits.device)), "hct_softmax_xent")
        return dlogits, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss() of main_downstream.py:214 (mean reduction, class-index targets, no weights) on the HIP kernel.
    A target outside [0, num_classes) makes the loss NaN (the reference's loop stops on a non-finite loss)."""
    if not logits.is_cuda or logits.dim() != 2 or target.shape != logits.shape[:1]:
        raise _lib.HctError("cross_entropy (HIP): logits [B, C] on 'cuda' and class-index targets [B] expected (no CPU fallback exists)")
    return _CrossEntropyFn.apply(logits.to(torch.float32).contiguous(), target.to(device=logits.device, dtype=torch.int64).contiguous())


class LinearClassifier(nn.Module):
    """classifier.py:7-33: BatchNorm1d(dim, affine=False, eps=1e-6) -> Linear(dim, num_classes) on [B, dim] features."""

    def __init__(self, dim: int, num_classes: int):
        super().__init__()
        self.bn = _BatchNormStats(dim)
        self.linear = _Affine(num_classes, dim, bias_shape=(num_classes,))
        with torch.no_grad():
            _init_linear(self.linear)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_eval_cuda(self, x, "LinearClassifier", need_eval=False)
        ncls, dim = self.linear.weight.shape
        if x.dim() != 2 or x.shape[1] != dim:
            raise _lib.HctError(f"input shape {tuple(x.shape)} != (B, {dim})")
        if self.training:  # linear probing on frozen features
            if x.requires_grad:
                raise _lib.HctError("LinearClassifier (HIP) trains on detached features (TRAIN.LOCK): the gradient with respect to "
                                    "the features is not built")
            return _LinearProbeFn.apply(x.to(torch.float32).contiguous(), self.linear.weight, self.linear.bias, self.bn)
        lib = _lib.load()
        with torch.no_grad(), torch.cuda.device(x.device):
            st = torch.cuda.current_stream().cuda_stream
            x = x.to(torch.float32).contiguous()
            out = torch.empty(x.shape[0], ncls, dtype=torch.float32, device=x.device)
            _lib.check(lib.hct_head_linear(x.data_ptr(), dim, 1, self.bn.running_mean.data_ptr(), self.bn.running_var.data_ptr(), self.bn.eps,
                                           self.linear.weight.data_ptr(), self.linear.bias.data_ptr(), _lib.HCT_ACT_NONE, out.data_ptr(),
                                           x.shape[0], dim, ncls, st), "hct_head_linear")
        return out


class AttentionClassifier(nn.Module):
    """classifier.py:35-99: `num_queries` learnt query tokens attend over the (batch-normalised) token features through a
    key/value projection `wkv`; the attended vectors are batch-normalised, averaged over the queries and classified."""

    def __init__(self, dim: int, num_classes: int, num_heads: int = 12, qkv_bias: bool = False, qk_scale: Optional[float] = None,
                 num_queries: int = 1, compute_dtype: str = "fp32"):
        super().__init__()
        if dim % num_heads:
            raise ValueError("dim should be divisible by num_heads.")
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        self.num_heads, self.num_queries, self.compute_dtype = num_heads, num_queries, compute_dtype
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.bn1 = _BatchNormStats(dim)
        self.bn2 = _BatchNormStats(dim)
        self.wkv = _Affine(dim * 2, dim, bias_shape=(dim * 2,) if qkv_bias else None)
        self.linear = _Affine(num_classes, dim, bias_shape=(num_classes,))
        self.cls_token = nn.Parameter(torch.zeros(1, num_queries, dim))
        with torch.no_grad():
            _init_linear(self.wkv)
            _init_linear(self.linear)
            nn.init.trunc_normal_(self.cls_token, std=0.02)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_eval_cuda(self, x, "AttentionClassifier")
        ncls, dim = self.linear.weight.shape
        if x.dim() != 3 or x.shape[2] != dim:
            raise _lib.HctError(f"input shape {tuple(x.shape)} != (B, N, {dim})")
        B, N, _ = x.shape
        H, Q, dh = self.num_heads, self.num_queries, dim // self.num_heads
        lib = _lib.load()
        dev = x.device
        bf = self.compute_dtype == "bf16"
        tdt, dt = (torch.bfloat16, _lib.HCT_BF16) if bf else (torch.float32, _lib.HCT_F32)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            x = x.to(torch.float32).contiguous()
            xn = torch.empty(B * N, dim, dtype=tdt, device=dev)  # bn1, classifier.py:89
            _lib.check(lib.hct_channel_norm(x.data_ptr(), self.bn1.running_mean.data_ptr(), self.bn1.running_var.data_ptr(), self.bn1.eps,
                                            xn.data_ptr(), dt, B * N, dim, st), "hct_channel_norm")
            w = self.wkv.weight.detach()
            if bf:
                wb = torch.empty(w.shape, dtype=torch.bfloat16, device=dev)
                _lib.check(lib.hct_cast(w.data_ptr(), _lib.HCT_F32, wb.data_ptr(), _lib.HCT_BF16, w.numel(), st), "hct_cast")
                w = wb
            kv = torch.empty(B * N, 2 * dim, dtype=tdt, device=dev)  # wkv, classifier.py:90: [B, N, 2, H, dh] as it lies
            g = _lib.GemmArgs()
            g.M, g.N, g.K = B * N, 2 * dim, dim
            g.A, g.a_dtype, g.lda, g.transA = xn.data_ptr(), dt, dim, 0
            g.B, g.b_dtype, g.ldb, g.transB = w.data_ptr(), dt, dim, 1
            g.C, g.c_dtype, g.ldc = kv.data_ptr(), dt, 2 * dim
            if self.wkv.bias is not None:
                g.bias = self.wkv.bias.data_ptr()
            g.alpha = 1.0
            _lib.check(lib.hct_gemm(C.byref(g), None, 0, st), "hct_gemm")
            att = torch.empty(B, H, Q, dh, dtype=torch.float32, device=dev)  # classifier.py:86, :93
            _lib.check(lib.hct_query_attention(self.cls_token.data_ptr(), Q, kv.data_ptr(), dt, B, N, H, dh, self.scale * dh ** -0.5,
                                               att.data_ptr(), st), "hct_query_attention")
            out = torch.empty(B, ncls, dtype=torch.float32, device=dev)  # reshape(B, Q, C) -> bn2 -> mean -> linear, :95-99
            _lib.check(lib.hct_head_linear(att.data_ptr(), Q * dim, Q, self.bn2.running_mean.data_ptr(), self.bn2.running_var.data_ptr(),
                                           self.bn2.eps, self.linear.weight.data_ptr(), self.linear.bias.data_ptr(), _lib.HCT_ACT_NONE,
                                           out.data_ptr(), B, dim, ncls, st), "hct_head_linear")
        return out
