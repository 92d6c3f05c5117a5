"""DINO pre-training models on the HIP path (BASELINE config #5): the trainable plain ViT backbone, the projection head and the
multi-crop wrapper.

Reference contracts mirrored here (constructor arguments, parameter names / shapes / registration order, outputs):
  * `ViTBackbone`      -- src/models/vit.py:25-173 (`ViT`): every patch embedded, class token, register tokens, blocks, final
                          LayerNorm eps 1e-6; returns `(x, hidden_states_out)` with `x` = the normalised tokens [B, 1+R+L, D].
                          Forward AND backward run in the native plan (csrc/mae_plan.hip, encoder-only mode).
  * `DINOHead`         -- src/models/dino_head.py:7-41 (use_bn=False, the reference yaml's setting): 3-layer GELU MLP, L2
                          normalisation, weight-normalised prototype layer; composed from hct_gemm + the head kernels of csrc/dino.hip.
  * `MultiCropWrapper` -- src/utils/misc.py:447-484: crops of one resolution are concatenated, one backbone pass per resolution,
                          head on the class-token features; returns {'dino_output': logits}.
PyTorch carries tensors between the native calls and owns the autograd graph edges (two custom Functions); no arithmetic of
the path runs in torch ops.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import HCT_BF16, HCT_F32, HctError
from .mae import FlatPlanModule, _Affine, _Holder, _block, build_sincos_position_embedding

_POS = {"none": 0, "learnable": 1, "sincos": 2}


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _code(t: torch.Tensor) -> int:
    return HCT_BF16 if t.dtype == torch.bfloat16 else HCT_F32


# ================================================================================================
# backbone
# ================================================================================================
class _ViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, train, *xs):
        # xs: the batch as one tensor or as several equally shaped ones in batch order (crops that are not concatenated first)
        B = sum(int(t.shape[0]) for t in xs)
        plan = model._plan_for(B)
        st = _st()
        model._ensure_weights_fresh(plan, st)
        xdt = _lib.HCT_F16 if xs[0].dtype == torch.float16 else HCT_F32
        ptrs = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        _lib.check(plan.lib.hct_vit_forward_parts(plan.handle, ptrs, len(xs), xdt, st), "hct_vit_forward_parts")
        plan.serial += 1
        ctx.model, ctx.plan, ctx.serial, ctx.nx = model, plan, plan.serial, len(xs)
        lat = plan.activation("latent").view(B, model.num_tokens, model.hidden_size)
        return lat.clone()  # the workspace is reused by the next forward at this batch size

    @staticmethod
    def backward(ctx, dlat):
        model, plan = ctx.model, ctx.plan
        if plan.serial != ctx.serial:
            raise HctError("backward of a stale ViT forward: another forward at the same batch size has overwritten the activations")
        tdt = torch.bfloat16 if model._dt == HCT_BF16 else torch.float32
        d = dlat.to(tdt).contiguous()
        if model._grad_prescale != 1.0:
            d = d * model._grad_prescale  # data-parallel mean (ddp.py)
        st = _st()
        lib = plan.lib
        model._keep_alive = d
        model._run_staged_backward(plan, lambda s: lib.hct_vit_backward_stage(plan.handle, s, d.data_ptr() if s == 0 else None, st),
                                   "hct_vit_backward_stage")
        return (None, None, None) + (None,) * ctx.nx


class ViTBackbone(FlatPlanModule):
    """Plain ViT (reference `ViT`, src/models/vit.py) with a native forward and backward."""

    def __init__(self, in_chans: int, img_size, patch_size, hidden_size: int = 768, mlp_dim: int = 3072, num_layers: int = 12,
                 num_heads: int = 12, patch_embed: str = "conv", pos_embed: str = "learnable", classification: bool = False,
                 num_classes: int = 2, dropout_rate: float = 0.0, spatial_dims: int = 3, num_register_tokens: int = 0,
                 post_activation: str = "Tanh", qkv_bias: bool = False, lora: bool = False, norm_layer=nn.LayerNorm,
                 compute_dtype: str = "bf16"):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if lora or classification or spatial_dims != 3 or patch_embed != "conv" or dropout_rate != 0.0 or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("HIP ViTBackbone: lora=False, classification=False, 3-D conv patch embedding, dropout 0, nn.LayerNorm")
        if pos_embed not in _POS:
            raise ValueError(f"pos_embed type {pos_embed} not supported.")
        S = img_size if isinstance(img_size, int) else img_size[0]
        P = patch_size if isinstance(patch_size, int) else patch_size[0]
        if S % P:
            raise ValueError("patch_size should be divisible by img_size.")
        self.in_chans, self.img_size, self.patch_size, self.hidden_size = in_chans, S, P, hidden_size
        self.grid = S // P
        self.num_patches = self.grid ** 3
        self.len_keep = self.num_patches  # every patch is embedded
        self.num_register_tokens = num_register_tokens
        self.num_tokens = 1 + num_register_tokens + self.num_patches
        self.compute_dtype = compute_dtype
        D = hidden_size
        # registration order of vit.py:103-131 (state_dict: own parameters first, then patch_embedding, blocks, norm)
        self.patch_embedding = _Holder()
        self.patch_embedding.n_patches = self.num_patches
        self.patch_embedding.position_embeddings = nn.Parameter(torch.zeros(1, self.num_patches, D)) if pos_embed != "none" else None
        self.patch_embedding.patch_embeddings = _Affine(D, in_chans, P, P, P, bias_shape=(D,))
        self.blocks = nn.ModuleList([_block(D, mlp_dim, qkv_bias) for _ in range(num_layers)])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.norm = _Affine(D, bias_shape=(D,))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, D)) if num_register_tokens else None
        self._ccfg = _lib.MaeConfig(
            input_size=S, patch_size=P, in_chans=in_chans, mask_ratio=0.0, pos_embed=_POS[pos_embed], encoder_depth=num_layers,
            encoder_embed_dim=D, encoder_mlp_dim=mlp_dim, encoder_num_heads=num_heads, decoder_depth=0, decoder_embed_dim=D,
            decoder_mlp_dim=mlp_dim, decoder_num_heads=num_heads, norm_pix_loss=0, use_bias=int(bool(qkv_bias)), encoder_only=1,
            num_register_tokens=num_register_tokens, final_norm_eps=1e-6)
        self._dt = HCT_BF16 if compute_dtype == "bf16" else HCT_F32
        self._init_flat_state()
        with torch.no_grad():  # PatchEmbeddingBlock init (patch_embedding.py:112-130) + torch defaults + vit.py:139-142
            pe = self.patch_embedding
            if pos_embed == "learnable":
                nn.init.trunc_normal_(pe.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
            elif pos_embed == "sincos":
                pe.position_embeddings.copy_(build_sincos_position_embedding([self.grid] * 3, D, 3))
            lin = [pe.patch_embeddings]
            for b_ in self.blocks:
                lin += [b_.attn.qkv, b_.attn.proj, b_.mlp.linear1, b_.mlp.linear2]
                for ln in (b_.att_norm, b_.ffn_norm):
                    ln.weight.fill_(1.0)
                    ln.bias.zero_()
            self.norm.weight.fill_(1.0)
            self.norm.bias.zero_()
            for m in lin:
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    bound = 1 / math.sqrt(m.weight[0].numel())
                    nn.init.uniform_(m.bias, -bound, bound)
            nn.init.normal_(self.cls_token, std=1e-6)
            if self.register_tokens is not None:
                nn.init.normal_(self.register_tokens, std=1e-6)
        self._build_flat(torch.device("cpu"))

    def forward(self, x):
        """x: `[B, C, S, S, S]`, or a list / tuple of equally shaped tensors standing for their concatenation along the batch (the
        crops of one resolution, which MultiCropWrapper would otherwise copy into one tensor first)."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x]
        for t in xs:
            if not t.is_cuda:
                raise HctError("ViTBackbone (HIP) got a CPU tensor: this path has no CPU fallback")
            expect = (t.shape[0], self.in_chans, self.img_size, self.img_size, self.img_size)
            if tuple(t.shape) != expect:
                raise HctError(f"input shape {tuple(t.shape)} != {expect}")
        if any(t.shape[0] != xs[0].shape[0] or t.dtype != xs[0].dtype for t in xs):
            xs = [torch.cat(xs)]  # unequal parts: the copy after all
        xs = [t.contiguous() if t.dtype == torch.float16 else t.contiguous().float() for t in xs]
        if all(p.grad is None for p in self.parameters()):
            self._grad_overwrite = True
        out = _ViTFunction.apply(self.cls_token, self, torch.is_grad_enabled(), *xs)
        return out, []  # (normalised tokens, hidden_states_out): the per-block states are not materialised on this path


# ================================================================================================
# projection head
# ================================================================================================
class _FlatParams:
    """Flat fp32 parameter / gradient buffers in 1024-element units for a module without a native plan (the DINO head), with the
    attributes HipAdamW / clip_gradients use (`_flat`, `_flat_grad`, `_layout`, `flat_segments`, `mark_weights_updated`)."""

    def _build_flat(self, device) -> None:
        named = list(self.named_parameters())
        layout, off = [], 0
        for n, p in named:
            layout.append((n, off, p.numel(), tuple(p.shape), bool(p.requires_grad), -1))
            off += (p.numel() + 1023) // 1024 * 1024
        flat = torch.zeros(off, dtype=torch.float32, device=device)
        for (n, o, numel, shape, rg, _), (_, p) in zip(layout, named):
            flat[o:o + numel].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[o:o + numel].view(shape)
            p.grad = None
        self._layout, self._flat = layout, flat
        self._flat_grad = torch.zeros(off, dtype=torch.float32, device=device)
        self._flat_bf16 = None
        self._seg_names = [n for n, *_ in layout]
        self._seg_off_host = [o for _, o, *_ in layout] + [off]
        self._named_cache = dict(named)
        self._weights_version = getattr(self, "_weights_version", 0) + 1

    def flat_segments(self):
        return self._seg_names, self._seg_off_host

    def mark_weights_updated(self, plain_bf16_fresh: bool = False) -> None:
        self._weights_version += 1

    def _attach_grads(self) -> None:
        for n, o, numel, shape, rg, _ in self._layout:
            p = self._named_cache[n]
            if rg:
                p.grad = self._flat_grad[o:o + numel].view(shape)


def _gemm(lib, A, B, transA, transB, M, N, K, out, bias=None, act=0, aux=None, ws=None, alpha=1.0):
    g = _lib.GemmArgs()
    g.M, g.N, g.K = M, N, K
    g.A, g.a_dtype, g.lda, g.transA = A.data_ptr(), _code(A), A.stride(0), int(transA)
    g.B, g.b_dtype, g.ldb, g.transB = B.data_ptr(), _code(B), B.stride(0), int(transB)
    g.C, g.c_dtype, g.ldc = out.data_ptr(), _code(out), N
    if bias is not None:
        g.bias = bias.data_ptr()
    g.act = act
    if aux is not None:
        g.aux, g.aux_dtype, g.ldaux = aux.data_ptr(), _code(aux), N
    g.alpha = alpha
    need = lib.hct_gemm_workspace_bytes(C.byref(g))
    w = torch.empty(max(16, need), dtype=torch.uint8, device=out.device) if need else None
    _lib.check(lib.hct_gemm(C.byref(g), _lib.ptr(w), w.numel() if w is not None else 0, _st()), "hct_gemm")
    return out


class _HeadFunction(torch.autograd.Function):
    """DINOHead.forward / backward over hct_gemm (bias + GELU epilogues, dGELU dgrad, wgrad) and the L2-norm / weight-norm kernels."""

    @staticmethod
    def forward(ctx, anchor, head, x):
        lib = _lib.load()
        dev = x.device
        cd = torch.bfloat16 if head.compute_dtype == "bf16" else torch.float32
        M, D = x.shape
        H, Bn, K = head.hidden_dim, head.bottleneck_dim, head.out_dim
        x = x.to(cd).contiguous()
        W = head._working_weights(cd)
        mk = lambda n, dt=cd: torch.empty(M, n, dtype=dt, device=dev)
        l0, l1, l2 = head._lin
        bn_saved = []
        if not head.use_bn:
            u1, h1, u2, h2 = mk(H), mk(H), mk(H), mk(H)
            _gemm(lib, x, W[f"mlp.{l0}.weight"], 0, 1, M, H, D, h1, bias=head.mlp[l0].bias, act=1, aux=u1)      # Linear + GELU (exact erf)
            _gemm(lib, h1, W[f"mlp.{l1}.weight"], 0, 1, M, H, H, h2, bias=head.mlp[l1].bias, act=1, aux=u2)
        else:
            # Linear -> BatchNorm1d -> GELU (dino_head.py:15-21): the Linear's output stays fp32, the statistics are the batch's in
            # training (shared over the ranks like the reference's SyncBatchNorm, main_pretrain_dino.py:183-185) and the running
            # ones in eval; xhat and gelu'(y) are kept for the backward
            u1 = u2 = None
            h1, h2 = mk(H), mk(H)
            inp, width = x, D
            for li, bi, hout in ((l0, head._bn[0], h1), (l1, head._bn[1], h2)):
                u = mk(H, torch.float32)
                _gemm(lib, inp, W[f"mlp.{li}.weight"], 0, 1, M, H, width, u, bias=head.mlp[li].bias)
                bn = head.mlp[bi]
                mean, var, count = head._bn_statistics(lib, bn, u, M, H)
                xhat, dact = mk(H, torch.float32), mk(H, torch.float32)
                _lib.check(lib.hct_bn_gelu_fwd(u.data_ptr(), mean.data_ptr(), var.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), M, H,
                                               hout.data_ptr(), _code(hout), xhat.data_ptr(), dact.data_ptr(), _st()), "hct_bn_gelu_fwd")
                bn_saved.append((xhat, dact, var, count))
                inp, width = hout, H
        z = mk(Bn, torch.float32)
        _gemm(lib, h2, W[f"mlp.{l2}.weight"], 0, 1, M, Bn, H, z, bias=head.mlp[l2].bias)
        zn, inv_z = mk(Bn), torch.empty(M, dtype=torch.float32, device=dev)
        _lib.check(lib.hct_l2norm_rows_fwd(z.data_ptr(), M, Bn, zn.data_ptr(), _code(zn), inv_z.data_ptr(), _st()), "hct_l2norm_rows_fwd")
        wn = torch.empty(K, Bn, dtype=cd, device=dev)
        inv_v = torch.empty(K, dtype=torch.float32, device=dev)
        ll = head.last_layer
        _lib.check(lib.hct_weight_norm_fwd(ll.weight_v.data_ptr(), ll.weight_g.data_ptr(), K, Bn, wn.data_ptr(), _code(wn), inv_v.data_ptr(), _st()),
                   "hct_weight_norm_fwd")
        logits = torch.empty(M, K, dtype=cd, device=dev)
        _gemm(lib, zn, wn, 0, 1, M, K, Bn, logits)
        # the dgrad of the prototype layer runs as a split-K product over the prototypes when it can (bf16, 16-aligned); only the
        # NT fallback needs W_n^T
        wn_t = None
        if not (cd == torch.bfloat16 and M % 16 == 0 and Bn % 16 == 0):
            wn_t = torch.empty(Bn, K, dtype=cd, device=dev)
            _lib.check(lib.hct_transpose_cast(wn.data_ptr(), _code(wn), wn_t.data_ptr(), _code(wn_t), K, Bn, _st()), "hct_transpose_cast")
        ctx.head, ctx.saved, ctx.wn, ctx.bn_saved = head, (x, u1, h1, u2, h2, zn, inv_z, wn_t, inv_v, W), wn, bn_saved
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        head = ctx.head
        x, u1, h1, u2, h2, zn, inv_z, wn_t, inv_v, W = ctx.saved
        dev, cd = x.device, x.dtype
        M, D = x.shape
        H, Bn, K = head.hidden_dim, head.bottleneck_dim, head.out_dim
        dl = dlogits.to(cd).contiguous()
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        ll = head.last_layer
        g = head._flat_grad
        gv = lambda name: g[head._off[name]:head._off[name] + head._named_cache[name].numel()].view(head._named_cache[name].shape)
        # prototype layer: dzn = dlogits . Wn ;  dWn = dlogits^T . zn  -> weight-norm backward into weight_v (weight_g frozen or not)
        # (as a split-K "TN" product over the 65 536 prototypes: dl^T [K, M] and Wn [K, Bn] -- the NT form is one 80 x 256 output
        #  tile with a 65 536-deep reduction on a single CU, 1.2 ms)
        dzn = f32(M, Bn)
        if wn_t is None:
            dl_t = torch.empty(K, M, dtype=cd, device=dev)
            _lib.check(lib.hct_transpose_cast(dl.data_ptr(), _code(dl), dl_t.data_ptr(), _code(dl_t), M, K, _st()), "hct_transpose_cast")
            _gemm(lib, dl_t, ctx.wn, 1, 0, M, Bn, K, dzn)
        else:
            _gemm(lib, dl, wn_t, 0, 1, M, Bn, K, dzn)
        dwn = f32(K, Bn)
        _gemm(lib, dl, zn, 1, 0, K, Bn, M, dwn)
        dg = gv("last_layer.weight_g") if ll.weight_g.requires_grad else None
        _lib.check(lib.hct_weight_norm_bwd(dwn.data_ptr(), ll.weight_v.data_ptr(), ll.weight_g.data_ptr(), inv_v.data_ptr(), K, Bn,
                                           gv("last_layer.weight_v").data_ptr(), _lib.ptr(dg), _st()), "hct_weight_norm_bwd")
        dz = f32(M, Bn)
        _lib.check(lib.hct_l2norm_rows_bwd(dzn.data_ptr(), zn.data_ptr(), _code(zn), inv_z.data_ptr(), M, Bn, dz.data_ptr(), _st()), "hct_l2norm_rows_bwd")
        dzc = dz.to(cd)
        ws = torch.empty(max(16, lib.hct_colsum_workspace_bytes(M, max(H, Bn))), dtype=torch.uint8, device=dev)
        colsum = lambda t, n, name: _lib.check(lib.hct_colsum(t.data_ptr(), _code(t), M, n, n, gv(name).data_ptr(), ws.data_ptr(), ws.numel(), _st()), "hct_colsum")
        l0, l1, l2 = head._lin
        # last Linear of the MLP
        _gemm(lib, dzc, h2, 1, 0, Bn, H, M, gv(f"mlp.{l2}.weight"))
        colsum(dzc, Bn, f"mlp.{l2}.bias")
        du2 = torch.empty(M, H, dtype=cd, device=dev)
        du1 = torch.empty(M, H, dtype=cd, device=dev)
        if not head.use_bn:
            _gemm(lib, dzc, W[f"mlp.{l2}.weight_t"], 0, 1, M, H, Bn, du2, act=2, aux=u2)        # (dz . W) * gelu'(u2)
        else:
            dh2 = torch.empty(M, H, dtype=cd, device=dev)
            _gemm(lib, dzc, W[f"mlp.{l2}.weight_t"], 0, 1, M, H, Bn, dh2)
            head._bn_backward(lib, head.mlp[head._bn[1]], dh2, ctx.bn_saved[1], M, H, du2, gv)
        # second Linear
        _gemm(lib, du2, h1, 1, 0, H, H, M, gv(f"mlp.{l1}.weight"))
        colsum(du2, H, f"mlp.{l1}.bias")
        if not head.use_bn:
            _gemm(lib, du2, W[f"mlp.{l1}.weight_t"], 0, 1, M, H, H, du1, act=2, aux=u1)
        else:
            dh1 = torch.empty(M, H, dtype=cd, device=dev)
            _gemm(lib, du2, W[f"mlp.{l1}.weight_t"], 0, 1, M, H, H, dh1)
            head._bn_backward(lib, head.mlp[head._bn[0]], dh1, ctx.bn_saved[0], M, H, du1, gv)
        # first Linear
        _gemm(lib, du1, x, 1, 0, H, D, M, gv(f"mlp.{l0}.weight"))
        colsum(du1, H, f"mlp.{l0}.bias")
        dx = torch.empty(M, D, dtype=torch.float32, device=dev)
        _gemm(lib, du1, W[f"mlp.{l0}.weight_t"], 0, 1, M, D, H, dx)
        head._attach_grads()
        return None, None, dx


class _BatchNorm(_Holder):
    """Parameters and buffers of nn.BatchNorm1d(n) / SyncBatchNorm under the reference's names (weight, bias, running_mean, running_var,
    num_batches_tracked): a holder, the arithmetic runs in the head's kernels."""

    def __init__(self, n: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class DINOHead(nn.Module, _FlatParams):
    """Reference DINOHead (src/models/dino_head.py) with nlayers=3; use_bn=False (the reference yaml) or True (the default of
    config.py:86: Linear -> BatchNorm1d -> GELU, batch statistics shared over the ranks like the SyncBatchNorm the reference converts
    to, main_pretrain_dino.py:183-185)."""

    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048, bottleneck_dim=256,
                 compute_dtype: str = "bf16"):
        super().__init__()
        if nlayers != 3:
            raise NotImplementedError("HIP DINOHead: nlayers=3 (the reference's head)")
        if in_dim % 4 or hidden_dim % 4 or bottleneck_dim % 4 or out_dim % 4:
            raise HctError("HIP DINOHead: dimensions must be multiples of 4")
        self.in_dim, self.out_dim, self.hidden_dim, self.bottleneck_dim = in_dim, out_dim, hidden_dim, bottleneck_dim
        self.compute_dtype = compute_dtype
        self.use_bn = bool(use_bn)
        lin = lambda o, i: _Affine(o, i, bias_shape=(o,))
        if self.use_bn:  # Sequential indices as in the reference: Linear 0, BatchNorm1d 1, GELU 2, Linear 3, BatchNorm1d 4, GELU 5, Linear 6
            mlp = [lin(hidden_dim, in_dim), _BatchNorm(hidden_dim), _Holder(), lin(hidden_dim, hidden_dim), _BatchNorm(hidden_dim), _Holder(),
                   lin(bottleneck_dim, hidden_dim)]
            self._lin, self._bn = (0, 3, 6), (1, 4)
        else:
            mlp = [lin(hidden_dim, in_dim), _Holder(), lin(hidden_dim, hidden_dim), _Holder(), lin(bottleneck_dim, hidden_dim)]
            self._lin, self._bn = (0, 2, 4), ()
        self.mlp = nn.Sequential(*mlp)  # the Linear / BatchNorm indices carry the parameters, as in the reference's Sequential
        self.last_layer = _Holder()
        self.last_layer.weight_g = nn.Parameter(torch.ones(out_dim, 1), requires_grad=not norm_last_layer)
        self.last_layer.weight_v = nn.Parameter(torch.empty(out_dim, bottleneck_dim))
        with torch.no_grad():
            for i in self._lin:
                nn.init.trunc_normal_(self.mlp[i].weight, std=.02)
                nn.init.constant_(self.mlp[i].bias, 0)
            nn.init.kaiming_uniform_(self.last_layer.weight_v, a=math.sqrt(5))  # nn.Linear default, then weight_norm splits g / v
        self._managed_updates = False
        self._grad_prescale = 1.0
        self._wver, self._wcache = -1, {}
        self._build_flat(torch.device("cpu"))
        self._off = {n: o for n, o, *_ in self._layout}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._build_flat(next(self.parameters()).device)
        self._off = {n: o for n, o, *_ in self._layout}
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._weights_version += 1
        return out

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=set_to_none)

    def _working_weights(self, cd) -> Dict[str, torch.Tensor]:
        """Linear weights W [out, in] and their transposes W^T [in, out] in the compute dtype: forward products are NT GEMMs with
        W, dgrad products NT GEMMs with W^T (the MFMA kernels take both operands K-contiguous).  Copies are refreshed when the
        masters changed; without a HipAdamW reporting updates they are rebuilt every forward."""
        if self._managed_updates and self._wver == self._weights_version and self._wcache.get("dtype") == cd:
            return self._wcache
        lib = _lib.load()
        code = HCT_BF16 if cd == torch.bfloat16 else HCT_F32
        out = {"dtype": cd}
        for i in self._lin:
            w = self.mlp[i].weight.detach()
            if cd == torch.float32:
                out[f"mlp.{i}.weight"] = w
            else:
                d = torch.empty(w.shape, dtype=cd, device=w.device)
                _lib.check(lib.hct_cast(w.data_ptr(), HCT_F32, d.data_ptr(), code, w.numel(), _st()), "hct_cast")
                out[f"mlp.{i}.weight"] = d
            t = torch.empty(w.shape[1], w.shape[0], dtype=cd, device=w.device)
            _lib.check(lib.hct_transpose_cast(w.data_ptr(), HCT_F32, t.data_ptr(), code, w.shape[0], w.shape[1], _st()), "hct_transpose_cast")
            out[f"mlp.{i}.weight_t"] = t
        self._wcache, self._wver = out, self._weights_version
        return out

    def _bn_statistics(self, lib, bn, u: torch.Tensor, M: int, H: int):
        """(mean, var, count) the BatchNorm normalises with.  Training: the batch's (biased variance), over the rows of ALL ranks when a
        process group is up (SyncBatchNorm), and the running statistics move (momentum 0.1, unbiased variance).  Eval: the running ones."""
        import torch.distributed as dist
        if not self.training:
            return bn.running_mean, bn.running_var, float(M)
        if M < 2:
            raise HctError("BatchNorm1d in training mode needs more than one row")
        mean, var = torch.empty(H, device=u.device), torch.empty(H, device=u.device)
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            _lib.check(lib.hct_batchnorm_stats(u.data_ptr(), M, H, float(bn.momentum), mean.data_ptr(), var.data_ptr(), bn.running_mean.data_ptr(),
                                               bn.running_var.data_ptr(), _st()), "hct_batchnorm_stats")
            bn.num_batches_tracked += 1
            return mean, var, float(M)
        # equal row counts on every rank (same batch size): global mean = mean of means, E[u^2] likewise (2 x H floats of glue)
        _lib.check(lib.hct_batchnorm_stats(u.data_ptr(), M, H, 0.0, mean.data_ptr(), var.data_ptr(), None, None, _st()), "hct_batchnorm_stats")
        both = torch.stack([mean, var + mean * mean])
        dist.all_reduce(both)
        both /= world
        gmean, gvar = both[0], (both[1] - both[0] * both[0]).clamp_min_(0.0)
        n = float(M * world)
        with torch.no_grad():
            bn.running_mean.mul_(1 - bn.momentum).add_(gmean, alpha=bn.momentum)
            bn.running_var.mul_(1 - bn.momentum).add_(gvar * (n / (n - 1)), alpha=bn.momentum)
            bn.num_batches_tracked += 1
        return gmean.contiguous(), gvar.contiguous(), n

    def _bn_backward(self, lib, bn, dh: torch.Tensor, saved, M: int, H: int, du: torch.Tensor, gv) -> None:
        """dh = gradient wrt the GELU's output -> du = gradient wrt the Linear's output; the BatchNorm's weight / bias gradients."""
        import torch.distributed as dist
        xhat, dact, var, count = saved
        name = next(f"mlp.{i}" for i in self._bn if self.mlp[i] is bn)
        sums = torch.empty(2, H, device=dh.device)
        _lib.check(lib.hct_bn_gelu_bwd_sums(dh.data_ptr(), _code(dh), dact.data_ptr(), xhat.data_ptr(), M, H, sums.data_ptr(), _st()), "hct_bn_gelu_bwd_sums")
        gv(name + ".bias").copy_(sums[0])    # this rank's rows: the data-parallel gradient mean adds the other ranks'
        gv(name + ".weight").copy_(sums[1])
        if count > M:  # statistics shared over the ranks: so are the two sums that enter du
            dist.all_reduce(sums)
        _lib.check(lib.hct_bn_gelu_bwd_apply(dh.data_ptr(), _code(dh), dact.data_ptr(), xhat.data_ptr(), bn.weight.data_ptr(), var.data_ptr(), float(bn.eps),
                                             sums.data_ptr(), float(count), M, H, du.data_ptr(), _code(du), _st()), "hct_bn_gelu_bwd_apply")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise HctError("DINOHead (HIP) got a CPU tensor: this path has no CPU fallback")
        return _HeadFunction.apply(self.mlp[0].weight, self, x)


# ================================================================================================
# multi-crop wrapper
# ================================================================================================
class MultiCropWrapper(nn.Module):
    """One backbone pass per crop resolution over the concatenated crops, head on the class-token features (misc.py:447-484)."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone, self.head = backbone, head

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        for m in (self.backbone, self.head):  # the children's fp32 masters changed underneath their bf16 working copies
            if hasattr(m, "mark_weights_updated"):
                m.mark_weights_updated()
        return out

    def forward(self, x):
        if not isinstance(x, list):
            x = [x]
        sizes = [int(t.shape[-1]) for t in x]
        feats, start = [], 0
        while start < len(x):  # runs of consecutive crops with the same last dimension (torch.unique_consecutive in the reference)
            end = start
            while end < len(x) and sizes[end] == sizes[start]:
                end += 1
            out = self.backbone(x[start:end] if hasattr(self.backbone, "_plan_for") else torch.cat(x[start:end]))  # the native backbone reads the crops in place
            feats.append(out[0] if isinstance(out, tuple) else out)
            start = end
        tokens = torch.cat(feats)
        return {'dino_output': self.head(tokens[:, 0, :])}
