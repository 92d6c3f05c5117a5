"""MaskedAutoencoderViT: drop-in host module over the MI355X HIP hot path.

Mirrors the reference's nn.Module contract (src/models/mae.py:20-317): identical constructor kwargs,
`forward(x) -> (loss, None, None)`, parameter names/shapes/registration order (so `state_dict()`,
`load_state_dict()`, `.parameters()` and checkpoints are interchangeable), the reference initialisation
(mae.py:125-148, patch_embedding.py:107-124), `patchify` / `unpatchify`.

All arithmetic runs in libheadct_hip.so.  PyTorch only provides device memory, the RNG draw of the
masking noise (mae.py:206) and the autograd hand-off.  There is no CPU fallback: calling forward on a
CPU tensor, or without the built library, raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HCT_BF16, HCT_F32, HctError

_POS = {"none": 0, "learnable": 1, "sincos": 2}


def _to_3tuple(x):
    return tuple(x) if isinstance(x, (list, tuple)) else (x, x, x)


def build_sincos_position_embedding(grid_size, embed_dim: int, spatial_dims: int = 3, temperature: float = 10000.0):
    """Fixed 3-D sine/cosine position table [1, L, D] (contract: src/utils/pos_embed.py:51-78).

    D/6 frequencies 1 / T^(j / (D/6)); token (a, b, c) of the row-major grid gets, in this order, sin and cos of its b, a
    and c coordinate times the frequencies.  (The reference names the axes so that the second grid axis comes first; for
    the cubic grids of this path only that order matters.)  fp32 throughout, one multiply per entry, so the table is
    bit-identical to the reference's (tests/golden/sincos.json)."""
    if spatial_dims != 3:
        raise NotImplementedError(f"Spatial Dimension Size {spatial_dims} Not Implemented!")
    if embed_dim % 6:
        raise AssertionError("Embed dimension must be divisible by 6 for 3D sin-cos position embedding")
    n0, n1, n2 = _to_3tuple(grid_size)
    nfreq = embed_dim // 6
    freq = 1.0 / (temperature ** (torch.arange(nfreq, dtype=torch.float32) / nfreq))
    # coordinates of every token along the three axes of the (n1, n0, n2) meshgrid the reference builds
    axes = [torch.arange(n, dtype=torch.float32) for n in (n1, n0, n2)]
    shape = (n1, n0, n2)
    coords = [ax.reshape([-1 if k == i else 1 for k in range(3)]).expand(shape).reshape(-1) for i, ax in enumerate(axes)]
    parts = []
    for i in (1, 0, 2):
        angle = coords[i][:, None] * freq[None, :]
        parts += [torch.sin(angle), torch.cos(angle)]
    return torch.cat(parts, dim=1).unsqueeze(0)


class _Holder(nn.Module):
    """Parameter container: gives parameters their reference names; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise HctError("sub-modules of the HIP MaskedAutoencoderViT are parameter holders; call the model itself")


class _Affine(_Holder):
    def __init__(self, *wshape, bias_shape=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        if bias_shape is not None:
            self.bias = nn.Parameter(torch.empty(*bias_shape))
        else:
            self.register_parameter("bias", None)


def _block(d: int, m: int, qkv_bias: bool) -> nn.Module:
    """Names of AttentionBlock (attentionblock.py:91-94) + MONAI MLPBlock (linear1/linear2)."""
    blk = _Holder()
    blk.mlp = _Holder()
    blk.mlp.linear1 = _Affine(m, d, bias_shape=(m,))
    blk.mlp.linear2 = _Affine(d, m, bias_shape=(d,))
    blk.att_norm = _Affine(d, bias_shape=(d,))
    blk.ffn_norm = _Affine(d, bias_shape=(d,))
    blk.attn = _Holder()
    blk.attn.qkv = _Affine(3 * d, d, bias_shape=(3 * d,) if qkv_bias else None)
    blk.attn.proj = _Affine(d, d, bias_shape=(d,))
    return blk


class _Plan:
    """One bound native plan (per batch size)."""

    def __init__(self, model: "MaskedAutoencoderViT", batch: int):
        lib = _lib.load()
        self.lib = lib
        self.batch = batch
        self.serial = 0
        self.handle = lib.hct_mae_plan_create(C.byref(model._ccfg), batch, model._dt)
        if not self.handle:
            raise HctError("hct_mae_plan_create: " + lib.hct_last_error_string().decode())
        if lib.hct_mae_plan_len_keep(self.handle) != model.len_keep:
            raise HctError(f"native plan keeps {lib.hct_mae_plan_len_keep(self.handle)} patches, the module {model.len_keep}")
        nbytes = lib.hct_mae_plan_workspace_bytes(self.handle)
        dev = model._flat.device
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.nstages = lib.hct_mae_num_backward_stages(self.handle)
        self.stage_ranges = []
        for s in range(self.nstages):
            b, e = C.c_int64(), C.c_int64()
            _lib.check(lib.hct_mae_backward_stage_range(self.handle, s, C.byref(b), C.byref(e)), "stage_range")
            self.stage_ranges.append((b.value, e.value))
        self.rebind(model)

    def rebind(self, model):
        _lib.check(self.lib.hct_mae_plan_bind(
            self.handle, model._flat.data_ptr(), model._flat_grad.data_ptr(),
            _lib.ptr(model._flat_bf16), _lib.ptr(model._flat_bf16_t), self.workspace.data_ptr(), self.workspace.numel()),
            "hct_mae_plan_bind")

    def activation(self, name: str) -> torch.Tensor:
        rows, cols, dt = C.c_int64(), C.c_int64(), C.c_int()
        p = self.lib.hct_mae_plan_activation(self.handle, name.encode(), C.byref(rows), C.byref(cols), C.byref(dt))
        if not p:
            raise KeyError(name)
        tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.int32}[dt.value]
        off = p - self.workspace.data_ptr()
        n = rows.value * cols.value
        esz = torch.empty(0, dtype=tdt).element_size()
        return self.workspace[off:off + n * esz].view(tdt).view(rows.value, cols.value)

    def __del__(self):
        try:
            if self.handle:
                self.lib.hct_mae_plan_destroy(self.handle)
        except Exception:
            pass


class _MAEFunction(torch.autograd.Function):
    """Autograd hand-off: forward enqueues the native forward, backward the staged native backward, which
    writes straight into the model's flat gradient buffer (`p.grad` are views of it)."""

    @staticmethod
    def forward(ctx, anchor, model, x, noise, train):
        plan = model._plan_for(x.shape[0])
        st = _lib.stream_ptr()
        model._ensure_weights_fresh(plan, st)
        # a training forward needs no prediction for the kept patches (the loss drops them, mae.py:298-299): the decoder's tail
        # then runs on the masked patches' rows only, unless the caller asked for the full prediction (`full_pred`)
        plan.tail = bool(plan.lib.hct_mae_plan_set_tail(plan.handle, int(train and not getattr(model, "full_pred", False))) == 1)
        if not getattr(model, "dec0_table", True):  # (testing: the first decoder block on every row instead of kept rows + one row per position)
            plan.lib.hct_mae_plan_set_dec0(plan.handle, 0)
        # training forward: the loss pass also leaves d(loss)/d(pred) (scaled by 1/world under data parallelism) for the backward
        xdt = _lib.HCT_F16 if x.dtype == torch.float16 else HCT_F32
        _lib.check(plan.lib.hct_mae_forward(plan.handle, x.data_ptr(), xdt, noise.data_ptr(), plan.loss.data_ptr(),
                                            float(model._grad_prescale) if train else 0.0, st), "hct_mae_forward")
        plan.serial += 1  # the plan's one activation workspace now belongs to this forward
        ctx.model, ctx.plan, ctx.x, ctx.serial = model, plan, x, plan.serial
        return plan.loss[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        model, plan, x = ctx.model, ctx.plan, ctx.x
        if plan.serial != ctx.serial:
            raise HctError("backward of a stale forward: another forward at the same batch size has overwritten the activation "
                           "workspace (run loss.backward() before the next model(...) call, e.g. before an eval pass)")
        model._run_backward(plan, x, grad_out)
        return None, None, None, None, None


class FlatPlanModule(nn.Module):
    """Host side shared by the plan-driven models (MAE, ViT backbone): every Parameter is a view into ONE flat fp32 buffer laid
    out by the native plan, gradients live in a second flat buffer that the staged native backward fills from its end to its
    start (= gradient-bucket order for the data-parallel all-reduce), bf16 working copies are refreshed when the masters change.
    Subclasses register their parameters under the reference's names, fill `self._ccfg` / `self._dt`, then call
    `_init_flat_state()` and `_build_flat(cpu)`."""

    def _init_flat_state(self) -> None:
        self._plans: Dict[int, _Plan] = {}
        self._weights_version = 0      # bumped whenever fp32 master weights may have changed
        self._shadow_version = -1      # version the bf16 working copies correspond to
        self._bucket_hook: Optional[Callable[[int, int, int], None]] = None  # (stage, begin, end): gradient range that became final
        self.wgrad_group_blocks: Optional[int] = None  # data parallelism: flush the queued weight gradients every n block stages
        self._post_backward_hook: Optional[Callable[[], None]] = None
        self._grad_overwrite = True    # next backward overwrites the flat gradient (set by zero_grad paths)
        self._grad_prescale = 1.0
        self._managed_updates = False  # True once a HipAdamW owns the weight updates
        self._plain_fresh = False
        self._layout: List[Tuple[str, int, int, Tuple[int, ...], bool, int]] = []

    # ------------------------------------------------------------------------------------------
    # flat storage: every Parameter is a view into one fp32 buffer laid out by the native plan
    # ------------------------------------------------------------------------------------------
    def _query_layout(self):
        lib = _lib.load()
        h = lib.hct_mae_plan_create(C.byref(self._ccfg), 1, self._dt)
        if not h:
            raise HctError("hct_mae_plan_create: " + lib.hct_last_error_string().decode())
        try:
            layout = []
            info = _lib.ParamInfo()
            for i in range(lib.hct_mae_plan_num_params(h)):
                _lib.check(lib.hct_mae_plan_param_info(h, i, C.byref(info)), "param_info")
                shape = tuple(int(info.shape[k]) for k in range(info.ndim))
                layout.append((info.name.decode(), int(info.offset), int(info.numel), shape, bool(info.requires_grad), int(info.bf16_t_offset)))
            total = int(lib.hct_mae_plan_param_elems(h))
            total_t = int(lib.hct_mae_plan_bf16_t_elems(h))
        finally:
            lib.hct_mae_plan_destroy(h)
        return layout, total, total_t

    def _build_flat(self, device: torch.device) -> None:
        layout, total, total_t = self._query_layout()
        named = dict(self.named_parameters())
        if set(named) != {n for n, *_ in layout}:
            raise HctError(f"parameter name mismatch between host module and native plan: {set(named) ^ {n for n, *_ in layout}}")
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        old_grads = {n: p.grad for n, p in named.items()}
        for name, off, numel, shape, rg, _ in layout:
            p = named[name]
            if tuple(p.shape) != shape:
                raise HctError(f"shape mismatch for {name}: {tuple(p.shape)} vs {shape}")
            flat[off:off + numel].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[off:off + numel].view(shape)
            p.requires_grad_(rg and p.requires_grad)
        self._layout = layout
        self._flat = flat
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        for name, off, numel, shape, rg, _ in layout:
            g = old_grads[name]
            if g is not None:
                self._flat_grad[off:off + numel].copy_(g.reshape(-1))
                named[name].grad = self._flat_grad[off:off + numel].view(shape)
        if self._dt == HCT_BF16:
            self._flat_bf16 = torch.zeros(total, dtype=torch.bfloat16, device=device)
            self._flat_bf16_t = torch.zeros(max(total_t, 1), dtype=torch.bfloat16, device=device)
        else:
            self._flat_bf16 = self._flat_bf16_t = None
        seg = sorted((off for _, off, *_ in layout)) + [total]
        self._seg_off_host = seg
        self._seg_names = [n for n, *_ in sorted(layout, key=lambda t: t[1])]
        self._plans = {}
        self._weights_version += 1
        self._plain_fresh = False
        self._named_cache = named

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        # parameters were moved/cast one by one; rebuild the flat buffer on their new device
        dev = next(self.parameters()).device
        self._build_flat(dev)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._weights_version += 1
        self._plain_fresh = False  # the bf16 copies written by the last optimizer step no longer match the masters
        return out

    def mark_weights_updated(self, plain_bf16_fresh: bool = False) -> None:
        """Tell the model the fp32 master weights changed (optimizer step / manual edit)."""
        self._weights_version += 1
        self._plain_fresh = plain_bf16_fresh

    def flat_segments(self):
        """(names, element offsets[nseg+1]) of the flat parameter/gradient buffers."""
        return self._seg_names, self._seg_off_host

    # ------------------------------------------------------------------------------------------
    def _plan_for(self, batch: int) -> _Plan:
        if not self._flat.is_cuda:
            raise HctError(f"{type(self).__name__} (HIP) needs its parameters on a GPU: call .to('cuda') first; "
                           "there is no CPU fallback for this path")
        plan = self._plans.get(batch)
        if plan is None:
            plan = _Plan(self, batch)
            self._plans[batch] = plan
        return plan

    def _ensure_weights_fresh(self, plan: _Plan, st: int) -> None:
        if self._dt != HCT_BF16:
            return
        # unless a fused optimizer reports every update (mark_weights_updated), assume the fp32 masters may
        # have been modified behind our back (e.g. torch.optim.AdamW) and refresh on every forward.
        if self._managed_updates and self._shadow_version == self._weights_version:
            return
        with_plain = 0 if getattr(self, "_plain_fresh", False) else 1
        _lib.check(plan.lib.hct_mae_refresh_weights(plan.handle, with_plain, st), "hct_mae_refresh_weights")
        self._plain_fresh = False
        self._shadow_version = self._weights_version

    def _attach_grads(self) -> bool:
        """Point every trainable parameter's .grad at its slice of the flat gradient buffer.
        Returns True when some parameter already held a gradient (accumulation requested)."""
        accumulate = False
        named = self._named_cache
        for name, off, numel, shape, rg, _ in self._layout:
            p = named[name]
            if not p.requires_grad:
                continue
            view = self._flat_grad[off:off + numel].view(shape)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() == view.data_ptr():
                accumulate = accumulate or not self._grad_overwrite
            else:  # foreign gradient tensor: fold it in
                view.copy_(p.grad)
                p.grad = view
                accumulate = True
        return accumulate

    def _run_backward(self, plan: _Plan, x: torch.Tensor, grad_out: torch.Tensor) -> None:
        """MAE: the staged native backward seeded by the (device) scalar dLoss."""
        lib = plan.lib
        g = grad_out.detach().to(dtype=torch.float32).reshape(1).contiguous()
        _lib.check(lib.hct_mae_set_loss_grad(plan.handle, g.data_ptr()), "hct_mae_set_loss_grad")
        self._keep_alive = g
        st = _lib.stream_ptr()
        self._run_staged_backward(plan, lambda s: lib.hct_mae_backward_stage(plan.handle, s, st), "hct_mae_backward_stage")

    def _run_staged_backward(self, plan: _Plan, stage_call, what: str) -> None:
        lib = plan.lib
        # a second backward without zero_grad() adds to what is there (torch semantics).  The native stages overwrite the
        # flat buffer, so the earlier gradient is parked and added back at the end -- AFTER the data-parallel reduction of
        # the fresh gradient (every backward is reduced, as torch's DDP does; the parked part is already the mean).
        parked = None
        if not self._grad_overwrite and any(p.grad is not None for p in self.parameters()):
            self._attach_grads()  # a foreign / preset .grad tensor is folded into the flat buffer first
            parked = self._flat_grad.clone()
        # weight gradients are queued across stages and run in grouped launches (csrc/mae_plan.hip: flush_wgrads), so a stage's
        # range is final only when the plan's watermark has passed it: the bucket hook gets [watermark, previous watermark)
        if self._bucket_hook is not None and getattr(self, "wgrad_group_blocks", None) is not None and getattr(plan, "_wg_blocks", None) != self.wgrad_group_blocks:
            lib.hct_mae_plan_set_wgrad_defer(plan.handle, 1, int(self.wgrad_group_blocks))
            plan._wg_blocks = self.wgrad_group_blocks
        final = self._flat_grad.numel()
        for s in range(plan.nstages):
            _lib.check(stage_call(s), f"{what}({s})")
            if self._bucket_hook is not None:
                now = int(lib.hct_mae_backward_final_offset(plan.handle))
                if now < final:
                    self._bucket_hook(s, now, final)
                    final = now
        if self._post_backward_hook is not None:
            self._post_backward_hook()  # data parallel: the compute stream now waits for the collectives
        if parked is not None:
            self._flat_grad.add_(parked)
        self._attach_grads()
        self._grad_overwrite = False

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=set_to_none)
        self._grad_overwrite = True


class MaskedAutoencoderViT(FlatPlanModule):
    """Masked Autoencoder with VisionTransformer backbone (HIP / gfx950 implementation)."""

    def __init__(self, input_size: int, patch_size: int, mask_ratio: float, in_chans: int = 1, dropout_rate: float = 0.,
                 spatial_dims: int = 3, patch_embed: str = 'conv', pos_embed: str = 'learnable', encoder_depth: int = 12,
                 encoder_embed_dim: int = 768, encoder_mlp_dim: int = 3072, encoder_num_heads: int = 12,
                 decoder_depth: int = 8, decoder_embed_dim: int = 768, decoder_mlp_dim: int = 3072,
                 decoder_num_heads: int = 16, norm_pix_loss: bool = False, use_bias: bool = False,
                 norm_layer=nn.LayerNorm, compute_dtype: str = "bf16"):
        super().__init__()
        input_size, patch_size = _to_3tuple(input_size), _to_3tuple(patch_size)
        if spatial_dims != 3 or len(set(input_size)) != 1 or len(set(patch_size)) != 1:
            raise HctError("the HIP MAE path supports cubic 3-D volumes and patches")
        if patch_embed != "conv":
            raise ValueError(f"patch_embed type {patch_embed} not supported.")
        if pos_embed not in _POS:
            raise ValueError(f"pos_embed type {pos_embed} not supported.")
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if dropout_rate != 0.0:
            raise HctError("dropout_rate != 0 is outside the HIP hot path (the reference MAE yaml uses 0.)")
        if norm_layer is not nn.LayerNorm:
            raise HctError("only nn.LayerNorm is supported on the HIP hot path (MAE.NORM_LAYER: layernorm)")
        if encoder_embed_dim % encoder_num_heads or decoder_embed_dim % decoder_num_heads:
            raise ValueError("hidden_size should be divisible by num_heads.")
        for m, p in zip(input_size, patch_size):
            if m < p:
                raise ValueError("patch_size should be smaller than img_size.")
            assert m % p == 0, "input size and patch size are not proper"
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")

        self.input_size, self.patch_size = input_size, patch_size
        self.mask_ratio, self.spatial_dims, self.pos_embed, self.norm_pix_loss = mask_ratio, spatial_dims, pos_embed, norm_pix_loss
        self.encoder_embed_dim, self.decoder_embed_dim = encoder_embed_dim, decoder_embed_dim
        self.in_chans = in_chans
        self.out_chans = in_chans * int(np.prod(patch_size))
        self.grid_size = [i // p for i, p in zip(input_size, patch_size)]
        self.compute_dtype = compute_dtype
        num_patches = int(np.prod(self.grid_size))
        self.num_patches = num_patches
        D, Dd, P = encoder_embed_dim, decoder_embed_dim, patch_size[0]

        # ---- parameters: the reference's names and registration order (mae.py:90-121) ----
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.decoder_cls_token = nn.Parameter(torch.zeros(1, 1, Dd))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches, Dd), requires_grad=False)
        self.patch_embedding = _Holder()
        self.patch_embedding.n_patches = num_patches  # attribute interpolate_pos_embed reads (patch_embedding.py:96)
        if pos_embed != "none":
            self.patch_embedding.position_embeddings = nn.Parameter(torch.zeros(1, num_patches, D))
        else:
            self.patch_embedding.position_embeddings = None
        self.patch_embedding.patch_embeddings = _Affine(D, in_chans, P, P, P, bias_shape=(D,))
        self.blocks = nn.ModuleList([_block(D, encoder_mlp_dim, use_bias) for _ in range(encoder_depth)])
        self.decoder_blocks = nn.ModuleList([_block(Dd, decoder_mlp_dim, use_bias) for _ in range(decoder_depth)])
        self.norm = _Affine(D, bias_shape=(D,))
        self.decoder_norm = _Affine(Dd, bias_shape=(Dd,))
        self.decoder_embed = _Affine(Dd, D, bias_shape=(Dd,) if use_bias else None)
        self.decoder_pred = _Affine(P ** 3 * in_chans, Dd, bias_shape=(P ** 3 * in_chans,) if use_bias else None)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, Dd))

        self._ccfg = _lib.MaeConfig(
            input_size=input_size[0], patch_size=P, in_chans=in_chans, mask_ratio=float(mask_ratio), pos_embed=_POS[pos_embed],
            encoder_depth=encoder_depth, encoder_embed_dim=D, encoder_mlp_dim=encoder_mlp_dim, encoder_num_heads=encoder_num_heads,
            decoder_depth=decoder_depth, decoder_embed_dim=Dd, decoder_mlp_dim=decoder_mlp_dim, decoder_num_heads=decoder_num_heads,
            norm_pix_loss=int(bool(norm_pix_loss)), use_bias=int(bool(use_bias)))
        self._dt = HCT_BF16 if compute_dtype == "bf16" else HCT_F32
        self.len_keep = int(num_patches * (1 - mask_ratio))  # mae.py:205
        self.full_pred = False  # True: training forwards also predict the kept patches (parity tests, reconstructions)

        self._init_flat_state()
        self.initialize_weights()
        self._build_flat(torch.device("cpu"))

    # ------------------------------------------------------------------------------------------
    # initialisation (mae.py:125-148; patch_embedding.py:107-124; Conv3d keeps torch's default)
    # ------------------------------------------------------------------------------------------
    def initialize_weights(self) -> None:
        D, Dd = self.encoder_embed_dim, self.decoder_embed_dim
        pe = self.patch_embedding
        with torch.no_grad():
            conv = pe.patch_embeddings
            nn.init.kaiming_uniform_(conv.weight, a=math.sqrt(5))  # torch Conv3d.reset_parameters
            fan_in = conv.weight[0].numel()
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(conv.bias, -bound, bound)
            if self.pos_embed == "learnable":
                nn.init.trunc_normal_(pe.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
            elif self.pos_embed == "sincos":
                pe.position_embeddings.copy_(build_sincos_position_embedding(self.grid_size, D, 3))
            if self.pos_embed == "sincos":
                self.decoder_pos_embed.copy_(build_sincos_position_embedding(self.grid_size, Dd, 3))
            else:
                nn.init.trunc_normal_(self.decoder_pos_embed, std=.02)
            nn.init.trunc_normal_(self.cls_token, std=.02)
            nn.init.trunc_normal_(self.decoder_cls_token, std=.02)
            nn.init.trunc_normal_(self.mask_token, std=.02)
            for name, m in self.named_modules():
                if not isinstance(m, _Affine) or m is conv:
                    continue
                if m.weight.dim() == 2:  # nn.Linear: xavier_uniform weight, zero bias (mae.py:143-146)
                    nn.init.xavier_uniform_(m.weight)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                else:  # nn.LayerNorm (mae.py:147-148)
                    nn.init.constant_(m.bias, 0)
                    nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------------------------------
    # public API (reference: mae.py:150-192, 303-317)
    # ------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None):
        if not x.is_cuda:
            raise HctError("MaskedAutoencoderViT (HIP) got a CPU tensor: this path has no CPU fallback")
        B = x.shape[0]
        expect = (B, self.in_chans) + tuple(self.input_size)
        if tuple(x.shape) != expect:
            raise HctError(f"input shape {tuple(x.shape)} != {expect} (run-time pos-embed interpolation is out of scope)")
        # fp16 volumes (the persistent cache's storage type, transforms.py:171-178) are consumed as they are: the patch gather
        # and the loss read them directly, which halves the two input passes of a step; anything else is taken as fp32
        x = x.contiguous() if x.dtype == torch.float16 else x.contiguous().float()
        if noise is None:
            noise = torch.rand(B, self.num_patches, device=x.device)  # mae.py:206
        noise = noise.contiguous().float()
        # a freshly zero_grad()-ed model (all .grad None) means the next backward overwrites
        if all(p.grad is None for p in self.parameters()):
            self._grad_overwrite = True
        # grad mode is read here: inside autograd.Function.forward it is always off
        loss = _MAEFunction.apply(self.cls_token, self, x, noise, torch.is_grad_enabled())
        return loss, None, None

    def activation(self, name: str, batch: int) -> torch.Tensor:
        """Named intermediate of the last forward (parity tests): e.g. 'latent', 'dec0.out', 'pred_full', 'mask'."""
        return self._plan_for(batch).activation(name)

    def last_pred(self, batch: int) -> torch.Tensor:
        """pred [B, L, pd] of the last forward (mae.py:272-273), fp32.  A training forward predicts the masked patches only
        (the loss takes no other row, mae.py:298-299) unless `model.full_pred = True`; forwards under `torch.no_grad()` always
        predict every patch."""
        plan = self._plan_for(batch)
        if getattr(plan, "tail", False):
            raise HctError("the last forward was a training forward, which predicts only the masked patches: set model.full_pred = True "
                           "(or run the forward under torch.no_grad()) to get the prediction of every patch")
        full = self.activation("pred_full", batch).float()
        return full.view(batch, self.num_patches + 1, -1)[:, 1:, :]

    def last_pred_masked(self, batch: int):
        """(rows, pred): prediction rows of the masked patches of the last forward and their row index b * (L + 1) + 1 + patch in
        the decoder layout -- available after every forward (in a training forward these are the only rows computed)."""
        plan = self._plan_for(batch)
        n = batch * (self.num_patches - self.len_keep)
        if getattr(plan, "tail", False):
            rows = plan.activation("tail_rows").view(-1)[:n].long()
            return rows, self.activation("pred_full", batch)[:n].float()
        ids_restore = self.activation("ids_restore", batch).long()
        rows = ((ids_restore >= self.len_keep).nonzero()[:, 0] * (self.num_patches + 1) + 1 + (ids_restore >= self.len_keep).nonzero()[:, 1])
        return rows, self.activation("pred_full", batch).float()[rows]

    def last_mask(self, batch: int) -> torch.Tensor:
        return self.activation("mask", batch)

    def patchify(self, x: torch.Tensor) -> torch.Tensor:
        B, Cc = x.shape[:2]
        gh, gw, gd = self.grid_size
        ph, pw, pd = self.patch_size
        x = x.reshape(B, Cc, gh, ph, gw, pw, gd, pd)
        return x.permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, gh * gw * gd, ph * pw * pd * Cc)

    def unpatchify(self, x: torch.Tensor, x_ori: torch.Tensor) -> torch.Tensor:
        """[B, L, pd] -> [B, C, H, W, D] "reconstructed voxels" (mae.py:172-192) via the HIP kernel."""
        B, Cc = x_ori.shape[:2]
        lib = _lib.load()
        if not x.is_cuda:
            raise HctError("unpatchify (HIP) needs a GPU tensor")
        src = x.contiguous()
        dt = HCT_BF16 if src.dtype == torch.bfloat16 else HCT_F32
        if dt == HCT_F32:
            src = src.float()
        vol = torch.empty((B, Cc) + tuple(self.input_size), dtype=torch.float32, device=x.device)
        _lib.check(lib.hct_unpatchify(src.data_ptr(), dt, 0, B, Cc, self.input_size[0], self.patch_size[0], vol.data_ptr(),
                                      _lib.stream_ptr()), "hct_unpatchify")
        return vol
