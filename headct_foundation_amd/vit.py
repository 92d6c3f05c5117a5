"""Plain ViT encoder for feature extraction (forward only) on the HIP kernels.

Mirror of `ViT` (src/models/vit.py:26-173) for the downstream use of a pre-trained encoder: same constructor arguments and
parameter names (`cls_token`, `register_tokens`, `patch_embedding.*`, `blocks.N.*`, `norm.*`), `forward(x) -> (x,
hidden_states_out)`: every patch embedded (+ position table, resized trilinearly when the volume is not the constructor's
size, patch_embedding.py:136-144), class token, register tokens, the blocks, final LayerNorm with eps 1e-6.  Built from the library's primitives (`hct_patch_gather`, `hct_gemm`, `hct_vit_assemble_fwd`,
`hct_layernorm_fwd`, `hct_attention_fwd`, `hct_head_linear`); there is no autograd and no CPU path.  With
`classification=True` the class-token head of vit.py:133-137 / :170-171 (Linear, Tanh unless `post_activation` says
otherwise) is applied and `forward` returns the class scores.  Not built: LoRA, 2-D inputs, the perceptron patch embedding.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch
import torch.nn as nn

from . import _lib
from .mae import _Affine, _Holder, _block, build_sincos_position_embedding


class ViT(nn.Module):
    def __init__(self, in_chans: int, img_size, patch_size, hidden_size: int = 768, mlp_dim: int = 3072, num_layers: int = 12,
                 num_heads: int = 12, patch_embed: str = "conv", pos_embed: str = "learnable", classification: bool = False,
                 num_classes: int = 2, dropout_rate: float = 0.0, spatial_dims: int = 3, num_register_tokens: int = 0,
                 post_activation: str = "Tanh", qkv_bias: bool = False, lora: bool = False, norm_layer=nn.LayerNorm,
                 compute_dtype: str = "bf16") -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise ValueError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise ValueError("hidden_size should be divisible by num_heads.")
        if lora or spatial_dims != 3 or patch_embed != "conv" or dropout_rate != 0.0 or norm_layer is not nn.LayerNorm:
            raise NotImplementedError("HIP ViT: forward only (lora=False, 3-D conv patch embedding, dropout 0, nn.LayerNorm)")
        if pos_embed not in ("learnable", "sincos", "none"):
            raise ValueError(f"pos_embed type {pos_embed} not supported.")
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        S = img_size if isinstance(img_size, int) else img_size[0]
        P = patch_size if isinstance(patch_size, int) else patch_size[0]
        if S % P:
            raise ValueError("patch_size should be divisible by img_size.")
        self.in_chans, self.S, self.P, self.D, self.mlp, self.heads = in_chans, S, P, hidden_size, mlp_dim, num_heads
        self.grid = S // P
        self.L = self.grid ** 3
        self.num_register_tokens = num_register_tokens
        self.compute_dtype = compute_dtype
        D = hidden_size
        self.patch_embedding = _Holder()
        self.patch_embedding.n_patches = self.L
        if pos_embed != "none":
            self.patch_embedding.position_embeddings = nn.Parameter(torch.zeros(1, self.L, D), requires_grad=pos_embed == "learnable")
        else:
            self.patch_embedding.position_embeddings = None
        self.patch_embedding.patch_embeddings = _Affine(D, in_chans, P, P, P, bias_shape=(D,))
        self.blocks = nn.ModuleList([_block(D, mlp_dim, qkv_bias) for _ in range(num_layers)])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.norm = _Affine(D, bias_shape=(D,))
        self.register_tokens = nn.Parameter(torch.zeros(1, num_register_tokens, D)) if num_register_tokens else None
        self.classification = classification
        self.post_activation = post_activation
        if classification:  # vit.py:133-137: Sequential(Linear, Tanh) -> keys `classification_head.0.*`, else a bare Linear
            head = _Affine(num_classes, D, bias_shape=(num_classes,))
            self.classification_head = nn.Sequential(head) if post_activation == "Tanh" else head
        with torch.no_grad():  # reference init: patch_embedding.py:112-130, nn.Linear / nn.LayerNorm defaults, vit.py:139-142
            if pos_embed == "learnable":
                nn.init.trunc_normal_(self.patch_embedding.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
            elif pos_embed == "sincos":
                self.patch_embedding.position_embeddings.copy_(build_sincos_position_embedding([self.grid] * 3, D, 3))
            for m in [self.norm] + [b_.att_norm for b_ in self.blocks] + [b_.ffn_norm for b_ in self.blocks]:
                m.weight.fill_(1.0)
                m.bias.zero_()
            import math
            lin = [pe_ for pe_ in [self.patch_embedding.patch_embeddings]]
            if classification:
                lin.append(self.classification_head[0] if post_activation == "Tanh" else self.classification_head)
            for b_ in self.blocks:
                lin += [b_.attn.qkv, b_.attn.proj, b_.mlp.linear1, b_.mlp.linear2]
            for m in lin:  # nn.Linear / nn.Conv3d defaults (the reference's ViT has no custom weight init)
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    fan_in = m.weight[0].numel()
                    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                    nn.init.uniform_(m.bias, -bound, bound)
            nn.init.normal_(self.cls_token, std=1e-6)
            if self.register_tokens is not None:
                nn.init.normal_(self.register_tokens, std=1e-6)
        self._wcache = {}

    # ---- low-level helpers over the C ABI -------------------------------------------------------
    def _dt(self):
        return _lib.HCT_BF16 if self.compute_dtype == "bf16" else _lib.HCT_F32

    def _tdt(self):
        return torch.bfloat16 if self.compute_dtype == "bf16" else torch.float32

    def _weight(self, p: torch.Tensor) -> torch.Tensor:
        """[out, in...] weight as a 2-D matrix in the compute dtype (cached bf16 copy, refreshed when the parameter changes)."""
        w2 = p.detach().reshape(p.shape[0], -1)
        if self.compute_dtype == "fp32":
            return w2
        key = id(p)
        ver = (p._version, p.data_ptr())
        hit = self._wcache.get(key)
        if hit is None or hit[0] != ver:
            dst = torch.empty(w2.shape, dtype=torch.bfloat16, device=p.device)
            _lib.check(self._lib.hct_cast(w2.data_ptr(), _lib.HCT_F32, dst.data_ptr(), _lib.HCT_BF16, w2.numel(), self._st), "hct_cast")
            hit = (ver, dst)
            self._wcache[key] = hit
        return hit[1]

    def _linear(self, a: torch.Tensor, w: torch.Tensor, bias, out_dtype, residual=None, act=0, aux=None) -> torch.Tensor:
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
        g = _lib.GemmArgs()
        g.M, g.N, g.K = M, N, K
        code = lambda t: _lib.HCT_BF16 if t.dtype == torch.bfloat16 else _lib.HCT_F32
        g.A, g.a_dtype, g.lda, g.transA = a.data_ptr(), code(a), K, 0
        g.B, g.b_dtype, g.ldb, g.transB = w.data_ptr(), code(w), K, 1
        g.C, g.c_dtype, g.ldc = out.data_ptr(), code(out), N
        if bias is not None:
            g.bias = bias.data_ptr()
        if residual is not None:
            g.residual, g.ldr = residual.data_ptr(), N
        g.act = act
        if aux is not None:
            g.aux, g.aux_dtype, g.ldaux = aux.data_ptr(), code(aux), N
        g.alpha = 1.0
        _lib.check(self._lib.hct_gemm(C.byref(g), None, 0, self._st), "hct_gemm")
        return out

    def _layernorm(self, h: torch.Tensor, ln, eps: float, out_dtype) -> torch.Tensor:
        rows, D = h.shape
        y = torch.empty(rows, D, dtype=out_dtype, device=h.device)
        mean = torch.empty(rows, dtype=torch.float32, device=h.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=h.device)
        code = _lib.HCT_BF16 if out_dtype == torch.bfloat16 else _lib.HCT_F32
        _lib.check(self._lib.hct_layernorm_fwd(h.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), rows, D, eps, y.data_ptr(), code,
                                               mean.data_ptr(), rstd.data_ptr(), self._st), "hct_layernorm_fwd")
        return y

    # ---- forward ---------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        if not x.is_cuda or not self.cls_token.is_cuda:
            raise _lib.HctError("ViT (HIP) runs on the GPU: move the module and the input to 'cuda' (no CPU fallback exists)")
        B = x.shape[0]
        S = x.shape[-1] if x.dim() == 5 else -1
        if x.dim() != 5 or tuple(x.shape[1:]) != (self.in_chans, S, S, S) or S <= 0 or S % self.P:
            raise _lib.HctError(f"input shape {tuple(x.shape)} != (B, {self.in_chans}, S, S, S) with S a multiple of {self.P}")
        self._lib = _lib.load()
        dev = x.device
        with torch.cuda.device(dev):
            self._st = torch.cuda.current_stream().cuda_stream
            dt, tdt = self._dt(), self._tdt()
            grid = S // self.P
            D, L, R, H = self.D, grid ** 3, self.num_register_tokens, self.heads
            T = 1 + R + L
            # fp16 volumes (the persistent cache's storage type) are read as they are; anything else goes through fp32
            x = x.contiguous() if x.dtype == torch.float16 else x.to(torch.float32).contiguous()
            xdt = _lib.HCT_F16 if x.dtype == torch.float16 else _lib.HCT_F32
            rows = torch.empty(B * L, self.in_chans * self.P ** 3, dtype=tdt, device=dev)
            # (no index table: every patch in grid order -- the library then moves whole pencils of patches)
            _lib.check(self._lib.hct_patch_gather(x.data_ptr(), xdt, None, B, self.in_chans, S, self.P, L, L, rows.data_ptr(), dt,
                                                  self._st), "hct_patch_gather")
            pe = self.patch_embedding
            tok = self._linear(rows, self._weight(pe.patch_embeddings.weight), pe.patch_embeddings.bias, tdt)
            h = torch.empty(B * T, D, dtype=torch.float32, device=dev)
            pos = pe.position_embeddings
            if pos is not None and grid != self.grid:
                # a volume of another size: the position table is resized trilinearly for this call, as
                # PatchEmbeddingBlock.forward does (patch_embedding.py:136-144 -> pos_embed.py:164-217)
                resized = torch.empty(1, L, D, dtype=torch.float32, device=dev)
                _lib.check(self._lib.hct_pos_embed_interp3d(pos.data_ptr(), self.grid, resized.data_ptr(), grid, D, 0, self._st),
                           "hct_pos_embed_interp3d")
                pos = resized
            _lib.check(self._lib.hct_vit_assemble_fwd(tok.data_ptr(), dt, self.cls_token.data_ptr(),
                                                      self.register_tokens.data_ptr() if R else None,
                                                      pos.data_ptr() if pos is not None else None, B, L, R, D, h.data_ptr(), self._st),
                       "hct_vit_assemble_fwd")
            hidden: List[torch.Tensor] = []
            for blk in self.blocks:  # AttentionBlock.forward, attentionblock.py:96-99
                xn = self._layernorm(h, blk.att_norm, 1e-5, tdt)
                qkv = self._linear(xn, self._weight(blk.attn.qkv.weight), getattr(blk.attn.qkv, "bias", None), tdt)
                o = torch.empty(B * T, D, dtype=tdt, device=dev)
                lse = torch.empty(B * H * T, dtype=torch.float32, device=dev)
                _lib.check(self._lib.hct_attention_fwd(qkv.data_ptr(), B, T, H, D // H, dt, o.data_ptr(), lse.data_ptr(), self._st),
                           "hct_attention_fwd")
                h_mid = self._linear(o, self._weight(blk.attn.proj.weight), blk.attn.proj.bias, torch.float32, residual=h)
                xn = self._layernorm(h_mid, blk.ffn_norm, 1e-5, tdt)
                pre = torch.empty(B * T, self.mlp, dtype=tdt, device=dev)
                g = self._linear(xn, self._weight(blk.mlp.linear1.weight), blk.mlp.linear1.bias, tdt, act=_lib.HCT_ACT_GELU, aux=pre)
                h = self._linear(g, self._weight(blk.mlp.linear2.weight), blk.mlp.linear2.bias, torch.float32, residual=h_mid)
                hidden.append(h.view(B, T, D))
            out = self._layernorm(h, self.norm, 1e-6, torch.float32).view(B, T, D)
            if self.classification:  # classification_head(x[:, 0]), vit.py:170-171
                tanh = self.post_activation == "Tanh"
                head = self.classification_head[0] if tanh else self.classification_head
                ncls = head.weight.shape[0]
                scores = torch.empty(B, ncls, dtype=torch.float32, device=dev)
                _lib.check(self._lib.hct_head_linear(out.data_ptr(), T * D, 1, None, None, 0.0, head.weight.data_ptr(), head.bias.data_ptr(),
                                                     _lib.HCT_ACT_TANH if tanh else _lib.HCT_ACT_NONE, scores.data_ptr(), B, D, ncls,
                                                     self._st), "hct_head_linear")
                out = scores
        return out, hidden
