"""CPU oracle for the MAE pre-training hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32, no MONAI/timm/yacs) restatement of the
reference algorithm for one MAE pre-training step.  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product path
(``headct_foundation_amd``) never imports anything from ``oracle/``.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4).  The oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py``
(which imports the reference's own ``src/models/mae.py`` etc. with stand-ins
for the seven absent MONAI/timm symbols) and committed under ``tests/golden``.

Every function cites the reference file:line it follows (paths relative to
the reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class MAEConfig:
    """Constructor surface of MaskedAutoencoderViT (src/models/mae.py:22-42)."""
    input_size: int = 96
    patch_size: int = 16
    mask_ratio: float = 0.75
    in_chans: int = 1
    pos_embed: str = "sincos"
    encoder_depth: int = 12
    encoder_embed_dim: int = 768
    encoder_mlp_dim: int = 3072
    encoder_num_heads: int = 12
    decoder_depth: int = 8
    decoder_embed_dim: int = 768
    decoder_mlp_dim: int = 3072
    decoder_num_heads: int = 16
    norm_pix_loss: bool = False
    use_bias: bool = False

    @property
    def grid(self) -> int:
        return self.input_size // self.patch_size

    @property
    def num_patches(self) -> int:  # L
        return self.grid ** 3

    @property
    def patch_dim(self) -> int:  # pd = C * P^3
        return self.in_chans * self.patch_size ** 3

    @property
    def len_keep(self) -> int:  # mae.py:205
        return int(self.num_patches * (1 - self.mask_ratio))

    def ctor_kwargs(self) -> dict:
        return dict(
            input_size=self.input_size, patch_size=self.patch_size, mask_ratio=self.mask_ratio,
            in_chans=self.in_chans, dropout_rate=0.0, spatial_dims=3, patch_embed="conv",
            pos_embed=self.pos_embed, encoder_depth=self.encoder_depth,
            encoder_embed_dim=self.encoder_embed_dim, encoder_mlp_dim=self.encoder_mlp_dim,
            encoder_num_heads=self.encoder_num_heads, decoder_depth=self.decoder_depth,
            decoder_embed_dim=self.decoder_embed_dim, decoder_mlp_dim=self.decoder_mlp_dim,
            decoder_num_heads=self.decoder_num_heads, norm_pix_loss=self.norm_pix_loss,
            use_bias=self.use_bias)


# Named configurations used by tests / bench (BASELINE.json configs).
CONFIGS: Dict[str, MAEConfig] = {
    # tiny full-tensor fixture case (not in BASELINE; small enough to commit whole)
    "micro": MAEConfig(input_size=32, patch_size=8, encoder_depth=2, encoder_embed_dim=48,
                       encoder_mlp_dim=96, encoder_num_heads=3, decoder_depth=1,
                       decoder_embed_dim=48, decoder_mlp_dim=96, decoder_num_heads=3,
                       use_bias=True, in_chans=1),
    # BASELINE config #1: ViT-Tiny, 2-layer decoder, 64^3, patch 16
    "tiny": MAEConfig(input_size=64, patch_size=16, encoder_depth=12, encoder_embed_dim=192,
                      encoder_mlp_dim=768, encoder_num_heads=3, decoder_depth=2,
                      decoder_embed_dim=192, decoder_mlp_dim=768, decoder_num_heads=3),
    # cut-down config #2: real ViT-B tile shapes (N=55/217, dh=64/48), depth 1+1
    "vitb_cut": MAEConfig(input_size=96, patch_size=16, encoder_depth=1, decoder_depth=1),
    # BASELINE config #2/#3: ViT-B/16^3, 96^3
    "vitb": MAEConfig(input_size=96, patch_size=16),
    # BASELINE config #4: ViT-L, 128^3, learnable pos-embed (1024 % 6 != 0, pos_embed.py:60)
    "vitl": MAEConfig(input_size=128, patch_size=16, pos_embed="learnable", encoder_depth=24,
                      encoder_embed_dim=1024, encoder_mlp_dim=4096, encoder_num_heads=16),
    # multi-channel / patch-12 shape of the shipped yaml (configs/mae/mae_HeadCT.yaml:31-50), cut down
    "yaml_cut": MAEConfig(input_size=48, patch_size=12, in_chans=3, encoder_depth=1,
                          encoder_embed_dim=96, encoder_mlp_dim=192, encoder_num_heads=3,
                          decoder_depth=1, decoder_embed_dim=96, decoder_mlp_dim=192,
                          decoder_num_heads=2, use_bias=True, norm_pix_loss=True),
}


# ----------------------------------------------------------------------------
# parameters: names / shapes (probe-dumped from the reference)
# ----------------------------------------------------------------------------
def param_shapes(cfg: MAEConfig) -> List[Tuple[str, Tuple[int, ...], bool]]:
    """(name, shape, requires_grad) in the reference's registration order
    (src/models/mae.py:86-121, attentionblock.py:91-94, MONAI MLPBlock linear1/linear2)."""
    D, Dd = cfg.encoder_embed_dim, cfg.decoder_embed_dim
    L, P, C = cfg.num_patches, cfg.patch_size, cfg.in_chans
    out: List[Tuple[str, Tuple[int, ...], bool]] = [
        ("cls_token", (1, 1, D), True),
        ("decoder_cls_token", (1, 1, Dd), True),
        ("decoder_pos_embed", (1, L, Dd), False),  # mae.py:92 requires_grad=False
        ("mask_token", (1, 1, Dd), True),
    ]
    if cfg.pos_embed != "none":
        out.append(("patch_embedding.position_embeddings", (1, L, D), True))
    out += [
        ("patch_embedding.patch_embeddings.weight", (D, C, P, P, P), True),
        ("patch_embedding.patch_embeddings.bias", (D,), True),
    ]

    def block(prefix: str, d: int, m: int):
        r = [
            (f"{prefix}.mlp.linear1.weight", (m, d), True),
            (f"{prefix}.mlp.linear1.bias", (m,), True),
            (f"{prefix}.mlp.linear2.weight", (d, m), True),
            (f"{prefix}.mlp.linear2.bias", (d,), True),
            (f"{prefix}.att_norm.weight", (d,), True),
            (f"{prefix}.att_norm.bias", (d,), True),
            (f"{prefix}.ffn_norm.weight", (d,), True),
            (f"{prefix}.ffn_norm.bias", (d,), True),
            (f"{prefix}.attn.qkv.weight", (3 * d, d), True),
        ]
        if cfg.use_bias:
            r.append((f"{prefix}.attn.qkv.bias", (3 * d,), True))
        r += [
            (f"{prefix}.attn.proj.weight", (d, d), True),
            (f"{prefix}.attn.proj.bias", (d,), True),
        ]
        return r

    for i in range(cfg.encoder_depth):
        out += block(f"blocks.{i}", D, cfg.encoder_mlp_dim)
    for i in range(cfg.decoder_depth):
        out += block(f"decoder_blocks.{i}", Dd, cfg.decoder_mlp_dim)
    out += [("norm.weight", (D,), True), ("norm.bias", (D,), True),
            ("decoder_norm.weight", (Dd,), True), ("decoder_norm.bias", (Dd,), True),
            ("decoder_embed.weight", (Dd, D), True)]
    if cfg.use_bias:
        out.append(("decoder_embed.bias", (Dd,), True))
    out.append(("decoder_pred.weight", (cfg.patch_dim, Dd), True))
    if cfg.use_bias:
        out.append(("decoder_pred.bias", (cfg.patch_dim,), True))
    return out


def build_sincos_position_embedding_3d(grid: int, embed_dim: int, temperature: float = 10000.0) -> torch.Tensor:
    """src/utils/pos_embed.py:51-78 (3-D branch; cubic grid so the h/w naming swap :54-55 is moot)."""
    assert embed_dim % 6 == 0, "Embed dimension must be divisible by 6 for 3D sin-cos position embedding"
    g = torch.arange(grid, dtype=torch.float32)
    gh, gw, gd = torch.meshgrid(g, g, g, indexing="ij")
    pos_dim = embed_dim // 6
    omega = torch.arange(pos_dim, dtype=torch.float32) / pos_dim
    omega = 1.0 / (temperature ** omega)
    oh = torch.einsum("m,d->md", [gh.flatten(), omega])
    ow = torch.einsum("m,d->md", [gw.flatten(), omega])
    od = torch.einsum("m,d->md", [gd.flatten(), omega])
    # reference concatenates (out_w, out_h, out_d) where out_h is built from the FIRST meshgrid
    # output and out_w from the SECOND (pos_embed.py:58,65-76)
    return torch.cat([torch.sin(ow), torch.cos(ow), torch.sin(oh), torch.cos(oh),
                      torch.sin(od), torch.cos(od)], dim=1)[None]


def interpolate_pos_embed_3d(pos: torch.Tensor, new_grid: int, num_extra_tokens: int = 0) -> torch.Tensor:
    """Trilinear resize of a learnable position table [1, extra + g^3, D] to [1, extra + new_grid^3, D]
    (src/utils/pos_embed.py:102-153, 3-D branch: F.interpolate(mode='trilinear', align_corners=False) on the
    [1, D, g, g, g] view; extra (class) tokens are kept).  Written out with explicit index arithmetic: source coordinate
    max(0, (dst + 0.5) * g/new - 0.5) in fp32, neighbours i0 = floor, i1 = min(i0 + 1, g - 1), weight = coordinate - i0;
    the three axes are applied innermost (depth) first, as ATen's upsample_trilinear3d accumulates them."""
    extra, tok = pos[:, :num_extra_tokens], pos[:, num_extra_tokens:]
    D = tok.shape[-1]
    g = int(round(tok.shape[1] ** (1.0 / 3.0)))
    assert g ** 3 == tok.shape[1], "position table is not a cubic grid"
    if g == new_grid:
        return pos.clone()
    src = tok.reshape(g, g, g, D).to(torch.float32)
    scale = np.float32(g) / np.float32(new_grid)
    c = np.maximum(np.float32(0), (np.arange(new_grid, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)).astype(np.float32)
    i0 = np.floor(c).astype(np.int64)
    i1 = np.minimum(i0 + 1, g - 1)
    w1 = torch.from_numpy((c - i0.astype(np.float32)).astype(np.float32))
    w0 = 1.0 - w1
    i0t, i1t = torch.from_numpy(i0), torch.from_numpy(i1)
    out = torch.zeros(new_grid, new_grid, new_grid, D, dtype=torch.float32)
    for a, (ia, wa) in enumerate(((i0t, w0), (i1t, w1))):
        for b, (ib, wb) in enumerate(((i0t, w0), (i1t, w1))):
            for c_, (ic, wc) in enumerate(((i0t, w0), (i1t, w1))):
                out += (wa[:, None, None, None] * wb[None, :, None, None] * wc[None, None, :, None]) * src[ia][:, ib][:, :, ic]
    return torch.cat([extra, out.reshape(1, new_grid ** 3, D)], dim=1)


def augment_volume(x: torch.Tensor, flip: Optional[Sequence[int]] = None, shift: Optional[Sequence[float]] = None) -> torch.Tensor:
    """Per-sample MAE input transforms with the random draws made explicit (src/data/transforms.py:193-228):
    CastToTyped(float32) -> RandFlipd(spatial_axis=0), (1), (2) -> RandShiftIntensityd (img + offset, no clipping).
    x: [B, C, S, S, S] of any float dtype (the persistent cache stores fp16, transforms.py:170-175);
    flip[b]: bit a set = spatial axis a flipped; shift[b]: the drawn offset (0 when the transform did not fire)."""
    out = x.to(torch.float32).clone()
    for b in range(x.shape[0]):
        f = int(flip[b]) if flip is not None else 0
        dims = [1 + a for a in range(3) if f & (1 << a)]
        v = out[b]
        if dims:
            v = torch.flip(v, dims)
        out[b] = v + (np.float32(shift[b]) if shift is not None else np.float32(0))
    return out


def gaussian_kernel_1d(sigma: float) -> torch.Tensor:
    """MONAI `gaussian_1d(sigma, truncated=4.0, approx="erf", normalize=False)` -- the kernel GaussianFilter builds for
    GaussianSmooth / RandGaussianSmoothd (src/data/transforms.py:230-238): tail = int(max(4 sigma, 0.5) + 0.5) taps either side,
    w(x) = 0.5 (erf(t (x + 0.5)) - erf(t (x - 0.5))) with t = 0.70710678 / |sigma|, clamped at 0; fp32.  (Stated from knowledge
    of MONAI 1.2 / 1.3: MONAI is not installed here, so parity with MONAI itself is unpinned.)"""
    sg = torch.tensor(float(sigma), dtype=torch.float32)
    tail = int(max(float(sg) * 4.0, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / torch.abs(sg)
    return (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)


def gaussian_smooth3d(x: torch.Tensor, sigma: Sequence[Sequence[float]], apply: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """RandGaussianSmoothd with its draws made explicit: sample b of x [B, C, S0, S1, S2] is filtered along each spatial axis a with
    gaussian_kernel_1d(sigma[b][a]) under zero padding (MONAI separable_filtering, mode "zeros") when apply[b], else left as is."""
    out = x.to(torch.float32).clone()
    for b in range(x.shape[0]):
        if apply is not None and not apply[b]:
            continue
        v = out[b].unsqueeze(0)  # [1, C, S0, S1, S2]
        C = v.shape[1]
        for a in range(3):
            k = gaussian_kernel_1d(sigma[b][a])
            shape = [1, 1, 1, 1, 1]
            shape[2 + a] = k.numel()
            pad = [0, 0, 0]
            pad[a] = (k.numel() - 1) // 2
            v = F.conv3d(v, k.reshape(shape).repeat(C, 1, 1, 1, 1), padding=pad, groups=C)
        out[b] = v[0]
    return out


def hu_window(hu: torch.Tensor, in_channels: int = 1) -> torch.Tensor:
    """Windowing step of loading_transforms (src/data/transforms.py:108-133): one channel = ScaleIntensityRanged(a_min 40-150,
    a_max 40+150, b 0..1, clip); three = MultipleWindowScaleStack (:8-36) over (centre, width) (40,80), (80,200), (600,2800)
    with a_min = l - w // 2, a_max = l + w // 2, concatenated on the channel axis.  MONAI's ScaleIntensityRange computes
    (img - a_min) / (a_max - a_min) * (b_max - b_min) + b_min and clips to [b_min, b_max] in fp32 (stated from knowledge of
    MONAI 1.2 / 1.3: MONAI is not installed here, so parity with MONAI itself is unpinned).  hu [B, 1, ...] -> [B, C, ...]."""
    if in_channels == 1:
        wins = [(40 - 150, 40 + 150)]
    elif in_channels == 3:
        wins = [(l - w // 2, l + w // 2) for l, w in ((40, 80), (80, 200), (600, 2800))]
    else:
        raise NotImplementedError(f"Channel size {in_channels} is not implemented.")
    x = hu.float()
    outs = []
    for a_min, a_max in wins:
        y = (x - float(a_min)) / float(a_max - a_min)
        y = y * (1.0 - 0.0) + 0.0
        outs.append(torch.clamp(y, 0.0, 1.0))
    return torch.cat(outs, dim=1)


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """Portable deterministic U[-1,1) stream (integer hash; independent of any RNG library)."""
    i = np.arange(n, dtype=np.uint64)
    x = i + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)  # wraps mod 2^64
    x ^= x >> np.uint64(30)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(27)
    x = (x * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31)
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24-bit mantissa: exact in fp32
    return (2.0 * u - 1.0).astype(np.float32)


def make_params(cfg: MAEConfig, seed: int = 0, generic: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic, library-independent parameters for parity fixtures.

    generic=True perturbs biases / LayerNorm affine away from their (0, 1) init values so every
    term of the forward/backward is exercised.  Scales follow the reference init
    (mae.py:125-148: xavier-uniform linears, std .02 tokens; conv keeps kaiming-uniform default).
    """
    out: Dict[str, torch.Tensor] = {}
    for k, (name, shape, _) in enumerate(param_shapes(cfg)):
        n = int(np.prod(shape))
        u = torch.from_numpy(hash_uniform(n, seed * 1000 + k + 1)).reshape(shape)
        if name == "decoder_pos_embed" or name == "patch_embedding.position_embeddings":
            if cfg.pos_embed == "sincos":
                d = shape[-1]
                t = build_sincos_position_embedding_3d(cfg.grid, d)
                if name.startswith("patch") and generic:
                    t = t + 0.02 * u  # trainable in the reference (patch_embedding.py:109,117-120)
            else:
                t = 0.02 * u
        elif name.endswith("token"):
            t = 0.02 * 1.7 * u
        elif name.endswith("norm.weight"):
            t = 1.0 + (0.1 * u if generic else 0.0 * u)
        elif name.endswith("bias"):
            t = 0.05 * u if generic else 0.0 * u
            if name == "patch_embedding.patch_embeddings.bias":
                t = u / math.sqrt(cfg.patch_dim)
        elif name == "patch_embedding.patch_embeddings.weight":
            t = u / math.sqrt(cfg.patch_dim)  # kaiming_uniform(a=sqrt(5)) bound = 1/sqrt(fan_in)
        else:  # linear weight: xavier-uniform bound
            fan_out, fan_in = shape
            t = u * math.sqrt(6.0 / (fan_in + fan_out))
        out[name] = t.contiguous().float()
    return out


def make_volume(cfg: MAEConfig, batch: int, seed: int = 0) -> torch.Tensor:
    """Synthetic CT volume batch U[0,1) (windowed-CT value range, src/data/transforms.py:120-128)."""
    S, C = cfg.input_size, cfg.in_chans
    u = hash_uniform(batch * C * S * S * S, 7777 + seed)
    return torch.from_numpy((u + 1.0) * 0.5).reshape(batch, C, S, S, S).contiguous()


def make_noise(cfg: MAEConfig, batch: int, seed: int = 0) -> torch.Tensor:
    """Tie-free masking noise in [0,1): a per-row permutation scaled to (0,1) (SURVEY 8a a5:
    torch.argsort is unstable, so fixtures must not contain ties)."""
    L = cfg.num_patches
    u = hash_uniform(batch * L, 4242 + seed).reshape(batch, L)
    order = np.argsort(u, axis=1, kind="stable")
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(L)[None, :].repeat(batch, 0), axis=1)
    return torch.from_numpy(((rank.astype(np.float32) + 0.5) / L).astype(np.float32))


# ----------------------------------------------------------------------------
# forward (functional; parameters in a dict keyed by the reference's names)
# ----------------------------------------------------------------------------
def patchify(cfg: MAEConfig, x: torch.Tensor) -> torch.Tensor:
    """src/models/mae.py:150-170: [B,C,H,W,D] -> [B, L, ph*pw*pd*C] (C fastest)."""
    B, C = x.shape[:2]
    g, p = cfg.grid, cfg.patch_size
    x = x.reshape(B, C, g, p, g, p, g, p)
    return x.permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, g ** 3, p ** 3 * C)


def unpatchify(cfg: MAEConfig, x: torch.Tensor) -> torch.Tensor:
    """src/models/mae.py:172-192 (inverse of patchify; "reconstructed voxels")."""
    B = x.shape[0]
    g, p, C = cfg.grid, cfg.patch_size, cfg.in_chans
    x = x.reshape(B, g, g, g, p, p, p, C)
    return x.permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, C, g * p, g * p, g * p)


def random_masking_from_noise(cfg: MAEConfig, noise: torch.Tensor):
    """src/models/mae.py:204-216 with the noise supplied (stable argsort: ties -> lower index first)."""
    K = cfg.len_keep
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :K]
    mask = torch.ones(noise.shape, dtype=torch.float32)
    mask[:, :K] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return ids_shuffle, ids_restore, ids_keep, mask


# bf16-storage emulation (tests only): with `emulate_bf16` the restatement rounds to bfloat16 exactly where the HIP path's bf16
# mode stores bfloat16 -- GEMM operand copies of the weights, LayerNorm outputs, qkv, softmax probabilities, attention output,
# pre-GELU / GELU activations, patch rows, tokens, latent, decoder input rows, prediction -- and keeps fp32 everywhere the HIP
# path does (residual stream, statistics, loss, master weights, gradients of parameters).  Comparing the HIP bf16 path with
# THIS oracle isolates logic from rounding: what is left is accumulation order and the rounding of backward intermediates.
_EMU = [False]


def _r(t: torch.Tensor) -> torch.Tensor:
    return t.bfloat16().float() if _EMU[0] else t


def _layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)  # nn.LayerNorm default eps (attentionblock.py:92-93)


def _block(p: Dict[str, torch.Tensor], prefix: str, h: torch.Tensor, heads: int, inter: Optional[dict], tag: str):
    """AttentionBlock.forward attentionblock.py:96-99; SelfAttention.forward :51-66; MONAI MLPBlock (exact-erf GELU)."""
    B, N, D = h.shape
    x1 = _r(_layer_norm(h, p[f"{prefix}.att_norm.weight"], p[f"{prefix}.att_norm.bias"]))
    qkv = _r(F.linear(x1, _r(p[f"{prefix}.attn.qkv.weight"]), p.get(f"{prefix}.attn.qkv.bias")))
    qkv = qkv.reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-1, -2)) * (D // heads) ** -0.5, dim=-1)  # SDPA, no mask, dropout 0 (:61)
    y = _r((_r(att) @ v).transpose(1, 2).contiguous().view(B, N, D))
    y = F.linear(y, _r(p[f"{prefix}.attn.proj.weight"]), p[f"{prefix}.attn.proj.bias"])
    h = h + y
    x2 = _r(_layer_norm(h, p[f"{prefix}.ffn_norm.weight"], p[f"{prefix}.ffn_norm.bias"]))
    u = F.linear(x2, _r(p[f"{prefix}.mlp.linear1.weight"]), p[f"{prefix}.mlp.linear1.bias"])
    gact = _r(F.gelu(u))  # nn.GELU() default = exact erf
    h = h + F.linear(gact, _r(p[f"{prefix}.mlp.linear2.weight"]), p[f"{prefix}.mlp.linear2.bias"])
    if inter is not None:
        inter[f"{tag}.out"] = h
    return h


def forward(cfg: MAEConfig, p: Dict[str, torch.Tensor], x: torch.Tensor, noise: torch.Tensor,
            want_inter: bool = False, emulate_bf16: bool = False):
    """MaskedAutoencoderViT.forward src/models/mae.py:303-317.  Returns (loss, pred, mask, inter)."""
    _EMU[0] = bool(emulate_bf16)
    try:
        return _forward(cfg, p, x, noise, want_inter)
    finally:
        _EMU[0] = False


def _forward(cfg: MAEConfig, p: Dict[str, torch.Tensor], x: torch.Tensor, noise: torch.Tensor, want_inter: bool = False):
    inter: Optional[dict] = {} if want_inter else None
    B = x.shape[0]
    D, Dd, L, K = cfg.encoder_embed_dim, cfg.decoder_embed_dim, cfg.num_patches, cfg.len_keep
    P = cfg.patch_size
    # --- forward_encoder mae.py:220-242 ---
    # PatchEmbeddingBlock.forward patch_embedding.py:149-156: Conv3d(k=s=P) -> flatten(2).transpose -> + pos
    tok = F.conv3d(_r(x), _r(p["patch_embedding.patch_embeddings.weight"]), p["patch_embedding.patch_embeddings.bias"], stride=P)
    tok = _r(tok.flatten(2).transpose(-1, -2))
    if "patch_embedding.position_embeddings" in p:
        tok = tok + p["patch_embedding.position_embeddings"]
    ids_shuffle, ids_restore, ids_keep, mask = random_masking_from_noise(cfg, noise)
    xm = torch.gather(tok, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))  # mae.py:212
    h = torch.cat((p["cls_token"].expand(B, -1, -1), xm), dim=1)  # mae.py:233-234
    if inter is not None:
        inter.update(patch_embed=tok, ids_restore=ids_restore, ids_keep=ids_keep, mask=mask, enc_in=h)
    for i in range(cfg.encoder_depth):
        h = _block(p, f"blocks.{i}", h, cfg.encoder_num_heads, inter, f"enc{i}")
    latent = _r(_layer_norm(h, p["norm.weight"], p["norm.bias"]))  # mae.py:240
    # --- forward_decoder mae.py:244-275 ---
    y = _r(F.linear(latent, _r(p["decoder_embed.weight"]), p.get("decoder_embed.bias")))
    mask_tokens = p["mask_token"].repeat(B, L + 1 - y.shape[1], 1)
    y_ = torch.cat([y[:, 1:, :], mask_tokens], dim=1)
    y_ = torch.gather(y_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, Dd))
    y = torch.cat([y[:, :1, :], y_], dim=1)
    dpe = torch.cat((p["decoder_cls_token"].expand(B, -1, -1), p["decoder_pos_embed"].expand(B, -1, -1)), dim=1)
    y = y + dpe  # decoder_cls_token is ADDED to position 0 (mae.py:262-265)
    if inter is not None:
        inter.update(latent=latent, dec_in=y)
    for i in range(cfg.decoder_depth):
        y = _block(p, f"decoder_blocks.{i}", y, cfg.decoder_num_heads, inter, f"dec{i}")
    y = _r(_layer_norm(y, p["decoder_norm.weight"], p["decoder_norm.bias"]))
    pred = _r(F.linear(y, _r(p["decoder_pred.weight"]), p.get("decoder_pred.bias")))[:, 1:, :]
    # --- forward_loss mae.py:277-301 ---
    target = patchify(cfg, x)
    if cfg.norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)  # unbiased (mae.py:292)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
    loss = ((pred - target) ** 2).mean(dim=-1)
    loss = (loss * mask).sum() / mask.sum()
    if inter is not None:
        inter.update(pred=pred, target=target)
    return loss, pred, mask, inter


def make_vit_params(shapes: Dict[str, Sequence[int]], seed0: int = 100) -> Dict[str, torch.Tensor]:
    """Deterministic non-trivial values for a ViT state dict (fixtures): hash stream i per tensor in sorted key order,
    x0.02 for matrices, x0.1 for vectors, +1 on LayerNorm weights; BatchNorm buffers of the classifier heads:
    running_var = 0.5 + |u|, num_batches_tracked = 0."""
    out = {}
    for i, n in enumerate(sorted(shapes)):
        shp = tuple(int(v) for v in shapes[n])
        if n.endswith("num_batches_tracked"):
            out[n] = torch.zeros(shp, dtype=torch.int64)
            continue
        numel = int(np.prod(shp))
        scale = 0.02 if len(shp) > 1 else 0.1
        t = torch.from_numpy(hash_uniform(numel, seed0 + i).reshape(shp).astype(np.float32)) * scale
        if n.endswith("norm.weight"):
            t = t + 1.0
        if n.endswith("running_var"):
            t = 0.5 + (t / scale).abs()
        out[n] = t
    return out


def vit_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, patch_size: int, num_heads: int, num_layers: int):
    """ViT.forward, src/models/vit.py:144-173: embed EVERY patch (+ position table), prepend the class token, insert the
    register tokens behind it (:150-160), run the blocks collecting each output, final LayerNorm with eps 1e-6 (:124) and,
    when the state dict holds a `classification_head`, that head on the class token.  Returns (x, hidden_states_out)."""
    B = x.shape[0]
    tok = F.conv3d(x, p["patch_embedding.patch_embeddings.weight"], p["patch_embedding.patch_embeddings.bias"], stride=patch_size)
    tok = tok.flatten(2).transpose(-1, -2)
    if "patch_embedding.position_embeddings" in p:
        pos = p["patch_embedding.position_embeddings"]
        if pos.shape[1] != tok.shape[1]:  # volume of another size: patch_embedding.py:136-144 -> pos_embed.py:164-217
            pos = interpolate_pos_embed_3d(pos, round(tok.shape[1] ** (1.0 / 3.0)), 0)
        tok = tok + pos
    h = torch.cat((p["cls_token"].expand(B, -1, -1), tok), dim=1)
    if "register_tokens" in p:
        h = torch.cat((h[:, :1], p["register_tokens"].expand(B, -1, -1), h[:, 1:]), dim=1)
    hidden = []
    for i in range(num_layers):
        h = _block(p, f"blocks.{i}", h, num_heads, None, "")
        hidden.append(h)
    out = F.layer_norm(h, (h.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-6)
    if "classification_head.0.weight" in p:  # nn.Sequential(Linear, Tanh) on the class token, vit.py:133-135, :170-171
        out = torch.tanh(out[:, 0] @ p["classification_head.0.weight"].T + p["classification_head.0.bias"])
    elif "classification_head.weight" in p:  # post_activation != "Tanh", vit.py:136-137
        out = out[:, 0] @ p["classification_head.weight"].T + p["classification_head.bias"]
    return out, hidden


def batchnorm1d_eval(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """nn.BatchNorm1d(dim, affine=False, eps=1e-6) in eval mode (classifier.py:18, :65-66), channels on the LAST axis here
    (the reference transposes to put them on axis 1 and back, classifier.py:89, :96)."""
    return (x - mean) / torch.sqrt(var + eps)


def linear_classifier_forward(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """LinearClassifier.forward, classifier.py:21-33 (eval): bn -> linear.  x: [B, dim]."""
    x = batchnorm1d_eval(x, p["bn.running_mean"], p["bn.running_var"])
    return x @ p["linear.weight"].T + p["linear.bias"]


def linear_probe_step(p: Dict[str, torch.Tensor], x: torch.Tensor, target: torch.Tensor, momentum: float = 0.1):
    """One linear-probing step on frozen features (engine_downstream.py:80-102 with TRAIN.LOCK): LinearClassifier in
    TRAINING mode (classifier.py:21-33: BatchNorm1d batch statistics, biased variance; running statistics updated with the
    unbiased variance and momentum 0.1), nn.CrossEntropyLoss() (main_downstream.py:214), backward.
    Returns (logits, loss, {linear.weight, linear.bias gradients}, {running_mean, running_var after the update})."""
    w = p["linear.weight"].clone().requires_grad_(True)
    b = p["linear.bias"].clone().requires_grad_(True)
    mean = x.mean(dim=0)
    xn = (x - mean) / torch.sqrt(x.var(dim=0, unbiased=False) + 1e-6)
    logits = xn @ w.T + b
    loss = (torch.logsumexp(logits, dim=1) - logits[torch.arange(x.shape[0]), target]).mean()
    loss.backward()
    stats = {"bn.running_mean": (1 - momentum) * p["bn.running_mean"] + momentum * mean,
             "bn.running_var": (1 - momentum) * p["bn.running_var"] + momentum * x.var(dim=0, unbiased=True)}
    return logits.detach(), loss.detach(), {"linear.weight": w.grad, "linear.bias": b.grad}, stats


def attention_classifier_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, num_heads: int, num_queries: int,
                                 qk_scale: Optional[float] = None) -> torch.Tensor:
    """AttentionClassifier.forward, classifier.py:73-99 (eval).  x: [B, N, C] token features.  The learnt queries are scaled
    by `scale` (:86) and F.scaled_dot_product_attention scales the logits by dh^-1/2 once more (:93); its output
    [B, H, Q, dh] is reshaped to [B, Q, C] without a permute (:95)."""
    B, N, C = x.shape
    dh = C // num_heads
    scale = qk_scale or dh ** -0.5
    q = p["cls_token"].expand(B, -1, -1).reshape(B, num_queries, num_heads, dh).permute(0, 2, 1, 3) * scale
    x = batchnorm1d_eval(x, p["bn1.running_mean"], p["bn1.running_var"])
    kv = x @ p["wkv.weight"].T
    if "wkv.bias" in p:
        kv = kv + p["wkv.bias"]
    kv = kv.reshape(B, N, 2, num_heads, dh).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    att = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    x_cls = (att @ v).reshape(B, num_queries, C)
    x_cls = batchnorm1d_eval(x_cls, p["bn2.running_mean"], p["bn2.running_var"]).mean(dim=1)
    return x_cls @ p["linear.weight"].T + p["linear.bias"]


def forward_backward(cfg: MAEConfig, params: Dict[str, torch.Tensor], x: torch.Tensor, noise: torch.Tensor,
                     want_inter: bool = False, emulate_bf16: bool = False):
    """loss + autograd gradients of every trainable parameter (engine_pretrain_mae.py:58-62, AMP off)."""
    frozen = {n for n, _, rg in param_shapes(cfg) if not rg}
    p = {k: (v.clone().requires_grad_(k not in frozen)) for k, v in params.items()}
    loss, pred, mask, inter = forward(cfg, p, x, noise, want_inter, emulate_bf16=emulate_bf16)
    loss.backward()
    grads = {k: v.grad for k, v in p.items() if v.grad is not None}
    return loss.detach(), pred.detach(), mask, grads, ({k: v.detach() for k, v in inter.items()} if inter else None)


# ----------------------------------------------------------------------------
# optimizer step: per-parameter clip + AdamW + cosine-warmup LR
# ----------------------------------------------------------------------------
def clip_gradients_(grads: Dict[str, torch.Tensor], clip: float) -> Dict[str, float]:
    """src/utils/misc.py:374-383: PER-TENSOR L2 clip, coef = clip/(norm+1e-6), applied iff coef < 1."""
    norms = {}
    for k, g in grads.items():
        n = g.norm(2)
        norms[k] = float(n)
        coef = clip / (n + 1e-6)
        if coef < 1:
            g.mul_(coef)
    return norms


def cosine_warmup_lambda(step: int, warmup: int, total: int, lr_init: float, lr_end: float) -> float:
    """src/utils/lr_sched.py:46-53 (num_cycles = 0.5)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    lr_range = lr_init - lr_end
    progress = float(step - warmup) / float(max(1, total - warmup))
    lr_new = lr_end + lr_range * 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress))
    return max(0.0, lr_new / lr_init)


def adamw_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
                beta1: float, beta2: float, eps: float, wd: float) -> None:
    """torch.optim.AdamW single-tensor math (optimizers.py:354-360 -> torch defaults eps=1e-8,
    decoupled weight decay on EVERY parameter, one param group).  `step` is 1-based."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


@dataclass
class TrainState:
    params: Dict[str, torch.Tensor]
    exp_avg: Dict[str, torch.Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, torch.Tensor] = field(default_factory=dict)
    step: int = 0  # optimizer steps taken == scheduler steps taken


def train_step(cfg: MAEConfig, st: TrainState, x: torch.Tensor, noise: torch.Tensor, *, base_lr: float,
               min_lr: float, warmup: int, total: int, weight_decay: float, beta1: float = 0.9,
               beta2: float = 0.95, grad_clip: float = 0.0, emulate_bf16: bool = False):
    """One iteration of train_one_epoch (engine_pretrain_mae.py:52-71), AMP disabled:
    zero_grad -> forward -> backward -> per-param clip -> AdamW(lr_t) -> scheduler.step()."""
    loss, pred, mask, grads, _ = forward_backward(cfg, st.params, x, noise, emulate_bf16=emulate_bf16)
    norms = clip_gradients_(grads, grad_clip) if grad_clip else {}
    lr = base_lr * cosine_warmup_lambda(st.step, warmup, total, base_lr, min_lr)  # LambdaLR: lr at step index
    st.step += 1
    with torch.no_grad():
        for k, g in grads.items():
            if k not in st.exp_avg:
                st.exp_avg[k] = torch.zeros_like(g)
                st.exp_avg_sq[k] = torch.zeros_like(g)
            adamw_step_(st.params[k], g, st.exp_avg[k], st.exp_avg_sq[k], st.step, lr, beta1, beta2, 1e-8, weight_decay)
    return float(loss), lr, grads, norms


# ----------------------------------------------------------------------------
# algorithmic FLOPs
# ----------------------------------------------------------------------------
def algorithmic_flops_per_volume(cfg: MAEConfig) -> Dict[str, float]:
    D, M, Dd, Md = cfg.encoder_embed_dim, cfg.encoder_mlp_dim, cfg.decoder_embed_dim, cfg.decoder_mlp_dim
    L, K, pd = cfg.num_patches, cfg.len_keep, cfg.patch_dim
    Ne, Nd = K + 1, L + 1
    blk = lambda N, d, m: N * (8 * d * d + 4 * d * m) + 4 * N * N * d
    out = dict(pe=2.0 * K * pd * D, enc=float(cfg.encoder_depth * blk(Ne, D, M)), dec_embed=2.0 * Ne * D * Dd,
               dec=float(cfg.decoder_depth * blk(Nd, Dd, Md)), pred=2.0 * L * Dd * pd)
    out["fwd"] = sum(out.values())
    out["train"] = 3.0 * out["fwd"]
    return out
