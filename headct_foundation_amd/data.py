"""Input side of the MAE engine.  The reference's MONAI loading pipeline (src/data/*.py: NIfTI -> RAS -> 1 mm -> HU
window -> resize -> fp16 persistent cache) needs MONAI, which is absent from the image; the engine is fed synthetic volumes
with the value range of windowed CT, U[0,1) (transforms.py:120-128), generated per rank with seed SEED + rank like the
reference seeds its ranks (main_pretrain_mae.py:213).  What IS built of the input path is its per-sample
device side: `DeviceAugment` = the train-time transforms of `mae3d_transforms` (cast of the cached fp16 volume, three axis
flips, intensity shift as one HIP kernel; the optional Gaussian smoothing as three 1-D passes) and `window_hu`."""
from __future__ import annotations

import torch


GAUSS_TAPS = 9  # taps per axis the kernel takes (kGaussTaps): sigma <= 1.06 at MONAI's truncation of 4 sigma


def gaussian_taps(sigma: torch.Tensor) -> torch.Tensor:
    """Centred 1-D kernels for sigmas [...] -> [..., GAUSS_TAPS] fp32, as MONAI's `gaussian_1d(sigma, truncated=4.0, approx="erf")`
    builds them (fp32: tail = int(max(4 sigma, 0.5) + 0.5); w(x) = 0.5 (erf(t (x + 0.5)) - erf(t (x - 0.5))), t = 0.70710678 / |sigma|,
    clamped at 0, NOT renormalised), zero beyond the tail.  Stated from knowledge of MONAI 1.2 / 1.3 (not installed here)."""
    sigma = sigma.to(torch.float32)
    tail = torch.clamp(sigma * 4.0, min=0.5).add(0.5).to(torch.int64)
    if int(tail.max()) > GAUSS_TAPS // 2:
        raise ValueError(f"sigma {float(sigma.max())} needs more than {GAUSS_TAPS} taps")
    x = torch.arange(-(GAUSS_TAPS // 2), GAUSS_TAPS // 2 + 1, dtype=torch.float32)
    t = (0.70710678 / sigma.abs()).unsqueeze(-1)
    w = (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)
    return torch.where(x.abs() <= tail.unsqueeze(-1).to(torch.float32), w, torch.zeros_like(w))


class DeviceAugment:
    """mae3d_transforms(mode='train') (src/data/transforms.py:193-238) on a device batch: CastToTyped(float32) ->
    RandFlipd(prob, axis 0/1/2) -> RandShiftIntensityd(offsets, prob) [-> RandGaussianSmoothd(sigma per axis ~ U(0.5, 1), prob 0.2),
    which the reference appends when `reshape` is False: `smooth_prob=0.2` here].  Input: [B,C,S,S,S] fp16 (the cache format,
    transforms.py:170-175), bf16 or fp32; output fp32.  Draws come from a torch generator on the host (MONAI's numpy RandomState
    stream is not reproduced); `last_draw` exposes (flip bits, shift[, smooth flags, sigmas]) for tests."""

    def __init__(self, flip_prob: float = 0.1, shift_offsets: float = 0.1, shift_prob: float = 0.5, seed: int = 0,
                 smooth_prob: float = 0.0, smooth_sigma=(0.5, 1.0)):
        self.flip_prob, self.shift_offsets, self.shift_prob = flip_prob, shift_offsets, shift_prob
        self.smooth_prob, self.smooth_sigma = smooth_prob, smooth_sigma
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        self.last_draw = None

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        from . import _lib
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.HctError("DeviceAugment runs on the GPU (libheadct_hip); no CPU fallback exists")
        B, C, S = x.shape[0], x.shape[1], x.shape[2]
        code = {torch.float16: _lib.HCT_F16, torch.bfloat16: _lib.HCT_BF16, torch.float32: _lib.HCT_F32}[x.dtype]
        u = torch.rand(B, 5, generator=self.gen)
        flip = ((u[:, 0] < self.flip_prob).to(torch.uint8) | ((u[:, 1] < self.flip_prob).to(torch.uint8) << 1)
                | ((u[:, 2] < self.flip_prob).to(torch.uint8) << 2))
        shift = torch.where(u[:, 3] < self.shift_prob, (u[:, 4] * 2 - 1) * self.shift_offsets, torch.zeros(B))
        self.last_draw = (flip.clone(), shift.clone())
        x = x.contiguous()
        flip_d, shift_d = flip.to(x.device), shift.to(device=x.device, dtype=torch.float32)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.hct_augment_volume(x.data_ptr(), code, out.data_ptr(), B, C, S, flip_d.data_ptr(), shift_d.data_ptr(), st),
                       "hct_augment_volume")
        if self.smooth_prob > 0:
            v = torch.rand(B, 4, generator=self.gen)
            fire = v[:, 0] < self.smooth_prob
            lo, hi = self.smooth_sigma
            sigma = lo + (hi - lo) * v[:, 1:4]  # one sigma per spatial axis, drawn whether or not the transform fires (as MONAI's randomize does)
            self.last_draw = self.last_draw + (fire.clone(), sigma.clone())
            if bool(fire.any()):
                out = gaussian_smooth(out, sigma, fire)
        return out


def gaussian_smooth(x: torch.Tensor, sigma: torch.Tensor, apply: torch.Tensor = None) -> torch.Tensor:
    """Separable Gaussian smoothing of fp32 volumes [B,C,S,S,S] with per-sample, per-axis sigmas [B,3] (zero padding at the borders):
    MONAI's GaussianSmooth as RandGaussianSmoothd applies it (src/data/transforms.py:230-238).  apply [B] bool: samples left as is."""
    from . import _lib
    lib = _lib.load()
    if not x.is_cuda:
        raise _lib.HctError("gaussian_smooth runs on the GPU (libheadct_hip); no CPU fallback exists")
    if x.dtype != torch.float32:
        raise _lib.HctError("gaussian_smooth takes fp32 volumes (the output of DeviceAugment)")
    x = x.contiguous()
    B, C, S = x.shape[0], x.shape[1], x.shape[2]
    taps = gaussian_taps(sigma.reshape(B, 3)).to(x.device).contiguous()
    flags = (torch.ones(B, dtype=torch.uint8) if apply is None else apply.to(torch.uint8)).to(x.device)
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.hct_gaussian_smooth3d(x.data_ptr(), out.data_ptr(), tmp.data_ptr(), B, C, S, taps.data_ptr(), flags.data_ptr(), st),
                   "hct_gaussian_smooth3d")
    return out


# (centre, width) of the reference's three-channel input (transforms.py:130) and its one-channel window 40 +- 150 (:121-122)
HU_WINDOWS = {1: [(-110.0, 190.0)], 3: [(l - w // 2, l + w // 2) for l, w in ((40, 80), (80, 200), (600, 2800))]}


def window_hu(hu: torch.Tensor, in_channels: int = 1, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """HU volumes [B, 1, ...] (fp32 or fp16) -> windowed [B, in_channels, ...] in [0, 1]: the windowing step of
    `loading_transforms` (src/data/transforms.py:108-133) on the device, cast to the cache's fp16 by default."""
    from . import _lib
    lib = _lib.load()
    if not hu.is_cuda:
        raise _lib.HctError("window_hu runs on the GPU (libheadct_hip); no CPU fallback exists")
    if in_channels not in HU_WINDOWS:
        raise NotImplementedError(f"Channel size {in_channels} is not implemented.")
    if hu.dtype not in (torch.float32, torch.float16):
        hu = hu.float()
    hu = hu.contiguous()
    B, vox = hu.shape[0], hu[0].numel()
    lo = torch.tensor([w[0] for w in HU_WINDOWS[in_channels]], dtype=torch.float32, device=hu.device)
    hi = torch.tensor([w[1] for w in HU_WINDOWS[in_channels]], dtype=torch.float32, device=hu.device)
    out = torch.empty((B, in_channels) + tuple(hu.shape[2:]), dtype=out_dtype, device=hu.device)
    code = {torch.float16: _lib.HCT_F16, torch.float32: _lib.HCT_F32}
    with torch.cuda.device(hu.device):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.hct_hu_window(hu.data_ptr(), code[hu.dtype], out.data_ptr(), code[out_dtype], B, vox, in_channels,
                                     lo.data_ptr(), hi.data_ptr(), st), "hct_hu_window")
    return out


class SyntheticVolumes:
    """A fixed pool of `n_batches` pre-generated [B,C,S,S,S] batches on `device`, cycled (len == n_batches)."""

    def __init__(self, n_batches, batch_size, in_chans, size, device, seed=0):
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.batches = [torch.rand(batch_size, in_chans, size, size, size, device=device, generator=gen) for _ in range(n_batches)]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def get_pretrain_dataloaders(config, device, rank=0, world_size=1):
    """train / val / test loaders.  Only DATA.SYNTHETIC is implemented (see module docstring)."""
    if not config.DATA.SYNTHETIC:
        raise NotImplementedError(
            "the MONAI/NIfTI data path is out of scope for this build (SURVEY 8f #2); set DATA.SYNTHETIC True")
    bs, n = config.DATA.BATCH_SIZE, config.DATA.SYNTHETIC_SAMPLES
    per_rank = max(1, n // max(1, world_size))
    nb = max(1, per_rank // bs)
    mk = lambda k, salt: SyntheticVolumes(k, bs, config.MAE.IN_CHANS, config.MAE.INPUT_SIZE, device, config.SEED + rank + salt)
    return mk(nb, 0), mk(max(1, nb // 4), 1000), mk(max(1, nb // 4), 2000)
