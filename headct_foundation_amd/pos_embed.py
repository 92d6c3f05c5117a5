"""Resume a checkpoint at another input resolution: position-table interpolation.

Host mirror of `interpolate_pos_embed(model, checkpoint_model, spatial_dims=3)` (src/utils/pos_embed.py:102-153), which the
reference calls on the loaded state dict right before `load_state_dict` (main_pretrain_mae.py:132).  Same contract: the
entry `patch_embedding.position_embeddings` of `checkpoint_model` is replaced IN PLACE by a table resized to the model's
patch grid (trilinear, align_corners=False), extra leading tokens are kept; nothing happens when the grids agree or the key
is absent.  The resize runs in the HIP library (`hct_pos_embed_interp3d`); there is no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib


def nth_root(N: int, k: int) -> int:
    """Greatest integer x with x**k <= N (pos_embed.py:87-95)."""
    x = int(N ** (1 / k))
    while (x + 1) ** k <= N:
        x += 1
    while x ** k > N:
        x -= 1
    return x


def interpolate_pos_embed(model: torch.nn.Module, checkpoint_model, spatial_dims: int = 3) -> None:
    key = 'patch_embedding.position_embeddings'
    if key not in checkpoint_model:
        return
    if spatial_dims != 3:
        raise NotImplementedError(f"Spatial Dimension Size {spatial_dims} Not Implemented!")  # the MAE path is 3-D only
    pos_embed_checkpoint = checkpoint_model[key]
    embedding_size = pos_embed_checkpoint.shape[-1]
    num_patches = model.patch_embedding.n_patches
    num_extra_tokens = model.patch_embedding.position_embeddings.shape[-2] - num_patches
    orig_size = nth_root(pos_embed_checkpoint.shape[-2] - num_extra_tokens, spatial_dims)
    new_size = nth_root(num_patches, spatial_dims)
    if orig_size == new_size:
        return
    print("Position interpolate from origin size %d to new size %d" % (orig_size, new_size))
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.HctError("interpolate_pos_embed runs on the GPU (libheadct_hip); no CPU fallback exists")
    dev = model.patch_embedding.position_embeddings.device
    if dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    src = pos_embed_checkpoint.detach().to(device=dev, dtype=torch.float32).contiguous()
    dst = torch.empty(1, num_extra_tokens + new_size ** 3, embedding_size, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.hct_pos_embed_interp3d(src.data_ptr(), orig_size, dst.data_ptr(), new_size, embedding_size, num_extra_tokens, st),
                   "hct_pos_embed_interp3d")
        torch.cuda.current_stream().synchronize()
    checkpoint_model[key] = dst.to(device=pos_embed_checkpoint.device, dtype=pos_embed_checkpoint.dtype)
