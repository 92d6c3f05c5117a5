"""Input side of the MAE engine.  The reference's MONAI pipeline (src/data/*.py: NIfTI -> RAS -> 1 mm -> HU window ->
resize -> fp16 persistent cache) is outside this round's scope and MONAI is absent from the image;
the engine is fed synthetic volumes with the value range of windowed CT, U[0,1) (transforms.py:120-128), generated
per rank with seed SEED + rank like the reference seeds its ranks (main_pretrain_mae.py:213)."""
from __future__ import annotations

import torch


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
