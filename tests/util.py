"""Shared helpers for the parity tests (HIP path vs. CPU oracle)."""
import json
import os

import numpy as np
import torch

from oracle import mae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def sample_of(t: torch.Tensor, entry):
    f = t.detach().double().cpu().flatten()
    idx = torch.tensor(entry["idx"], dtype=torch.long)
    return f[idx], torch.tensor(entry["val"], dtype=torch.float64), float(f.norm()), entry["l2"]


def build_hip_model(cfg: O.MAEConfig, params, device, compute_dtype="fp32", full_pred=True):
    """full_pred=True: training forwards also predict the kept patches, so that pred and the last decoder block's output can be
    compared row by row with the oracle; False = the module's default (compact decoder tail: masked patches only)."""
    from headct_foundation_amd import MaskedAutoencoderViT
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs(), compute_dtype=compute_dtype)
    missing = m.load_state_dict(params, strict=True)
    m.full_pred = full_pred
    return m.to(device)


def grads_by_name(model):
    return {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
