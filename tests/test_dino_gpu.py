"""GPU: the DINO pieces on the HIP path (BASELINE config #5) against the oracle and the fixture generated from the reference's
own DINOLoss / _update_momentum_encoder (tests/golden/dino.json)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dino_oracle as D
from oracle import mae_oracle as O
from tests.util import GOLDEN, rel_err, sample_of

pytestmark = pytest.mark.gpu


def _hu(shape, seed, lo, hi):
    return torch.from_numpy(O.hash_uniform(int(np.prod(shape)), seed)).float().reshape(shape) * (hi - lo) + lo


def test_dino_loss_vs_reference_fixture(lib, cuda):
    from headct_foundation_amd.dino import DINOLoss
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))["loss"]
    V, B, K = fx["V"], fx["B"], fx["K"]
    student = _hu((V * B, K), fx["student_seed"], -3.0, 3.0).to(cuda).requires_grad_(True)
    teacher = _hu((2 * B, K), fx["teacher_seed"], -3.0, 3.0).to(cuda)
    crit = DINOLoss(K, V, 0.04, 0.07, 3, 10, student_temp=fx["student_temp"], center_momentum=fx["center_momentum"]).to(cuda)
    crit.center.copy_(_hu((1, K), fx["center_seed"], -0.5, 0.5))
    assert list(crit.teacher_temp_schedule) == fx["teacher_temp_schedule"]
    loss = crit(student, teacher, 1)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - fx["loss"]) < 1e-5 * abs(fx["loss"])
    got, want, l2, l2w = sample_of(student.grad, fx["dstudent"])
    assert abs(l2 - l2w) < 1e-4 * l2w and torch.allclose(got, want, rtol=2e-4, atol=1e-8)
    got, want, _, _ = sample_of(crit.center, fx["center_after"])
    assert torch.allclose(got, want, rtol=1e-5, atol=2e-6)  # column sums over the batch are added in another order than torch.sum


@pytest.mark.parametrize("V,B,K,dtype", [(10, 8, 65536, torch.float32), (10, 4, 65536, torch.bfloat16), (2, 5, 1000, torch.float32)])
def test_dino_loss_full_width_vs_oracle(lib, cuda, V, B, K, dtype):
    """The reference's prototype count (65 536), 2 global + 8 local crops; fp32 logits at 1e-5, bf16 logits against the oracle on
    the same rounded values."""
    from headct_foundation_amd.dino import DINOLoss
    student = (_hu((V * B, K), 601, -4.0, 4.0)).to(dtype)
    teacher = (_hu((2 * B, K), 602, -4.0, 4.0)).to(dtype)
    center = _hu((1, K), 603, -1.0, 1.0)
    so = student.float().detach().clone().requires_grad_(True)
    o_loss = D.dino_loss(so, teacher.float(), center, V, 0.1, 0.04)
    o_loss.backward()
    crit = DINOLoss(K, V, 0.04, 0.04, 0, 5).to(cuda)
    crit.center.copy_(center)
    s = student.detach().clone().to(cuda).requires_grad_(True)
    loss = crit(s, teacher.to(cuda), 0)
    (loss * 3.0).backward()  # the incoming gradient scales dstudent
    torch.cuda.synchronize()
    assert abs(float(loss) - float(o_loss)) < 2e-5 * abs(float(o_loss))
    assert rel_err(s.grad.float() / 3.0, so.grad) < (1e-4 if dtype == torch.float32 else 6e-3)
    assert torch.allclose(crit.center.cpu(), D.update_center(center, teacher.float(), 0.9), rtol=1e-5, atol=2e-6)


def test_momentum_encoder_update_is_bit_exact(lib, cuda):
    from headct_foundation_amd.dino import ema_update_
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))["ema"]
    for shape, qs, ks, want in zip(fx["shapes"], fx["q_seeds"], fx["k_seeds"], fx["k_after"]):
        n = int(np.prod(shape))
        pad = (-n) % 4
        q = torch.cat([_hu(tuple(shape), qs, -1, 1).flatten(), torch.zeros(pad)]).to(cuda)
        k = torch.cat([_hu(tuple(shape), ks, -1, 1).flatten(), torch.zeros(pad)]).to(cuda)
        ema_update_(k, q, fx["m"])
        assert torch.equal(k.cpu()[:n], torch.tensor(want))
    big_q, big_k = _hu((1 << 20,), 611, -1, 1), _hu((1 << 20,), 612, -1, 1)
    ref = [big_k.clone()]
    D.update_momentum_encoder([big_q], ref, 0.9995)
    kd = big_k.to(cuda)
    ema_update_(kd, big_q.to(cuda), 0.9995)
    assert torch.equal(kd.cpu(), ref[0])
