"""Repeated-call data checks of the MFMA kernels at multi-tile sizes (the single-shot parity tests are in test_kernels_gpu).

An intermittent hardware hazard (gfx950: VALU write of a 16-byte buffer store's data registers right behind the store, see
HCT_STORE_GUARD in csrc/gemm.hip) corrupted 0.01-1 % of a large GEMM output in most calls but let the small single-shot
cases pass most of the time.  These tests repeat each call, pre-fill the output with a sentinel, compare every element with
an fp32 torch reference and require bit-identical results between repeats (every kernel here is deterministic by design).
"""
import pytest
import torch
import torch.nn.functional as F

from headct_foundation_amd import _lib
from test_kernels_gpu import _attn_ref, _dt, _rand, _st, gemm

pytestmark = pytest.mark.gpu
REPS = 6


def _outliers(x, ref, tol):
    return int(((x.float() - ref).abs() > tol + tol * ref.abs()).sum())


@pytest.mark.parametrize("M,N,K", [(2048, 3072, 768), (4096, 768, 768), (1000, 2304, 768)])
def test_nt_epilogues_repeated(lib, cuda, M, N, K):
    A = _rand((M, K), cuda, torch.bfloat16, 1)
    B = _rand((N, K), cuda, torch.bfloat16, 2, 0.05)
    bias = _rand((N,), cuda, torch.float32, 3)
    res = _rand((M, N), cuda, torch.float32, 4)
    aux_in = _rand((M, N), cuda, torch.bfloat16, 7)
    plain = A.float() @ B.float().t()
    u = aux_in.float().requires_grad_(True)
    F.gelu(u).sum().backward()
    pre = plain + bias
    refs = {"res": plain + bias + res, "dgelu": plain * u.grad, "plain": plain, "gelu": F.gelu(pre)}
    first = {}
    for rep in range(REPS):
        for tag in ("res", "dgelu", "plain", "gelu"):
            dt = torch.float32 if tag == "res" else torch.bfloat16
            t = torch.full((M, N), 777.0, dtype=dt, device=cuda)  # sentinel in the block the allocator hands out next
            torch.cuda.synchronize()
            del t
            if tag == "res":
                o = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res)
            elif tag == "dgelu":
                o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, act=2, aux=aux_in)
            elif tag == "plain":
                o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16)
            else:
                ax = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
                o = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, bias=bias, act=1, aux=ax)
                assert _outliers(ax, pre, 0.03) == 0, (tag, rep)
            assert _outliers(o, refs[tag], 1e-3 if tag == "res" else 0.03) == 0, (tag, rep)
            if tag in first:
                assert torch.equal(o, first[tag]), (tag, rep)
            else:
                first[tag] = o.clone()
            del o


@pytest.mark.parametrize("R,M,N", [(14080, 2304, 768), (8192, 768, 3072)])
def test_tn_wgrad_repeated(lib, cuda, R, M, N):
    A = _rand((R, M), cuda, torch.bfloat16, 5)
    B = _rand((R, N), cuda, torch.bfloat16, 6)
    ref = A.float().t() @ B.float()
    first = None
    for rep in range(REPS):
        o = gemm(lib, A, B, 1, 0, M, N, R)
        assert _outliers(o, ref, 2e-3 * R ** 0.5) == 0, rep
        if first is None:
            first = o.clone()
        assert torch.equal(o, first), rep


@pytest.mark.parametrize("B,N,H,dh", [(32, 217, 16, 48), (32, 55, 12, 64)])
def test_attention_repeated(lib, cuda, B, N, H, dh):
    qkv = _rand((B, N, 3 * H * dh), cuda, torch.bfloat16, 11)
    d_o = _rand((B, N, H * dh), cuda, torch.bfloat16, 12)
    qr = qkv.float().requires_grad_(True)
    o_ref, _ = _attn_ref(qr, B, N, H, dh)
    (o_ref * d_o.float()).sum().backward()
    first = None
    for rep in range(REPS):
        o = torch.empty(B, N, H * dh, dtype=torch.bfloat16, device=cuda)
        lse = torch.empty(B, H, N, dtype=torch.float32, device=cuda)
        _lib.check(lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, _dt(qkv), o.data_ptr(), lse.data_ptr(), _st()), "fwd")
        dqkv = torch.full_like(qkv, float("nan"))
        _lib.check(lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, _dt(qkv),
                                         dqkv.data_ptr(), _st()), "bwd")
        assert _outliers(o, o_ref.detach(), 0.03) == 0 and _outliers(dqkv, qr.grad, 0.06) == 0, rep
        cur = torch.cat([o.flatten(), dqkv.flatten()])
        if first is None:
            first = cur.clone()
        assert torch.equal(cur, first), rep
