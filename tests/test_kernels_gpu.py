"""Per-kernel parity through the C ABI (include/headct_hip.h) against plain PyTorch fp32 references of the same op.
Shapes are the MAE tile shapes (N = 55 / 217 tokens, head dim 64 / 48, D = 768 ...) plus ragged / edge cases."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from headct_foundation_amd import _lib
from headct_foundation_amd._lib import HCT_BF16, HCT_F32, GemmArgs
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _st():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    return HCT_BF16 if t.dtype == torch.bfloat16 else HCT_F32


def gemm(lib, A, B, transA, transB, M, N, K, out_dtype=torch.float32, bias=None, residual=None, act=0, aux=None,
         force_generic=False, c2_dtype=None):
    dev = A.device
    Cm = torch.empty(M, N, dtype=out_dtype, device=dev)
    C2 = torch.empty(M, N, dtype=c2_dtype, device=dev) if c2_dtype is not None else None
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), _dt(A), A.stride(0), int(transA)
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), _dt(B), B.stride(0), int(transB)
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), _dt(Cm), N
    a.bias = bias.data_ptr() if bias is not None else None
    a.residual = residual.data_ptr() if residual is not None else None
    a.ldr = N
    a.act = act
    if aux is not None:
        a.aux, a.aux_dtype, a.ldaux = aux.data_ptr(), _dt(aux), N
    if C2 is not None:
        a.C2, a.c2_dtype, a.ldc2 = C2.data_ptr(), _dt(C2), N
    a.alpha = 1.0
    a.force_generic = int(force_generic)
    ws_bytes = lib.hct_gemm_workspace_bytes(C.byref(a))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), _st()), "hct_gemm")
    return (Cm, C2) if C2 is not None else Cm


def _rand(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(device=dev, dtype=dtype)


@pytest.mark.parametrize("M,N,K", [(55 * 3, 768, 768), (217 * 2, 2304, 768), (130, 3072, 768), (128, 768, 3072),
                                   (100, 192, 192), (1, 64, 64), (257, 4096, 768), (300, 48, 64),
                                   (14080, 768, 768), (1000, 512, 256), (9999, 768, 3072)])  # 192-row tiles (MT = 3): the encoder's shape; ragged last row tiles
def test_gemm_nt_bf16_mfma(lib, cuda, M, N, K):
    A = _rand((M, K), cuda, torch.bfloat16, 1)
    B = _rand((N, K), cuda, torch.bfloat16, 2, 0.05)
    bias = _rand((N,), cuda, torch.float32, 3)
    res = _rand((M, N), cuda, torch.float32, 4)
    ref = A.float() @ B.float().t() + bias + res
    out = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res)
    gen = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res, force_generic=True)
    assert rel_err(out, ref) < 1e-5 and rel_err(gen, ref) < 1e-5  # exact bf16 products, fp32 accumulate
    # the default dispatch takes 192-row tiles where 256-row tiles fill less than one round of CUs; -15 switches them off: bit-equal
    lib.hct_debug_set_gemm_variant(-15)
    try:
        o256 = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res)
    finally:
        lib.hct_debug_set_gemm_variant(-14)
    assert torch.equal(o256, out)
    for variant in (128, 256, 4):  # every tuned NT kernel (2-stage 128^2, persistent 256^2, 2-WG/CU 256x128)
        lib.hct_debug_set_gemm_variant(variant)
        try:
            o = gemm(lib, A, B, 0, 1, M, N, K, bias=bias, residual=res)
            ob = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16)
        finally:
            lib.hct_debug_set_gemm_variant(0)
        assert rel_err(o, ref) < 1e-5, variant
        assert rel_err(ob, A.float() @ B.float().t()) < 4e-3, variant
    # GELU epilogue with pre-activation side output, bf16 stores
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
    o2 = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, bias=bias, act=1, aux=aux)
    pre = A.float() @ B.float().t() + bias
    assert rel_err(aux, pre) < 4e-3 and rel_err(o2, F.gelu(pre)) < 4e-3
    # dGELU epilogue
    dy = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, act=2, aux=aux)
    u = aux.float().requires_grad_(True)
    F.gelu(u).sum().backward()
    assert rel_err(dy, (A.float() @ B.float().t()) * u.grad) < 4e-3


def _nt_call(lib, A, B, M, N, K, ws, out_dtype, bias=None, residual=None, act=0, aux=None, colsum=None, armed=0):
    Cm = torch.empty(M, N, dtype=out_dtype, device=A.device)
    a = GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), _dt(A), K, 0
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), _dt(B), K, 1
    a.C, a.c_dtype, a.ldc = Cm.data_ptr(), _dt(Cm), N
    a.bias = bias.data_ptr() if bias is not None else None
    a.residual = residual.data_ptr() if residual is not None else None
    a.ldr = N
    a.act = act
    if aux is not None:
        a.aux, a.aux_dtype, a.ldaux = aux.data_ptr(), _dt(aux), N
    a.colsum_out = colsum.data_ptr() if colsum is not None else None
    a.alpha = 1.0
    a.workspace_armed = armed
    assert ws is None or ws.numel() >= lib.hct_gemm_workspace_bytes(C.byref(a))
    _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _st()), "hct_gemm")
    return Cm


@pytest.mark.parametrize("M,N,K,reserve", [(2048, 2048, 1024, 0), (3000, 1008, 1536, 0), (55552, 768, 3072, 0), (55552, 3072, 768, 0),
                                           (14080, 768, 2304, 0), (2000, 496, 512, 0), (3000, 1008, 1536, 16),
                                           (14080, 768, 2304, 64)])
def test_gemm_nt_stream_k_remainder(lib, cuda, M, N, K, reserve):
    """Persistent NT kernel with the remainder round shared out by K range: every epilogue instance against the whole-tile
    schedule (hook -1000-k: stream-K only for K >= k) and against fp32 torch, bit-identical repeats, flags consumed, no timeout.
    The shapes give 4 (first, last), 6 and ~26 / ~45 stage pairs per K range, ragged M and N, and the step's own worst cases
    (651 tiles = 2.54 rounds; 2604 tiles = 10.17 rounds)."""
    A = _rand((M, K), cuda, torch.bfloat16, 11)
    B = _rand((N, K), cuda, torch.bfloat16, 12, 0.05)
    bias = _rand((N,), cuda, torch.float32, 13)
    res = _rand((M, N), cuda, torch.float32, 14)
    nbytes = (M + 255) // 256 * 4 * N * 4 + (1 << 20) + 64 * 1024 * 1024 + 4096
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=cuda)
    off = lib.hct_gemm_nt_flags_offset(ws.numel())
    assert off != 2 ** 64 - 1 and off % 256 == 0
    flags = ws[off:off + 4096].view(torch.int32)
    ref = A.float() @ B.float().t()

    def run(armed):
        out = {}
        out["plain"] = _nt_call(lib, A, B, M, N, K, ws, torch.bfloat16, armed=armed)
        out["res"] = _nt_call(lib, A, B, M, N, K, ws, torch.float32, bias=bias, residual=res, armed=armed)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
        out["gelu"] = _nt_call(lib, A, B, M, N, K, ws, torch.bfloat16, bias=bias, act=1, aux=aux, armed=armed)
        out["aux"] = aux
        out["dgelu"] = _nt_call(lib, A, B, M, N, K, ws, torch.bfloat16, act=2, aux=aux, armed=armed)
        cs = torch.empty(N, dtype=torch.float32, device=cuda)
        out["dgelu_cs"] = _nt_call(lib, A, B, M, N, K, ws, torch.bfloat16, act=2, aux=aux, colsum=cs, armed=armed)
        out["cs"] = cs
        out["generic"] = _nt_call(lib, A, B, M, N, K, ws, torch.float32, bias=bias, armed=armed)  # fp32 out without residual: runtime-flag epilogue
        torch.cuda.synchronize()
        return out

    lib.hct_debug_set_gemm_variant(-1000 - 512)
    lib.hct_debug_set_gemm_variant(-100 - 1)  # take every remainder round that saves at least one pair
    lib.hct_set_cu_reserve(reserve)  # the data-parallel wrapper leaves CUs to RCCL during the backward: grids of 240 / 192 workgroups
    try:
        flags.zero_()
        sk = run(0)
        used = int((flags[:256] != 0).sum())
        assert used > 0, "stream-K did not engage"
        assert int(flags[512]) == 0, "a partial never arrived"
        sk2 = run(1)  # armed: no reset between launches, flags carry the previous launches' sequence numbers
        assert int(flags[512]) == 0
    finally:
        lib.hct_debug_set_gemm_variant(-1000 - 512)
        lib.hct_debug_set_gemm_variant(-100 - 20)
        lib.hct_set_cu_reserve(0)
    lib.hct_debug_set_gemm_variant(-1000 - (1 << 24))
    try:
        whole = run(0)
    finally:
        lib.hct_debug_set_gemm_variant(-1000 - 512)
    for k in sk:
        assert torch.equal(sk[k], sk2[k]), k  # fixed summation order
    assert rel_err(sk["res"], ref + bias + res) < 1e-5 and rel_err(sk["generic"], ref + bias) < 1e-5
    assert rel_err(sk["plain"], ref) < 4e-3
    for k in ("res", "generic"):  # fp32 outputs: the K ranges only reorder the fp32 sums
        assert rel_err(sk[k], whole[k]) < 1e-6, k
    for k in ("plain", "gelu", "aux", "dgelu", "dgelu_cs"):  # bf16 outputs: a different summation order may flip a rounding
        assert rel_err(sk[k], whole[k]) < 2e-3, k
    assert rel_err(sk["cs"], whole["cs"]) < 1e-4


def test_stream_k_timeout_poisons_the_output(lib, cuda):
    """A stream-K partial that never arrives (a grid that is not wholly resident, e.g. beside a communication kernel that holds
    more CUs than were reserved for it) must not produce silently wrong numbers: the owner of the shared tile sets the error word
    AND poisons its tile with NaN, which the engine's finite-loss check catches.  The debug hook -8 makes every follower publish a
    wrong sequence number; -9 restores it, after which the same call is finite and correct again.  NT kernel and grouped wgrad."""
    M, N, K = 2048, 2048, 1024
    A = _rand((M, K), cuda, torch.bfloat16, 11)
    B = _rand((N, K), cuda, torch.bfloat16, 12, 0.05)
    ws = torch.zeros((1 << 20) + 64 * 1024 * 1024 + 4096, dtype=torch.uint8, device=cuda)
    flags = ws[lib.hct_gemm_nt_flags_offset(ws.numel()):][:4096].view(torch.int32)
    ops = [(_rand((2000, 768), cuda, torch.bfloat16, 21), _rand((2000, 512), cuda, torch.bfloat16, 22, 0.1), 1.0)]
    lib.hct_debug_set_gemm_variant(-100 - 1)
    try:
        lib.hct_debug_set_gemm_variant(-8)
        bad = _nt_call(lib, A, B, M, N, K, ws, torch.float32)
        torch.cuda.synchronize()
        # 64 tiles on 256 CUs: every tile is shared and every owner timed out; an owner poisons one accumulator tile per wave (4 of the
        # 128 outputs a lane holds), which is what reaches the loss through every later layer
        assert int(flags[512]) == 0xDEAD and torch.isnan(bad).any()
        assert all(bool(torch.isnan(bad[r0:r0 + 256, c0:c0 + 256]).any()) for r0 in range(0, M, 256) for c0 in range(0, N, 256))
        with pytest.raises(AssertionError):
            _tn_group(lib, ops)  # (its own check of the error word)
        lib.hct_debug_set_gemm_variant(-9)
        flags.zero_()
        good = _nt_call(lib, A, B, M, N, K, ws, torch.float32)
        torch.cuda.synchronize()
        assert int(flags[512]) == 0 and rel_err(good, A.float() @ B.float().t()) < 1e-5
        (g,) = _tn_group(lib, ops)
        assert rel_err(g, ops[0][0].float().t() @ ops[0][1].float()) < 2e-5
    finally:
        lib.hct_debug_set_gemm_variant(-9)
        lib.hct_debug_set_gemm_variant(-100 - 20)


@pytest.mark.parametrize("R,M,N", [(217 * 4, 768, 768), (55 * 5, 2304, 768), (1000, 768, 3072), (64, 192, 192),
                                   (5000, 4096, 768), (33, 48, 64), (130, 576, 192)])
def test_gemm_tn_bf16_mfma(lib, cuda, R, M, N):
    """wgrad: C[M,N] = A[R,M]^T . B[R,N]; ragged reduction lengths exercise the zero-fill path and split-K."""
    A = _rand((R, M), cuda, torch.bfloat16, 5)
    B = _rand((R, N), cuda, torch.bfloat16, 6)
    ref = A.float().t() @ B.float()
    out = gemm(lib, A, B, 1, 0, M, N, R)
    gen = gemm(lib, A, B, 1, 0, M, N, R, force_generic=True)
    assert rel_err(out, ref) < 2e-5 and rel_err(gen, ref) < 2e-5
    out2 = gemm(lib, A, B, 1, 0, M, N, R)
    assert torch.equal(out, out2)  # deterministic split-K fold
    # the in-launch fold (testing hook -7; the default is the separate fold kernel) sums the split partials in split order
    # too: bit-identical results, and bit-identical between repeats whatever the arrival order of the splits
    lib.hct_debug_set_gemm_variant(-7)
    try:
        fused = gemm(lib, A, B, 1, 0, M, N, R)
        fused2 = gemm(lib, A, B, 1, 0, M, N, R)
    finally:
        lib.hct_debug_set_gemm_variant(-6)
    assert torch.equal(out, fused) and torch.equal(fused, fused2)


def _tn_group(lib, ops, reps=1):
    """ops: list of (A [R,M] bf16, B [R,N] bf16, alpha).  Returns the fp32 products of the grouped launch (last repetition)."""
    n = len(ops)
    jobs = (GemmArgs * n)()
    outs = []
    for i, (A, B, alpha) in enumerate(ops):
        R, M = A.shape
        N = B.shape[1]
        Cm = torch.full((M, N), float("nan"), device=A.device)
        outs.append(Cm)
        a = jobs[i]
        a.M, a.N, a.K = M, N, R
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, A.stride(0), 1
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, B.stride(0), 0
        a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_F32, N
        a.alpha = alpha
    nbytes = lib.hct_gemm_tn_group_workspace_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ops[0][0].device)
    _lib.check(lib.hct_gemm_tn_group_prepare(C.cast(jobs, C.c_void_p), n, ws.data_ptr(), nbytes, _st()), "tn_group_prepare")
    for _ in range(reps):
        _lib.check(lib.hct_gemm_tn_group_run(C.cast(jobs, C.c_void_p), n, ws.data_ptr(), nbytes, _st()), "tn_group_run")
    torch.cuda.synchronize()
    flags = ws[lib.hct_gemm_tn_group_workspace_bytes(n) - (1024 * 262144 + 16384):][:4096].view(torch.int32)  # head of the stream-K region
    if int(flags[512]) != 0:  # a partial never arrived: the owners' tiles must then be NaN, not silently wrong
        assert any(torch.isnan(o).any() for o in outs)
    assert int(flags[512]) == 0, "a stream-K partial of the grouped wgrad never arrived"
    return outs


@pytest.mark.parametrize("case", ["stream_k_only", "rounds_plus_remainder", "mixed_small", "one_tile", "reserve16"])
def test_gemm_tn_group(lib, cuda, case):
    """Grouped weight gradients (one persistent launch, whole 256x256 tiles over the full reduction + a stream-K remainder round):
    every product against a float reference (bf16 products are exact in fp32: 2e-5), repeat launches bit-identical, and the same
    bits on a 240-workgroup grid split differently (reserve16: different partial sums, so 1e-5 rather than bit equality)."""
    shapes = {
        # (R, M, N) per job: 108 tiles on 256 CUs -> no whole round, every tile split over the reduction
        "stream_k_only": [(2000, 768, 3072), (2000, 3072, 768), (2016, 768, 768), (2000, 2304, 768)],
        # 8 x 72 = 576 tiles -> two whole rounds + 64 remainder tiles; two reduction lengths
        "rounds_plus_remainder": [(300, 1536, 3072)] * 5 + [(1000, 3072, 1536)] * 3,
        "mixed_small": [(130, 48, 64), (33, 16, 16), (4000, 272, 528), (257, 768, 16)],
        "one_tile": [(5000, 256, 256)],
        "reserve16": [(2000, 768, 3072), (2000, 3072, 768), (2016, 768, 768), (2000, 2304, 768)],
    }[case]
    ops = []
    for i, (R, M, N) in enumerate(shapes):
        ops.append((_rand((R, M), cuda, torch.bfloat16, 100 + i), _rand((R, N), cuda, torch.bfloat16, 200 + i, 0.1), 1.0 if i % 2 == 0 else 0.5))
    if case == "reserve16":
        lib.hct_set_cu_reserve(16)
    try:
        got = _tn_group(lib, ops)
        again = _tn_group(lib, ops, reps=3)
    finally:
        lib.hct_set_cu_reserve(0)
    for (A, B, alpha), g, g2 in zip(ops, got, again):
        want = alpha * (A.float().t() @ B.float())
        assert torch.isfinite(g).all()
        assert rel_err(g, want) < 2e-5, (case, tuple(A.shape), tuple(B.shape))
        assert torch.equal(g, g2), "grouped wgrad is not bit-reproducible"


def test_gemm_tn_group_rejects_other_products(lib, cuda):
    A, B = _rand((64, 32), cuda, torch.bfloat16, 1), _rand((64, 48), cuda, torch.bfloat16, 2)
    jobs = (GemmArgs * 1)()
    a = jobs[0]
    a.M, a.N, a.K = 32, 48, 64
    a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, 32, 0  # not transposed: not a wgrad
    a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, 48, 0
    Cm = torch.empty(32, 48, device=cuda)
    a.C, a.c_dtype, a.ldc, a.alpha = Cm.data_ptr(), HCT_F32, 48, 1.0
    n = lib.hct_gemm_tn_group_workspace_bytes(1)
    ws = torch.empty(n, dtype=torch.uint8, device=cuda)
    assert lib.hct_gemm_tn_group_prepare(C.cast(jobs, C.c_void_p), 1, ws.data_ptr(), n, _st()) != 0
    assert lib.hct_gemm_tn_group_prepare(C.cast(jobs, C.c_void_p), 1, ws.data_ptr(), 16, _st()) != 0  # workspace too small


def test_gemm_generic_fp32_all_layouts(lib, cuda):
    M, N, K = 77, 53, 45
    A, B = _rand((M, K), cuda, torch.float32, 7), _rand((K, N), cuda, torch.float32, 8)
    ref = A @ B
    assert rel_err(gemm(lib, A, B.contiguous(), 0, 0, M, N, K), ref) < 1e-5
    assert rel_err(gemm(lib, A, B.t().contiguous(), 0, 1, M, N, K), ref) < 1e-5
    assert rel_err(gemm(lib, A.t().contiguous(), B.contiguous(), 1, 0, M, N, K), ref) < 1e-5
    assert rel_err(gemm(lib, A.t().contiguous(), B.t().contiguous(), 1, 1, M, N, K), ref) < 1e-5
    assert gemm(lib, A[:0], B.contiguous(), 0, 0, 0, N, K).numel() == 0  # empty


def _attn_ref(qkv, B, N, H, dh):
    q, k, v = qkv.float().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * dh ** -0.5
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, N, H * dh)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,H,dh,dtype", [(2, 55, 12, 64, torch.bfloat16), (2, 217, 16, 48, torch.bfloat16),
                                            (3, 17, 3, 64, torch.bfloat16), (1, 1, 2, 48, torch.bfloat16),
                                            (2, 65, 3, 64, torch.float32), (2, 217, 4, 48, torch.float32),
                                            (2, 33, 3, 16, torch.float32), (1, 129, 2, 64, torch.bfloat16),
                                            (1, 513, 2, 48, torch.bfloat16), (2, 200, 16, 48, torch.bfloat16),
                                            (2, 100, 4, 64, torch.bfloat16), (2, 40, 2, 48, torch.bfloat16),
                                            (3, 256, 2, 48, torch.bfloat16), (2, 192, 2, 64, torch.bfloat16),
                                            (20, 217, 16, 48, torch.bfloat16), (37, 193, 8, 48, torch.bfloat16),
                                            (2, 70, 3, 64, torch.bfloat16), (2, 165, 2, 48, torch.bfloat16), (2, 226, 2, 48, torch.bfloat16),
                                            (2, 517, 3, 64, torch.bfloat16), (1, 300, 2, 48, torch.bfloat16), (5, 513, 16, 48, torch.bfloat16), (1, 576, 2, 48, torch.bfloat16), (3, 450, 3, 48, torch.bfloat16), (20, 129, 16, 64, torch.bfloat16), (3, 160, 2, 64, torch.bfloat16),
                                            (2, 529, 3, 64, torch.bfloat16),
                                            (3, 224, 2, 48, torch.bfloat16), (3, 208, 2, 48, torch.bfloat16)])  # bwd4: no padded key at all / a key tile wholly past N  # lowest lengths of the forward's 10 / 14 / 18-tile instances
def test_attention_fwd_bwd(lib, cuda, B, N, H, dh, dtype):
    qkv = _rand((B, N, 3 * H * dh), cuda, dtype, 11)
    d_o = _rand((B, N, H * dh), cuda, dtype, 12)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, N, H, dh)
    (o_ref * d_o.float()).sum().backward()
    dt = _dt(qkv)
    tol = 1.5e-2 if dtype == torch.bfloat16 else 2e-5
    # 0 default (five-product key-owner backward where it applies), 1 fp32-math kernels, 2 online-softmax forward,
    # 14 single-phase backward, 42 two-phase seven-product backward
    # 100003: bwd3 key-owner backward on every shape it covers (the default uses it for head dim 64 only); 100000: two-phase
    # everywhere; 100014: the opt-in persistent forward (fwd4) and the 8-wave form of bwd4; 100950: the long-sequence five-product
    # kernel (bwd5) from 225 tokens on and for head dim 64 too (default: 449 .. 576 tokens at head dim 48); the default (101206: + bit 10, bwd4 for 129 .. 160 tokens at head dim 64) takes the 16-wave
    # persistent bwd4 for head dim 48
    # with 193 .. 224 tokens -- the last two cases
    # have more (batch, head) items than CUs, so its workgroups walk several items through both LDS buffers
    modes = (0, 1, 2, 14, 42, 100003, 100000, 100014, 100950)
    if N > 400 and B * H * N > 20000:  # the larger long-sequence cases: the default dispatch, the two-phase kernel and bwd5 forced onto every length it covers
        modes = (0, 100000, 100950)
    for force_simple in (modes if dtype == torch.bfloat16 else (0,)):
        lib.hct_debug_force_simple_attention(force_simple)
        try:
            o = torch.empty(B, N, H * dh, dtype=dtype, device=cuda)
            lse = torch.empty(B, H, N, dtype=torch.float32, device=cuda)
            _lib.check(lib.hct_attention_fwd(qkv.data_ptr(), B, N, H, dh, dt, o.data_ptr(), lse.data_ptr(), _st()), "fwd")
            dqkv = torch.full_like(qkv, float("nan"))
            _lib.check(lib.hct_attention_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), B, N, H, dh, dt,
                                             dqkv.data_ptr(), _st()), "bwd")
            torch.cuda.synchronize()
        finally:
            lib.hct_debug_force_simple_attention(0)
            lib.hct_debug_force_simple_attention(10)
            lib.hct_debug_force_simple_attention(101206)
        assert rel_err(o, o_ref) < tol, force_simple
        assert (lse - lse_ref).abs().max() < (2e-2 if dtype == torch.bfloat16 else 1e-4)
        assert torch.isfinite(dqkv.float()).all()
        assert rel_err(dqkv, qr.grad) < 2 * tol, force_simple


@pytest.mark.parametrize("rows,D,dtype", [(55 * 2, 768, torch.bfloat16), (217, 768, torch.float32), (7, 48, torch.float32),
                                          (1030, 1024, torch.bfloat16), (5, 192, torch.bfloat16)])
def test_layernorm_fwd_bwd(lib, cuda, rows, D, dtype):
    x = _rand((rows, D), cuda, torch.float32, 21) * 2 + 0.3
    gamma, beta = 1 + 0.1 * _rand((D,), cuda, torch.float32, 22), 0.1 * _rand((D,), cuda, torch.float32, 23)
    dy = _rand((rows, D), cuda, dtype, 24)
    dres = _rand((rows, D), cuda, torch.float32, 25)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    (y_ref * dy.float()).sum().backward()
    y = torch.empty(rows, D, dtype=dtype, device=cuda)
    mean, rstd = torch.empty(rows, device=cuda), torch.empty(rows, device=cuda)
    _lib.check(lib.hct_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, D, 1e-5, y.data_ptr(), _dt(y),
                                     mean.data_ptr(), rstd.data_ptr(), _st()), "ln fwd")
    assert rel_err(y, y_ref) < (4e-3 if dtype == torch.bfloat16 else 1e-5)
    ws = torch.empty(lib.hct_layernorm_bwd_workspace_bytes(rows, D), dtype=torch.uint8, device=cuda)
    dx = dres.clone()
    shadow = torch.empty(rows, D, dtype=dtype, device=cuda)
    dg, db, dc = (torch.empty(D, device=cuda) for _ in range(3))
    _lib.check(lib.hct_layernorm_bwd(dy.data_ptr(), _dt(dy), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     dx.data_ptr(), rows, D, dx.data_ptr(), shadow.data_ptr(), _dt(shadow), dg.data_ptr(),
                                     db.data_ptr(), dc.data_ptr(), ws.data_ptr(), ws.numel(), _st()), "ln bwd")
    want = xr.grad + dres
    assert rel_err(dx, want) < 1e-5
    assert rel_err(shadow, want) < (4e-3 if dtype == torch.bfloat16 else 1e-5)
    assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    assert rel_err(dc, want.sum(0)) < 1e-4


def test_mask_rank_with_ties(lib, cuda):
    B, L, K = 5, 216, 54
    noise = torch.rand(B, L)
    noise[0, 10] = noise[0, 3]
    noise[1, :] = 0.5  # all ties: stable order = identity
    nz = noise.to(cuda)
    idr, ids = torch.empty(B, L, dtype=torch.int32, device=cuda), torch.empty(B, L, dtype=torch.int32, device=cuda)
    mask = torch.empty(B, L, device=cuda)
    _lib.check(lib.hct_mask_rank(nz.data_ptr(), B, L, K, idr.data_ptr(), ids.data_ptr(), mask.data_ptr(), _st()), "rank")
    shuffle = torch.argsort(noise, dim=1, stable=True)
    restore = torch.argsort(shuffle, dim=1, stable=True)
    assert torch.equal(ids.cpu().long(), shuffle) and torch.equal(idr.cpu().long(), restore)
    assert torch.equal(mask.cpu(), (restore >= K).float())
    assert torch.equal(ids[1].cpu().long(), torch.arange(L))


def test_tail_rows_gather_and_mapped_layernorm_bwd(lib, cuda):
    """The three pieces of the compact decoder tail: hct_tail_rows (rows of the masked patches in shuffle order + inverse),
    hct_gather_rows (gather, and scatter-with-zeros through the inverse map; bit-exact copies), hct_layernorm_bwd_mapped (residual
    gradient taken from the compact matrix through the inverse map)."""
    B, L, K, D = 3, 216, 54, 768
    noise = torch.rand(B, L)
    shuffle = torch.argsort(noise, dim=1, stable=True)
    restore = torch.argsort(shuffle, dim=1, stable=True).to(torch.int32).to(cuda)
    Lm, Md = L - K, B * (L + 1)
    rows = torch.empty(B * Lm, dtype=torch.int32, device=cuda)
    inv = torch.empty(Md, dtype=torch.int32, device=cuda)
    _lib.check(lib.hct_tail_rows(restore.data_ptr(), B, L, K, rows.data_ptr(), inv.data_ptr(), _st()), "tail_rows")
    want_rows = (torch.arange(B)[:, None] * (L + 1) + 1 + shuffle[:, K:]).reshape(-1)
    assert torch.equal(rows.cpu().long(), want_rows)
    want_inv = torch.full((Md,), -1, dtype=torch.long)
    want_inv[want_rows] = torch.arange(B * Lm)
    assert torch.equal(inv.cpu().long(), want_inv)
    for dtype in (torch.bfloat16, torch.float32):
        src = _rand((Md, D), cuda, dtype, 5)
        comp = torch.empty(B * Lm, D, dtype=dtype, device=cuda)
        _lib.check(lib.hct_gather_rows(src.data_ptr(), rows.data_ptr(), B * Lm, D * src.element_size(), comp.data_ptr(), _st()), "gather")
        assert torch.equal(comp, src[rows.long()])
        back = torch.full((Md, D), 7.0, dtype=dtype, device=cuda)
        _lib.check(lib.hct_gather_rows(comp.data_ptr(), inv.data_ptr(), Md, D * src.element_size(), back.data_ptr(), _st()), "scatter")
        want = torch.zeros_like(src)
        want[rows.long()] = src[rows.long()]
        assert torch.equal(back, want)
    # LayerNorm backward with the residual gradient read through the inverse map == the plain call on the scattered matrix
    x, dy = _rand((Md, D), cuda, torch.float32, 11), _rand((Md, D), cuda, torch.bfloat16, 12)
    gamma = _rand((D,), cuda, torch.float32, 13)
    dres_c = _rand((B * Lm, D), cuda, torch.float32, 14)
    dres_full = torch.zeros(Md, D, device=cuda)
    dres_full[rows.long()] = dres_c
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    ws = torch.empty(lib.hct_layernorm_bwd_workspace_bytes(Md, D), dtype=torch.uint8, device=cuda)
    outs = []
    for mapped in (False, True):
        dx = torch.empty(Md, D, device=cuda)
        shadow = torch.empty(Md, D, dtype=torch.bfloat16, device=cuda)
        dg, db, dc = (torch.empty(D, device=cuda) for _ in range(3))
        if mapped:
            _lib.check(lib.hct_layernorm_bwd_mapped(dy.data_ptr(), _dt(dy), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                                    dres_c.data_ptr(), inv.data_ptr(), Md, D, dx.data_ptr(), shadow.data_ptr(), _dt(shadow),
                                                    dg.data_ptr(), db.data_ptr(), dc.data_ptr(), ws.data_ptr(), ws.numel(), _st()), "ln bwd mapped")
        else:
            _lib.check(lib.hct_layernorm_bwd(dy.data_ptr(), _dt(dy), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                             dres_full.data_ptr(), Md, D, dx.data_ptr(), shadow.data_ptr(), _dt(shadow), dg.data_ptr(),
                                             db.data_ptr(), dc.data_ptr(), ws.data_ptr(), ws.numel(), _st()), "ln bwd")
        outs.append((dx, shadow, dg, db, dc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # a mapped residual gradient that aliases the output is refused
    assert lib.hct_layernorm_bwd_mapped(dy.data_ptr(), _dt(dy), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                        inv.data_ptr(), Md, D, dx.data_ptr(), None, 0, dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel(), _st()) != 0


def test_colsum_cast_transpose(lib, cuda):
    for rows, cols, dtype in ((1000, 768, torch.bfloat16), (37, 2304, torch.float32), (1, 4, torch.float32)):
        x = _rand((rows, cols), cuda, dtype, 31)
        ws = torch.empty(max(16, lib.hct_colsum_workspace_bytes(rows, cols)), dtype=torch.uint8, device=cuda)
        out = torch.empty(cols, device=cuda)
        _lib.check(lib.hct_colsum(x.data_ptr(), _dt(x), rows, cols, cols, out.data_ptr(), ws.data_ptr(), ws.numel(), _st()), "colsum")
        assert rel_err(out, x.float().sum(0)) < 1e-5
    w = _rand((300, 130), cuda, torch.float32, 32)
    wt = torch.empty(130, 300, dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.hct_transpose_cast(w.data_ptr(), HCT_F32, wt.data_ptr(), HCT_BF16, 300, 130, _st()), "tcast")
    assert torch.equal(wt, w.t().to(torch.bfloat16))
    flat = _rand((100003,), cuda, torch.float32, 33)
    fb = torch.empty(100003, dtype=torch.bfloat16, device=cuda)
    _lib.check(lib.hct_cast(flat.data_ptr(), HCT_F32, fb.data_ptr(), HCT_BF16, flat.numel(), _st()), "cast")
    assert torch.equal(fb, flat.to(torch.bfloat16))


def test_clip_and_adamw_vs_torch(lib, cuda):
    """hct_grad_norms + hct_adamw_step against torch.optim.AdamW + the reference's per-tensor clip."""
    sizes = [1024, 3072, 2048, 1024]
    seg = [0]
    for s in sizes:
        seg.append(seg[-1] + s)
    total = seg[-1]
    p0 = _rand((total,), cuda, torch.float32, 41)
    g0 = _rand((total,), cuda, torch.float32, 42)
    g0[seg[1]:seg[2]] *= 100.0  # this tensor gets clipped
    g0[seg[2]:seg[3]] *= 1e-3   # this one does not
    clip = 3.0
    params = [torch.nn.Parameter(p0[seg[i]:seg[i + 1]].clone()) for i in range(4)]
    opt = torch.optim.AdamW(params, lr=1e-3, betas=(0.9, 0.95), weight_decay=5e-3)
    p, g = p0.clone(), g0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    seg_t = torch.tensor(seg, dtype=torch.int64, device=cuda)
    skip = torch.tensor([0, 0, 0, 1], dtype=torch.uint8, device=cuda)
    norms, coef = torch.empty(4, device=cuda), torch.empty(4, device=cuda)
    ws = torch.empty(lib.hct_grad_norms_workspace_bytes(total), dtype=torch.uint8, device=cuda)
    shadow = torch.empty(total, dtype=torch.bfloat16, device=cuda)
    for step in range(1, 4):
        gstep = g0 * (1.0 + 0.1 * step)
        g.copy_(gstep)
        for i, prm in enumerate(params):
            prm.grad = gstep[seg[i]:seg[i + 1]].clone()
            n = prm.grad.norm(2)
            c = clip / (n + 1e-6)
            if c < 1:
                prm.grad.mul_(c)
        params[3].grad = None  # frozen segment
        opt.step()
        _lib.check(lib.hct_grad_norms(g.data_ptr(), seg_t.data_ptr(), 4, total, clip, 0, norms.data_ptr(), coef.data_ptr(),
                                      ws.data_ptr(), ws.numel(), _st()), "norms")
        _lib.check(lib.hct_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), seg_t.data_ptr(), coef.data_ptr(),
                                      skip.data_ptr(), 4, total, 1e-3, 0.9, 0.95, 1e-8, 5e-3, step, shadow.data_ptr(), _st()), "adamw")
        for i in range(3):
            assert rel_err(p[seg[i]:seg[i + 1]], params[i].data) < 1e-6, (step, i)
            assert rel_err(g[seg[i]:seg[i + 1]], params[i].grad) < 1e-6  # clipped gradient written back
        assert torch.equal(p[seg[3]:], p0[seg[3]:])  # skipped segment untouched
    assert float(coef[1]) < 1.0 and float(coef[2]) == 1.0
    assert torch.equal(shadow[:seg[3]], p[:seg[3]].to(torch.bfloat16))


def test_epilogue_gelu_accuracy_on_a_grid(lib, cuda):
    """The GELU / GELU' of the MFMA GEMM epilogues (clamped polynomial normal CDF, csrc/common.h) on a grid of pre-activations
    in [-10, 10]: against the exact-erf functions of the reference (nn.GELU(), MONAI MLPBlock) the error stays within the bf16
    rounding of the stored value plus 2e-4 (the polynomial's own bound is 1.4e-4 for gelu, 3.5e-5 for gelu')."""
    M, N, K = 256, 4096, 128
    x = torch.linspace(-10.0, 10.0, N, device=cuda)
    A = torch.zeros(M, K, dtype=torch.bfloat16, device=cuda)
    B = _rand((N, K), cuda, torch.bfloat16, 3)
    # forward: out = gelu(0 + bias), aux = the pre-activation as stored
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
    out = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=torch.bfloat16, bias=x, act=1, aux=aux)
    want = F.gelu(x)
    ulp = want.abs() * 2.0 ** -8 + 1e-30
    assert torch.equal(aux[0].float(), x.bfloat16().float())
    assert float(((out[0].float() - want).abs() - ulp).max()) < 2e-4 and torch.equal(out[0], out[-1])
    # backward factor: out = (ones . ones^T) * gelu'(aux)   (K ones: 128, exact in bf16)
    A1 = torch.ones(M, K, dtype=torch.bfloat16, device=cuda)
    B1 = torch.ones(N, K, dtype=torch.bfloat16, device=cuda)
    u = aux.float().requires_grad_(True)
    F.gelu(u).sum().backward()
    out2 = gemm(lib, A1, B1, 0, 1, M, N, K, out_dtype=torch.bfloat16, act=2, aux=aux)
    want2 = K * u.grad
    ulp2 = want2.abs() * 2.0 ** -8
    assert float(((out2.float() - want2).abs() - ulp2).max()) < K * 2e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_saved_gelu_derivative(lib, cuda, dtype):
    """HCT_ACT_GELU_D (out = gelu(acc + bias), aux <- gelu'(acc + bias)) and HCT_ACT_MULAUX (out = acc * aux): the pair the plan
    uses for the MLP (attentionblock.py:91 / MONAI MLPBlock), on the MFMA path (bf16) and the generic path (fp32)."""
    M, N, K = 300, 768, 256
    A = _rand((M, K), cuda, dtype, 61)
    B = _rand((N, K), cuda, dtype, 62, 0.05)
    bias = _rand((N,), cuda, torch.float32, 63)
    pre = (A.float() @ B.float().t() + bias).requires_grad_(True)
    F.gelu(pre).sum().backward()
    aux = torch.empty(M, N, dtype=dtype, device=cuda)
    out = gemm(lib, A, B, 0, 1, M, N, K, out_dtype=dtype, bias=bias, act=4, aux=aux)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-5
    assert rel_err(out, F.gelu(pre.detach())) < tol and rel_err(aux, pre.grad) < tol
    G = _rand((M, K), cuda, dtype, 64)
    out2 = gemm(lib, G, B, 0, 1, M, N, K, out_dtype=dtype, act=5, aux=aux)
    assert rel_err(out2, (G.float() @ B.float().t()) * aux.float()) < tol


def test_gemm_dgelu_fused_colsum(lib, cuda):
    """dgrad through GELU with the fused bias-gradient column sum (persistent NT kernel), whole and partial row tiles."""
    for M, N, K in ((512, 768, 256), (217 * 2, 3072, 768), (256, 256, 128), (1000, 3072, 256), (70000, 512, 128)):  # partial last row tiles; more tiles than CUs
        A = _rand((M, K), cuda, torch.bfloat16, 51)
        B = _rand((N, K), cuda, torch.bfloat16, 52, 0.05)
        aux = _rand((M, N), cuda, torch.bfloat16, 53)
        u = aux.float().requires_grad_(True)
        F.gelu(u).sum().backward()
        want = (A.float() @ B.float().t()) * u.grad
        a = GemmArgs()
        Cm = torch.empty(M, N, dtype=torch.bfloat16, device=cuda)
        cs = torch.full((N,), float("nan"), device=cuda)
        a.M, a.N, a.K = M, N, K
        a.A, a.a_dtype, a.lda, a.transA = A.data_ptr(), HCT_BF16, K, 0
        a.B, a.b_dtype, a.ldb, a.transB = B.data_ptr(), HCT_BF16, K, 1
        a.C, a.c_dtype, a.ldc = Cm.data_ptr(), HCT_BF16, N
        a.act, a.aux, a.aux_dtype, a.ldaux = 2, aux.data_ptr(), HCT_BF16, N
        a.alpha = 1.0
        a.colsum_out = cs.data_ptr()
        ws = torch.empty(lib.hct_gemm_workspace_bytes(C.byref(a)), dtype=torch.uint8, device=cuda)
        _lib.check(lib.hct_gemm(C.byref(a), ws.data_ptr(), ws.numel(), _st()), "gemm+colsum")
        assert rel_err(Cm, want) < 4e-3
        # the fused sums are taken before bf16 rounding, the fallback after: both within bf16 noise of the fp32 sum
        assert rel_err(cs, want.sum(0)) < 3e-3, (M, N, K)


def test_pos_embed_interp3d_vs_oracle_and_reference_fixture(lib, cuda):
    """hct_pos_embed_interp3d and the host mirror interpolate_pos_embed vs the oracle and the reference-generated fixture."""
    import json, os, types
    import numpy as np
    from oracle import mae_oracle as O
    from headct_foundation_amd.pos_embed import interpolate_pos_embed
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pos_interp.json")))
    for key, rec in fx.items():
        g_old, g_new, d, seed = (int(v) for v in key.split("_"))
        if d % 4:
            continue  # the kernel moves 4 channels per thread (embed dims of the path are multiples of 4)
        table = torch.from_numpy(O.hash_uniform(g_old ** 3 * d, seed).reshape(1, g_old ** 3, d).astype(np.float32))
        ref = torch.tensor(rec["ref"], dtype=torch.float32).reshape(1, g_new ** 3, d)
        fake = types.SimpleNamespace(patch_embedding=types.SimpleNamespace(
            n_patches=g_new ** 3, position_embeddings=torch.zeros(1, g_new ** 3, d, device=cuda)))
        ckpt = {"patch_embedding.position_embeddings": table.clone()}
        interpolate_pos_embed(fake, ckpt)
        out = ckpt["patch_embedding.position_embeddings"]
        assert out.device.type == "cpu" and out.shape == ref.shape
        assert float((out - ref).abs().max()) < 2e-6, key
        assert float((out - O.interpolate_pos_embed_3d(table, g_new, 0)).abs().max()) < 2e-6, key
    # class row kept, ViT-B-sized table (6^3 -> 8^3, D = 768) against the oracle
    t = torch.from_numpy(O.hash_uniform((1 + 216) * 768, 9).reshape(1, 217, 768).astype(np.float32))
    src = t.to(cuda)
    dst = torch.empty(1, 1 + 512, 768, device=cuda)
    _lib.check(lib.hct_pos_embed_interp3d(src.data_ptr(), 6, dst.data_ptr(), 8, 768, 1, _st()), "interp")
    torch.cuda.synchronize()
    want = O.interpolate_pos_embed_3d(t, 8, 1)
    assert torch.equal(dst[:, :1].cpu(), t[:, :1]) and float((dst.cpu() - want).abs().max()) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_augment_volume_vs_oracle(lib, cuda, dtype):
    """hct_augment_volume / DeviceAugment vs the oracle's restatement of the MAE input transforms (transforms.py:193-228).
    The reference's transforms are MONAI's (absent): parity against MONAI itself is unpinned; flips and the shift are exact ops."""
    from oracle import mae_oracle as O
    from headct_foundation_amd.data import DeviceAugment
    B, Cc, S = 9, 2, 16
    x = _rand((B, Cc, S, S, S), cuda, torch.float32, 21).abs().to(dtype)
    flips = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 5], dtype=torch.uint8)
    shifts = torch.tensor([0.0, 0.05, -0.1, 0.0, 0.0999, -0.03, 0.0, 0.07, 0.01])
    out = torch.empty(B, Cc, S, S, S, device=cuda)
    code = {torch.float16: _lib.HCT_F16, torch.bfloat16: _lib.HCT_BF16, torch.float32: _lib.HCT_F32}[dtype]
    flips_d, shifts_d = flips.to(cuda), shifts.to(cuda)  # keep both alive: temporaries would share one allocator block
    _lib.check(lib.hct_augment_volume(x.data_ptr(), code, out.data_ptr(), B, Cc, S, flips_d.data_ptr(), shifts_d.data_ptr(), _st()), "augment")
    torch.cuda.synchronize()
    want = O.augment_volume(x.cpu(), flips.tolist(), shifts.tolist())
    assert torch.equal(out.cpu(), want)  # cast, index remap and one fp32 add: bit-exact
    # NULL flags / offsets = plain cast
    _lib.check(lib.hct_augment_volume(x.data_ptr(), code, out.data_ptr(), B, Cc, S, None, None, _st()), "augment")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), x.cpu().float())
    # host mirror: draws exposed, statistics of the draws as configured
    aug = DeviceAugment(flip_prob=0.5, shift_offsets=0.1, shift_prob=0.5, seed=3)
    y = aug(x)
    f, sft = aug.last_draw
    assert torch.equal(y.cpu(), O.augment_volume(x.cpu(), f.tolist(), sft.tolist()))
    assert float(sft.abs().max()) <= 0.1


def test_gaussian_smooth_vs_oracle(lib, cuda):
    """hct_gaussian_smooth3d / DeviceAugment(smooth_prob) vs the oracle's restatement of RandGaussianSmoothd (transforms.py:230-238):
    per-sample, per-axis sigmas in [0.5, 1], zero padding, samples whose transform did not fire left bit-identical.  Parity with
    MONAI's GaussianFilter itself is unpinned (MONAI is not installed); against the oracle 2e-6 (sums of <= 9 fp32 products per pass)."""
    from oracle import mae_oracle as O
    from headct_foundation_amd.data import DeviceAugment, gaussian_smooth, gaussian_taps
    B, Cc, S = 5, 2, 16
    x = _rand((B, Cc, S, S, S), cuda, torch.float32, 31).abs()
    sigma = torch.tensor([[0.5, 0.75, 1.0], [1.0, 1.0, 1.0], [0.6, 0.5, 0.9], [0.5, 0.5, 0.5], [0.99, 0.51, 0.7]])
    fire = torch.tensor([True, True, False, True, True])
    got = gaussian_smooth(x, sigma, fire)
    want = O.gaussian_smooth3d(x.cpu(), sigma.tolist(), fire.tolist())
    assert torch.equal(got[2].cpu(), x[2].cpu())
    assert float((got.cpu() - want).abs().max()) < 2e-6
    # the host taps are the oracle's kernels, zero beyond the tail
    taps = gaussian_taps(sigma)
    for b in range(B):
        for a in range(3):
            k = O.gaussian_kernel_1d(float(sigma[b, a]))
            pad = (9 - k.numel()) // 2
            assert torch.equal(taps[b, a, pad:9 - pad], k) and float(taps[b, a, :pad].abs().sum() + taps[b, a, 9 - pad:].abs().sum()) == 0.0
    # a constant volume stays constant away from the borders (the erf kernel sums to 1 within 1e-4)
    c = gaussian_smooth(torch.full((1, 1, S, S, S), 0.7, device=cuda), torch.tensor([[1.0, 1.0, 1.0]]))
    half = float(O.gaussian_kernel_1d(1.0)[4:].sum())  # zero padding: the corner keeps the in-volume half of each 1-D kernel
    assert float((c[0, 0, 4:-4, 4:-4, 4:-4] - 0.7).abs().max()) < 1e-4 and abs(float(c[0, 0, 0, 0, 0]) - 0.7 * half ** 3) < 1e-6
    # inside DeviceAugment: same draws -> same volumes as the oracle's chain
    aug = DeviceAugment(flip_prob=0.5, shift_offsets=0.1, shift_prob=0.5, seed=2, smooth_prob=0.6)  # seed 2: samples 1, 3, 4 fire
    y = aug(x.half())
    f, sft, fired, sg = aug.last_draw
    ref = O.gaussian_smooth3d(O.augment_volume(x.half().cpu(), f.tolist(), sft.tolist()), sg.tolist(), fired.tolist())
    assert bool(fired.any()) and not bool(fired.all())
    assert float((y.cpu() - ref).abs().max()) < 2e-6
    with pytest.raises(ValueError):
        gaussian_taps(torch.tensor([1.5]))


@pytest.mark.parametrize("channels", [1, 3])
def test_hu_window_vs_oracle(lib, cuda, channels):
    """HU windowing of loading_transforms (transforms.py:108-133) on the device: bit-equal to the oracle's fp32 restatement, and
    equal to it after the cache's fp16 cast; window edges, values far outside and fp16 HU input included.  (Parity with MONAI's
    ScaleIntensityRange itself is unpinned: MONAI is not installed.)"""
    from headct_foundation_amd.data import window_hu
    from oracle import mae_oracle as O
    g = torch.Generator(device="cpu").manual_seed(3)
    hu = (torch.rand(2, 1, 8, 12, 16, generator=g) * 4000 - 1200).round()
    hu.view(-1)[:12] = torch.tensor([-110., 190., -111., 191., 0., 80., -20., 180., -800., 2000., -3000., 5000.])
    want = O.hu_window(hu, channels)
    got32 = window_hu(hu.to(cuda), channels, out_dtype=torch.float32)
    assert got32.shape == want.shape and torch.equal(got32.cpu(), want)
    got16 = window_hu(hu.to(cuda), channels)
    assert got16.dtype == torch.float16 and torch.equal(got16.cpu(), want.half())
    got_h = window_hu(hu.half().to(cuda), channels, out_dtype=torch.float32)  # integer HU up to 2048 are exact in fp16
    assert torch.equal(got_h.cpu(), O.hu_window(hu.half().float(), channels))
    assert float(got32.min()) == 0.0 and float(got32.max()) == 1.0


@pytest.mark.parametrize("B,C,S,P,xdt,rdt", [(2, 3, 96, 12, torch.float32, torch.bfloat16), (3, 1, 96, 16, torch.float16, torch.bfloat16),
                                            (2, 3, 48, 12, torch.float32, torch.float32), (1, 2, 32, 8, torch.float16, torch.float32),
                                            (1, 1, 128, 16, torch.float32, torch.bfloat16)])
def test_patch_gather_all_patches(lib, cuda, B, C, S, P, xdt, rdt):
    """hct_patch_gather without an index table (plain ViT / DINO crops: every patch in grid order) -- the pencil kernel where its LDS image
    fits, the per-patch kernel otherwise -- against the Conv3d unfolding order (c, ph, pw, pd) of patch_embedding.py:149, and against the
    same call WITH an identity table (the per-patch kernel): bit-equal."""
    g = S // P
    L = g ** 3
    x = _rand((B, C, S, S, S), cuda, torch.float32, 77).to(xdt)
    ref = x.float().view(B, C, g, P, g, P, g, P).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * L, C * P ** 3)
    xd = _lib.HCT_F16 if xdt == torch.float16 else HCT_F32
    rows = torch.full((B * L, C * P ** 3), float("nan"), dtype=rdt, device=cuda)
    _lib.check(lib.hct_patch_gather(x.data_ptr(), xd, None, B, C, S, P, L, L, rows.data_ptr(), _dt(rows), _st()), "patch_gather")
    ids = torch.arange(L, dtype=torch.int32, device=cuda).repeat(B, 1).contiguous()
    rows2 = torch.full_like(rows, float("nan"))
    _lib.check(lib.hct_patch_gather(x.data_ptr(), xd, ids.data_ptr(), B, C, S, P, L, L, rows2.data_ptr(), _dt(rows), _st()), "patch_gather")
    torch.cuda.synchronize()
    assert torch.equal(rows, rows2)
    assert torch.equal(rows.float(), ref.to(rdt).float())
