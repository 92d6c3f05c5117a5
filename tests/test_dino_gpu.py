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


def _build_pair(fx, dev, dtype, which):
    """MultiCropWrapper(ViTBackbone, DINOHead) with the fixture's hash weights (which = 0 student, 1 teacher)."""
    from headct_foundation_amd.dino_model import DINOHead, MultiCropWrapper, ViTBackbone
    vk, hk = dict(fx["vit"]), dict(fx["head"])
    b = ViTBackbone(**vk, compute_dtype=dtype)
    assert list(b.state_dict().keys()) == fx["backbone_keys"]  # the reference ViT's names and registration order
    pb = O.make_vit_params({k: list(v.shape) for k, v in b.state_dict().items()}, seed0=fx["backbone_seed"][which])
    b.load_state_dict(pb, strict=True)
    h = DINOHead(**hk, compute_dtype=dtype)
    ph = D.make_head_params(hk["in_dim"], hk["out_dim"], hk["hidden_dim"], hk["bottleneck_dim"], fx["head_seed"][which])
    assert list(h.state_dict().keys()) == list(ph.keys())
    h.load_state_dict(ph, strict=True)
    return MultiCropWrapper(b, h).to(dev), pb, ph


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_dino_head_with_batchnorm_vs_reference_fixture(lib, cuda, dtype, tol):
    """DINOHead(use_bn=True) on the HIP path (Linear -> hct_batchnorm_stats + hct_bn_gelu_fwd -> ..., backward through
    hct_bn_gelu_bwd_sums / _apply) against the reference module's own outputs (tests/golden/dino.json["head_bn"]) and the oracle:
    state-dict keys, training output, input and parameter gradients, running statistics / num_batches_tracked after one forward,
    eval-mode output.  fp32 1e-3 (north_star); bf16 storage 6e-2 (ten rows: the batch statistics of so few rows amplify the rounding of
    the bf16 operands; the whole-step test above holds bf16 to 5e-2 on its larger batch)."""
    from headct_foundation_amd.dino_model import DINOHead
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))["head_bn"]
    p = D.make_head_bn_params(fx)
    head = DINOHead(fx["in_dim"], fx["out_dim"], use_bn=True, norm_last_layer=True, nlayers=3, hidden_dim=fx["hidden"], bottleneck_dim=fx["bottleneck"],
                    compute_dtype=dtype)
    assert list(head.state_dict().keys()) == fx["keys"]
    head.load_state_dict(p, strict=True)
    head = head.to(cuda).train()
    x = _hu((fx["rows"], fx["in_dim"]), fx["x_seed"], -1, 1).to(cuda).requires_grad_(True)
    dy = _hu((fx["rows"], fx["out_dim"]), fx["dy_seed"], -1, 1).to(cuda)
    y = head(x)
    (y.float() * dy).sum().backward()
    torch.cuda.synchronize()
    got, want, l2, l2w = sample_of(y.float(), fx["y"])
    assert abs(l2 - l2w) < tol * l2w and float((got - want).norm() / want.norm()) < tol
    got, want, l2, l2w = sample_of(x.grad.float(), fx["dx"])
    assert float((got - want).norm() / want.norm()) < tol
    gscale = max(e["l2"] for e in fx["grads"].values())
    for n, e in fx["grads"].items():
        g = dict(head.named_parameters())[n].grad
        got, want, l2, l2w = sample_of(g.float(), e)
        if n in ("mlp.0.bias", "mlp.3.bias"):  # mathematically zero (the BatchNorm removes the mean): round-off of the column sums
            assert float(g.abs().max()) < (1e-4 if dtype == "fp32" else 5e-2) * gscale, n
        else:
            assert float((got - want).norm() / (want.norm() + 1e-30)) < tol, (n, float((got - want).norm() / (want.norm() + 1e-30)))
    sd = head.state_dict()
    for n, want in fx["running_after"].items():
        assert torch.allclose(sd[n].cpu().flatten(), torch.tensor(want), rtol=tol, atol=tol * 1e-2), n
    assert int(sd["mlp.1.num_batches_tracked"]) == fx["num_batches_tracked"] == int(sd["mlp.4.num_batches_tracked"])
    head.eval()
    with torch.no_grad():
        ye = head(x.detach())
    got, want, l2, l2w = sample_of(ye.float(), fx["y_eval"])
    assert float((got - want).norm() / want.norm()) < tol
    assert int(head.state_dict()["mlp.1.num_batches_tracked"]) == fx["num_batches_tracked"]  # eval does not move the statistics


@pytest.mark.gpu
def test_backbone_reads_crops_in_place(lib, cuda):
    """ViTBackbone on a list of equally shaped crops (hct_vit_forward_parts: patch rows gathered straight from each tensor) is
    bit-identical, forward and backward, to the same backbone on their concatenation (MultiCropWrapper's torch.cat,
    misc.py:467-480); unequal parts fall back to the copy."""
    from headct_foundation_amd.dino_model import ViTBackbone
    torch.manual_seed(3)
    b = ViTBackbone(img_size=32, patch_size=16, in_chans=3, hidden_size=192, mlp_dim=384, num_layers=2, num_heads=3,
                    num_register_tokens=2, compute_dtype="bf16").to(cuda)
    crops = [torch.rand(2, 3, 32, 32, 32, device=cuda) for _ in range(3)]
    outs, grads = [], []
    for arg in (torch.cat(crops), crops, tuple(crops), [crops[0], torch.cat(crops[1:])]):
        for p in b.parameters():
            p.grad = None
        out = b(arg)[0]
        out.float().square().mean().backward()
        torch.cuda.synchronize()
        outs.append(out.detach().clone())
        grads.append({n: p.grad.detach().clone() for n, p in b.named_parameters() if p.grad is not None})
    for o, g in zip(outs[1:], grads[1:]):
        assert torch.equal(o, outs[0])
        assert g.keys() == grads[0].keys() and all(torch.equal(g[k], grads[0][k]) for k in g)


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [("fp32", 2e-5, 1e-3), ("bf16", 2e-3, 5e-2)])
def test_dino_step_vs_reference_fixture(lib, cuda, dtype, tol_loss, tol_grad):
    """One DINO iteration (engine_pretrain_dino.py:59-104: teacher on the two global crops, student on all crops through
    MultiCropWrapper, DINOLoss, backward, last-layer gradients cancelled, centre update, momentum-teacher update) on the HIP path
    against the fixture produced by the reference's own ViT / DINOHead / MultiCropWrapper / DINOLoss (tests/golden/dino.json,
    'step') and against the oracle.  fp32 mode: loss 2e-5, every gradient 1e-3 relative L2; bf16 mode: 2e-3 / 5e-2."""
    from headct_foundation_amd.dino import DINOLoss, update_momentum_encoder
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))["step"]
    student, sb, sh = _build_pair(fx, cuda, dtype, 0)
    teacher, tb, th = _build_pair(fx, cuda, dtype, 1)
    V, Bc = fx["crops"], fx["batch"]
    S = fx["vit"]["img_size"]
    crops = [_hu((Bc, 3, S, S, S), fx["crop_seed0"] + i, 0.0, 1.0) for i in range(V)]
    center0 = _hu((1, fx["head"]["out_dim"]), fx["center_seed"], -0.2, 0.2)
    crit = DINOLoss(fx["head"]["out_dim"], V, 0.04, 0.07, 3, 10).to(cuda)
    crit.center.copy_(center0)
    dcrops = [c.to(cuda) for c in crops]
    with torch.no_grad():
        t_out = teacher(dcrops[:2])['dino_output']
    s_out = student(dcrops)['dino_output']
    loss = crit(s_out.float(), t_out.float(), 0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - fx["loss"]) < tol_loss * abs(fx["loss"]), (float(loss), fx["loss"])
    got, want, l2, l2w = sample_of(s_out.float(), fx["logits"])
    assert abs(l2 - l2w) < (1e-4 if dtype == "fp32" else 2e-2) * l2w
    grads = {("backbone." + n): p.grad for n, p in student.backbone.named_parameters() if p.grad is not None}
    grads.update({("head." + n): p.grad for n, p in student.head.named_parameters() if p.grad is not None})
    # cancel_gradients_last_layer (misc.py:366-371): the fixture was taken at epoch 0 < FREEZE_LAST_LAYER
    grads = {n: g for n, g in grads.items() if "last_layer" not in n}
    assert set(grads) == set(fx["grads"]), set(grads) ^ set(fx["grads"])
    o_loss, o_gb, o_gh, o_center = D.dino_step(sb, sh, tb, th, crops, center0, patch=fx["vit"]["patch_size"], heads=fx["vit"]["num_heads"],
                                               layers=fx["vit"]["num_layers"], student_temp=0.1, teacher_temp=fx["teacher_temp"])
    worst = []
    for n, g in grads.items():
        o = o_gb[n[len("backbone."):]] if n.startswith("backbone.") else o_gh[n[len("head."):]]
        worst.append((rel_err(g.float(), o), n))
        got, want, l2, l2w = sample_of(g.float(), fx["grads"][n])
        if not n.endswith("qkv.bias"):
            assert abs(l2 - l2w) <= tol_grad * l2w + 1e-12, n
        else:
            # config #5 runs with USE_BIAS: only the K third of this gradient is mathematically zero (softmax is invariant to a key
            # bias) -- its round-off noise is compared absolutely; the Q and V thirds (column sums of dqkv) like any other gradient
            third = g.numel() // 3
            gq, gk, gv = g.float()[:third], g.float()[third:2 * third], g.float()[2 * third:]
            oq, ok, ov = o[:third], o[third:2 * third], o[2 * third:]
            assert rel_err(gq, oq) < tol_grad and rel_err(gv, ov) < tol_grad, (n, rel_err(gq, oq), rel_err(gv, ov))
            assert float(gk.abs().max()) <= 1e-6 + tol_grad * float(torch.cat([oq, ov]).abs().max()), n
            assert torch.allclose(got, want, rtol=0.0, atol=1e-6 + 2 * tol_grad * float(want.abs().max())), n
    bad = [w for w in worst if not w[1].endswith("qkv.bias")]
    assert max(bad)[0] < tol_grad, sorted(bad)[-5:]
    got, want, _, _ = sample_of(crit.center, fx["center_after"])
    assert torch.allclose(got, want, rtol=1e-3 if dtype == "fp32" else 2e-2, atol=1e-5 if dtype == "fp32" else 2e-3)
    # momentum teacher: one fused launch per flat buffer (backbone, head)
    update_momentum_encoder(student.backbone, teacher.backbone, 0.996)
    update_momentum_encoder(student.head, teacher.head, 0.996)
    torch.cuda.synchronize()
    tnamed = dict(("backbone." + n, p) for n, p in teacher.backbone.named_parameters())
    tnamed.update(("head." + n, p) for n, p in teacher.head.named_parameters())
    for n, entry in fx["teacher_after"].items():
        got, want, _, _ = sample_of(tnamed[n], entry)
        assert torch.equal(got, want), n  # exact: the same fp32 operations in the same order


def test_main_pretrain_dino_plumbing_run(cuda, tmp_path):
    """python -m torch.distributed.run --nproc-per-node 1 main_pretrain_dino.py ... (toy ViT, 2 + 2 crops of 24^3 x 3ch, 2 epochs with
    validation, checkpoints with the momentum model, test): the reference's entry-point flow on the HIP path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29537",
           os.path.join(root, "main_pretrain_dino.py"), "--local_rank", "0", "--model_name", "dino", "--batch_size", "2", "--max_epochs", "2",
           "--base_lr", "5e-4", "--cfg", os.path.join(root, "configs/dino/dino_tiny_plumbing.yaml"), "--optimizer", "AdamW", "--scheduler", "cosine",
           "--opts", "MODEL.DIR", str(tmp_path / "ckpt"), "LOG.OUTPUT_DIR", str(tmp_path / "log"), "OUTPUT", str(tmp_path / "json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "train completed" in r.stdout and "test completed" in r.stdout
    ck = torch.load(tmp_path / "ckpt" / "last_dino_tiny.pt", map_location="cpu", weights_only=True)
    assert sorted(ck.keys()) == ["best_loss", "epoch", "momentum_model_state_dict", "optimizer", "scheduler", "state_dict"]
    assert "module.backbone.blocks.0.attn.qkv.weight" in ck["state_dict"] and "module.head.last_layer.weight_v" in ck["state_dict"]
    assert ck["momentum_model_state_dict"] is not None and set(ck["momentum_model_state_dict"]) == set(ck["state_dict"])
    # the teacher follows the student by the momentum update only: close to it, not equal
    a, b = ck["state_dict"]["module.backbone.norm.weight"], ck["momentum_model_state_dict"]["module.backbone.norm.weight"]
    assert not torch.equal(a, b) and float((a - b).abs().max()) < 1e-2
    assert (tmp_path / "ckpt" / "best_dino_tiny.pt").exists()


def test_dino_full_size_step_properties(lib, cuda):
    """BASELINE config #5 at full width and depth (ViT-B/12^3: 517 tokens, 12 blocks, head 768-2048-2048-256-65536, 2 global + 8
    local crops of 96^3 x 3 channels, bf16): too large for the oracle, so the size-independent properties -- the iteration is
    bit-reproducible, every trainable parameter receives a finite gradient, the frozen prototype scale receives none, an optimizer
    step on the same batch lowers the loss, and after the momentum update the teacher lies between its old value and the student."""
    import bench
    from headct_foundation_amd.dino import DINOLoss, DinoOptimizer, update_momentum_encoder
    from headct_foundation_amd.dino_model import DINOHead, MultiCropWrapper, ViTBackbone
    torch.manual_seed(3)
    mk = lambda: MultiCropWrapper(ViTBackbone(**bench.DINO["vit"], compute_dtype="bf16"), DINOHead(**bench.DINO["head"], compute_dtype="bf16")).to(cuda)
    student, teacher = mk(), mk()
    teacher.load_state_dict(student.state_dict())
    B, V = 2, bench.DINO["crops"]
    g = torch.Generator(device=cuda)
    g.manual_seed(5)
    crops = [torch.rand(B, 3, 96, 96, 96, device=cuda, generator=g) for _ in range(V)]
    crit = DINOLoss(65536, V, 0.04, 0.04, 30, 200).to(cuda)

    def iteration():
        for p in student.parameters():
            p.grad = None
        crit.center.zero_()
        with torch.no_grad():
            t_out = teacher(crops[:2])['dino_output']
        loss = crit(student(crops)['dino_output'].float(), t_out.float(), 0)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}

    l1, g1 = iteration()
    l2, g2 = iteration()
    assert l1 == l2 and all(torch.equal(g1[k], g2[k]) for k in g1), "the DINO iteration is not bit-reproducible"
    assert 9.5 < l1 < 11.5  # around ln(65536) = 11.09 at initialisation (student near uniform over the prototypes)
    trainable = [n for n, p in student.named_parameters() if p.requires_grad]
    assert set(g1) == set(trainable) and "head.last_layer.weight_g" not in g1
    assert all(torch.isfinite(v).all() for v in g1.values()) and all(float(v.abs().max()) > 0 for v in g1.values())
    opt = DinoOptimizer(student, lr=2e-4, weight_decay=0.04)
    opt.step()
    before = teacher.backbone.norm.weight.detach().clone()
    update_momentum_encoder(student.backbone, teacher.backbone, 0.9)
    after, stud = teacher.backbone.norm.weight.detach(), student.backbone.norm.weight.detach()
    assert torch.allclose(after, before * 0.9 + stud * (1 - 0.9), rtol=1e-6, atol=1e-7)
    teacher.load_state_dict(student.state_dict())  # identical networks see identical logits on the global crops ...
    l3, _ = iteration()
    assert l3 < l1, (l1, l3)
