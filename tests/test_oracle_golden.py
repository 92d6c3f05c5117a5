"""CPU: the oracle (oracle/mae_oracle.py) against the golden fixtures generated from the REFERENCE itself
(tests/golden/make_golden.py imports the reference's own modules with stand-ins for MONAI/timm symbols)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mae_oracle as O
from tests.util import GOLDEN, load_golden, sample_of

CASES = [("micro", 2, 0), ("yaml_cut", 2, 1), ("tiny", 2, 0), ("vitb_cut", 2, 0)]


@pytest.mark.parametrize("name,batch,seed", CASES)
def test_oracle_forward_backward_matches_reference_outputs(name, batch, seed):
    fx = load_golden(f"{name}_b{batch}_s{seed}")
    cfg = O.CONFIGS[name]
    params = O.make_params(cfg, seed)
    x, noise = O.make_volume(cfg, batch, seed), O.make_noise(cfg, batch, seed)
    loss, pred, mask, grads, inter = O.forward_backward(cfg, params, x, noise, want_inter=True)
    assert abs(float(loss) - fx["loss"]) <= 2e-6 * abs(fx["loss"])
    assert float(mask.sum()) == fx["mask_sum"]
    for key, entry in fx["act"].items():
        got, want, l2, l2w = sample_of(inter[key], entry)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), key
        assert abs(l2 - l2w) <= 1e-5 * l2w
    got, want, l2, l2w = sample_of(pred, fx["pred"])
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)
    got, want, _, _ = sample_of(O.unpatchify(cfg, pred), fx["unpatchify_pred"])
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)
    got, want, _, _ = sample_of(O.patchify(cfg, x), fx["patchify_x"])
    assert torch.equal(got, want)
    assert set(grads) == set(fx["grads"])
    for k, entry in fx["grads"].items():
        got, want, l2, l2w = sample_of(grads[k], entry)
        if k.endswith("qkv.bias"):  # K-third: mathematically zero gradient, round-off only
            assert (got - want).abs().max() < 1e-7 + 1e-4 * float(want.abs().max())
        else:
            assert torch.allclose(got, want, rtol=2e-4, atol=1e-5 * float(want.abs().max()) + 1e-12), k


@pytest.mark.parametrize("name,batch,seed", [("micro", 2, 0), ("tiny", 2, 0)])
def test_oracle_train_curve_matches_reference_engine(name, batch, seed):
    """loss curve / LR values / parameters after N steps of the reference's own train_one_epoch."""
    fx = load_golden(f"{name}_b{batch}_s{seed}")
    hp, steps = fx["train"]["hp"], fx["train"]["steps"]
    cfg = O.CONFIGS[name]
    st = O.TrainState(O.make_params(cfg, seed))
    losses, lrs = [], []
    for i in range(steps):
        l, lr, _, _ = O.train_step(cfg, st, O.make_volume(cfg, batch, seed + 10 + i), O.make_noise(cfg, batch, seed + 10 + i), **hp)
        losses.append(l); lrs.append(lr)
    assert np.allclose(lrs, fx["train"]["lrs"], rtol=1e-12)
    assert np.allclose(losses, fx["train"]["logged_losses"], atol=6e-5)
    for k, entry in fx["train"]["params_after"].items():
        got, want, _, _ = sample_of(st.params[k], entry)
        atol = 4 * hp["base_lr"] if k.endswith("qkv.bias") else 1e-6
        assert torch.allclose(got, want, rtol=1e-5, atol=atol), k


def test_oracle_full_tensors_micro():
    z = np.load(os.path.join(GOLDEN, "micro_b2_s0_full.npz"))
    cfg = O.CONFIGS["micro"]
    loss, pred, _, grads, inter = O.forward_backward(cfg, O.make_params(cfg, 0), O.make_volume(cfg, 2, 0), O.make_noise(cfg, 2, 0), True)
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    assert np.allclose(pred.numpy(), z["pred"], rtol=1e-4, atol=1e-5)
    assert np.allclose(inter["latent"].numpy(), z["latent"], rtol=1e-4, atol=1e-5)
    for k, g in grads.items():
        if not k.endswith("qkv.bias"):
            assert np.allclose(g.numpy(), z["grad." + k], rtol=1e-3, atol=1e-6 * np.abs(z["grad." + k]).max() + 1e-9), k


def test_lr_schedule_and_sincos_golden():
    with open(os.path.join(GOLDEN, "lr_schedule.json")) as f:
        fx = json.load(f)
    got = [fx["base_lr"] * O.cosine_warmup_lambda(s, fx["warmup"], fx["total"], fx["base_lr"], fx["min_lr"]) for s in range(len(fx["lrs"]))]
    assert np.allclose(got, fx["lrs"], rtol=1e-12)
    with open(os.path.join(GOLDEN, "sincos.json")) as f:
        sc = json.load(f)
    for key, entry in sc.items():
        g, d = map(int, key.split("_"))
        got, want, _, _ = sample_of(O.build_sincos_position_embedding_3d(g, d), entry)
        assert torch.equal(got, want)


def test_algorithmic_flops_match_survey():
    f = O.algorithmic_flops_per_volume(O.CONFIGS["vitb"])
    assert abs(f["train"] / 1e9 - 110.85) < 0.01  # SURVEY 8d / BASELINE.md
    assert abs(O.algorithmic_flops_per_volume(O.CONFIGS["tiny"])["train"] / 1e9 - 1.295) < 0.001


def test_pos_embed_interpolation_against_reference_fixture():
    """oracle.interpolate_pos_embed_3d vs outputs of the reference's interpolate_pos_embed (pos_embed.py:102-153)."""
    import json, os
    import numpy as np
    import torch
    from oracle import mae_oracle as O
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pos_interp.json")))
    assert len(fx) >= 4
    for key, rec in fx.items():
        g_old, g_new, d, seed = (int(v) for v in key.split("_"))
        table = torch.from_numpy(O.hash_uniform(g_old ** 3 * d, seed).reshape(1, g_old ** 3, d).astype(np.float32))
        out = O.interpolate_pos_embed_3d(table, g_new, 0)
        ref = torch.tensor(rec["ref"], dtype=torch.float32).reshape(1, g_new ** 3, d)
        assert out.shape == ref.shape
        assert float((out - ref).abs().max()) < 2e-6, key  # fp32 trilinear weights; summation order differs from ATen's
    # extra (class) tokens are carried over unchanged, equal grids are a no-op
    t = torch.arange(2 * (8 + 1) * 4, dtype=torch.float32).reshape(1, 2 * 9, 4)[:, :9]
    o = O.interpolate_pos_embed_3d(t, 3, 1)
    assert o.shape == (1, 28, 4) and torch.equal(o[:, :1], t[:, :1])
    assert torch.equal(O.interpolate_pos_embed_3d(t, 2, 1), t)


def test_vit_feature_extraction_against_reference_fixture():
    """oracle.vit_forward vs the reference's ViT.forward (src/models/vit.py:144-173): register tokens, final LN eps 1e-6."""
    import json, os
    import numpy as np
    import torch
    from oracle import mae_oracle as O
    from tests.util import sample_of
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vit_features.json")))
    params = O.make_vit_params({k: v["shape"] for k, v in fx["state_dict"].items()})
    x = torch.from_numpy(O.hash_uniform(2 * 32 ** 3, 7).reshape(2, 1, 32, 32, 32).astype(np.float32)) * 0.5 + 0.5
    out, hidden = O.vit_forward(params, x, 16, 3, 2)
    assert list(out.shape) == fx["out"]["shape"] == [2, 11, 192] and len(hidden) == 2
    for t, entry in [(out, fx["out"])] + list(zip(hidden, fx["hidden"])):
        got, want, l2, l2w = sample_of(t, entry)
        assert torch.allclose(got, want, rtol=0, atol=5e-6) and abs(l2 - l2w) < 1e-4 * l2w
    # same weights on a 48^3 volume: the position table is resized for the call (patch_embedding.py:136-144)
    r = fx["resized_48"]
    x48 = torch.from_numpy(O.hash_uniform(2 * 48 ** 3, r["x_seed"]).reshape(2, 1, 48, 48, 48).astype(np.float32)) * 0.5 + 0.5
    out, hidden = O.vit_forward(params, x48, 16, 3, 2)
    assert list(out.shape) == r["out"]["shape"] == [2, 30, 192]
    for t, entry in [(out, r["out"])] + list(zip(hidden, r["hidden"])):
        got, want, l2, l2w = sample_of(t, entry)
        assert torch.allclose(got, want, rtol=0, atol=5e-6) and abs(l2 - l2w) < 1e-4 * l2w


def _head_inputs(name, e):
    """Parameters and input of one entry of tests/golden/classifier_heads.json, rebuilt from their seeds."""
    import numpy as np
    import torch
    from oracle import mae_oracle as O
    params = O.make_vit_params({k: v["shape"] for k, v in e["state_dict"].items()}, seed0=e["seed0"])
    if name.startswith("attention"):
        params["cls_token"] = params["cls_token"] * e["cls_token_gain"]
        shp = e["x_shape"]
        x = torch.from_numpy(O.hash_uniform(int(np.prod(shp)), e["x_seed"]).reshape(shp).astype(np.float32)) * e["x_gain"]
    elif name.startswith("linear"):
        shp = e["x_shape"]
        x = torch.from_numpy(O.hash_uniform(int(np.prod(shp)), e["x_seed"]).reshape(shp).astype(np.float32))
    else:
        for k in params:
            if k.startswith("classification_head") and k.endswith("weight"):
                params[k] = params[k] * e["head_weight_gain"]
        x = torch.from_numpy(O.hash_uniform(2 * 32 ** 3, 7).reshape(2, 1, 32, 32, 32).astype(np.float32)) * 0.5 + 0.5
    return params, x


def _head_oracle(name, e, params, x):
    from oracle import mae_oracle as O
    if name == "linear":
        return O.linear_classifier_forward(params, x)
    if name.startswith("attention"):
        return O.attention_classifier_forward(params, x, e["ctor"]["num_heads"], e["ctor"]["num_queries"])
    return O.vit_forward(params, x, 16, 3, 2)[0]


def test_classifier_heads_oracle_matches_reference_fixture():
    """LinearClassifier / AttentionClassifier (eval) and ViT(classification=True): the oracle against outputs of the
    reference's own modules (tests/golden/make_golden.py: classifier_fixture)."""
    import torch
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "classifier_heads.json")))
    assert set(fx) == {"linear", "linear_probe_step", "attention_q1", "attention_q3", "vit_tanh", "vit_linear"}
    for name, e in fx.items():
        if name == "linear_probe_step":
            continue
        params, x = _head_inputs(name, e)
        out = _head_oracle(name, e, params, x)
        want = torch.tensor(e["out"]).reshape(out.shape)
        assert torch.allclose(out, want, rtol=0, atol=5e-6), name


def test_classifier_modules_mirror_reference_state_dict():
    """Host mirrors register the reference's parameter / buffer names in the reference's order (no GPU needed)."""
    from headct_foundation_amd import AttentionClassifier, LinearClassifier, ViT
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "classifier_heads.json")))
    for name, e in fx.items():
        cls = LinearClassifier if name.startswith("linear") else AttentionClassifier if name.startswith("attention") else ViT
        m = cls(**e["ctor"])
        sd = m.state_dict()
        assert list(sd.keys()) == list(e["state_dict"].keys()), name
        assert all(list(sd[k].shape) == v["shape"] for k, v in e["state_dict"].items()), name
    import pytest
    import torch
    with pytest.raises(Exception, match="eval"):
        AttentionClassifier(8, 2, num_heads=2)(torch.zeros(1, 3, 8))
    with pytest.raises(Exception, match="GPU"):
        LinearClassifier(8, 2)(torch.zeros(2, 8))
    with pytest.raises(Exception, match="GPU"):
        LinearClassifier(8, 2).eval()(torch.zeros(1, 8))


def test_linear_probe_step_oracle_matches_reference_fixture():
    """LinearClassifier.train() + nn.CrossEntropyLoss() + backward of the reference (one step) vs the oracle."""
    import torch
    from oracle import mae_oracle as O
    e = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "classifier_heads.json")))["linear_probe_step"]
    params, x = _head_inputs("linear_probe_step", e)
    logits, loss, grads, stats = O.linear_probe_step(params, x, torch.tensor(e["target"]))
    close = lambda a, b: torch.allclose(a.flatten(), torch.tensor(b).flatten(), rtol=0, atol=2e-6)
    assert close(logits, e["logits"]) and abs(float(loss) - e["loss"]) < 2e-6
    assert close(grads["linear.weight"], e["grad_weight"]) and close(grads["linear.bias"], e["grad_bias"])
    assert close(stats["bn.running_mean"], e["running_mean"]) and close(stats["bn.running_var"], e["running_var"])


def _hu(shape, seed, lo, hi):
    import numpy as np
    return torch.from_numpy(O.hash_uniform(int(np.prod(shape)), seed)).float().reshape(shape) * (hi - lo) + lo


def test_dino_oracle_vs_reference_fixture():
    """oracle/dino_oracle.py against tests/golden/dino.json (outputs of the reference's DINOLoss, _update_momentum_encoder,
    wd_cosine_scheduler and DINOHead on hash-generated inputs)."""
    import json, os
    import numpy as np
    from oracle import dino_oracle as D
    from tests.util import GOLDEN, sample_of
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))
    L = fx["loss"]
    student = _hu((L["V"] * L["B"], L["K"]), L["student_seed"], -3.0, 3.0).requires_grad_(True)
    teacher, center = _hu((2 * L["B"], L["K"]), L["teacher_seed"], -3.0, 3.0), _hu((1, L["K"]), L["center_seed"], -0.5, 0.5)
    loss = D.dino_loss(student, teacher, center, L["V"], L["student_temp"], L["teacher_temp"])
    loss.backward()
    assert abs(float(loss) - L["loss"]) < 1e-6 * abs(L["loss"])
    got, want, _, _ = sample_of(student.grad, L["dstudent"])
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-9)
    got, want, _, _ = sample_of(D.update_center(center, teacher, L["center_momentum"]), L["center_after"])
    assert torch.equal(got, want)
    assert np.array_equal(D.teacher_temp_schedule(0.04, 0.07, 3, 10), np.array(L["teacher_temp_schedule"]))
    E = fx["ema"]
    q = [_hu(tuple(s), sd, -1, 1) for s, sd in zip(E["shapes"], E["q_seeds"])]
    k = [_hu(tuple(s), sd, -1, 1) for s, sd in zip(E["shapes"], E["k_seeds"])]
    D.update_momentum_encoder(q, k, E["m"])
    for t, want in zip(k, E["k_after"]):
        assert torch.equal(t.flatten(), torch.tensor(want))
    for key in ("wd", "momentum"):
        S = fx["schedules"][key]
        assert np.array_equal(D.cosine_scheduler(S["base"], S["final"], S["epochs"], S["niter"]), np.array(S["values"]))
    S = fx["schedules"]["warm"]
    assert np.array_equal(D.cosine_scheduler(S["base"], S["final"], S["epochs"], S["niter"], S["warmup_epochs"], S["start"]), np.array(S["values"]))
    from headct_foundation_amd.dino import wd_cosine_scheduler
    assert np.array_equal(wd_cosine_scheduler(S["base"], S["final"], S["epochs"], S["niter"], S["warmup_epochs"], S["start"]), np.array(S["values"]))


def test_dino_head_with_batchnorm_oracle_vs_reference_fixture():
    """DINOHead(use_bn=True) -- the default of config.py:86 -- restated in oracle/dino_oracle.py against the outputs of the reference's
    own module (tests/golden/dino.json["head_bn"]): training-mode output, input gradient, every parameter gradient (the Linear
    biases in front of a BatchNorm have a mathematically zero gradient), running statistics after one forward, eval-mode output."""
    import json, os
    from oracle import dino_oracle as D
    from tests.util import GOLDEN, sample_of
    fx = json.load(open(os.path.join(GOLDEN, "dino.json")))["head_bn"]
    p = D.make_head_bn_params(fx)
    assert list(p.keys()) == fx["keys"]
    po = {n: (t.clone().requires_grad_(True) if (t.is_floating_point() and "running" not in n and not n.endswith("weight_g")) else t.clone()) for n, t in p.items()}
    x = _hu((fx["rows"], fx["in_dim"]), fx["x_seed"], -1, 1).requires_grad_(True)
    dy = _hu((fx["rows"], fx["out_dim"]), fx["dy_seed"], -1, 1)
    y = D.dino_head_forward(po, x, training=True)
    (y * dy).sum().backward()
    got, want, l2, l2w = sample_of(y, fx["y"])
    assert abs(l2 - l2w) < 1e-5 * l2w and torch.allclose(got, want, rtol=1e-4, atol=1e-6)
    got, want, l2, l2w = sample_of(x.grad, fx["dx"])
    assert abs(l2 - l2w) < 1e-4 * l2w
    for n, e in fx["grads"].items():
        got, want, l2, l2w = sample_of(po[n].grad, e)
        if n in ("mlp.0.bias", "mlp.3.bias"):
            assert float(po[n].grad.abs().max()) < 1e-5 and l2w < 1e-4, n
        else:
            assert abs(l2 - l2w) < 1e-4 * l2w, n
    for n, want in fx["running_after"].items():
        assert torch.allclose(po[n].flatten(), torch.tensor(want), rtol=1e-5, atol=1e-7), n
    ye = D.dino_head_forward({n: t.detach() for n, t in po.items()}, x.detach(), training=False)
    got, want, l2, l2w = sample_of(ye, fx["y_eval"])
    assert abs(l2 - l2w) < 1e-5 * l2w and torch.allclose(got, want, rtol=1e-4, atol=1e-6)


def test_gaussian_smoothing_restatement_properties():
    """RandGaussianSmoothd restatement (transforms.py:230-238; MONAI absent, so pinned by properties only): kernel length and mass
    per sigma, separability (three 1-D passes == one dense 3-D correlation with the outer-product kernel), zero padding, samples
    that did not fire untouched."""
    from oracle import mae_oracle as O
    for sg, taps in ((0.5, 5), (0.62, 5), (0.63, 7), (0.87, 7), (0.88, 9), (1.0, 9)):
        k = O.gaussian_kernel_1d(sg)
        assert k.numel() == taps and abs(float(k.sum()) - 1.0) < 1e-4 and torch.equal(k, k.flip(0)) and float(k.min()) >= 0
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 2, 8, 8, 8, generator=g)
    sigma = [[0.5, 0.8, 1.0], [0.7, 0.7, 0.7]]
    y = O.gaussian_smooth3d(x, sigma, [True, False])
    assert torch.equal(y[1], x[1])
    kx, ky, kz = (O.gaussian_kernel_1d(s) for s in sigma[0])
    dense = torch.einsum("i,j,k->ijk", kx, ky, kz)
    pad = [(n - 1) // 2 for n in dense.shape]
    want = torch.nn.functional.conv3d(x[:1].reshape(2, 1, 8, 8, 8), dense[None, None], padding=pad).reshape(2, 8, 8, 8)
    assert float((y[0] - want).abs().max()) < 1e-6
    # zero padding: the corner voxel of a constant volume keeps only the in-volume half of each 1-D kernel
    c = O.gaussian_smooth3d(torch.ones(1, 1, 12, 12, 12), [[1.0, 1.0, 1.0]])
    half = float(O.gaussian_kernel_1d(1.0)[4:].sum())
    assert abs(float(c[0, 0, 0, 0, 0]) - half ** 3) < 1e-6 and abs(float(c[0, 0, 6, 6, 6]) - 1.0) < 1e-4
