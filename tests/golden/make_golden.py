"""Generate golden fixtures from the REFERENCE itself (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the reference's own src/models/mae.py, vit.py, classifier.py, attentionblock.py,
patch_embedding.py, pos_embed.py, misc.py, lr_sched.py and engine_pretrain_mae.py from /root/reference,
with stand-ins registered in sys.modules for the seven third-party symbols that are
absent from this image (timm to_2tuple/to_3tuple; MONAI Conv,
trunc_normal_, MLPBlock, ensure_tuple_rep, optional_import, look_up_option).
Nothing of the reference's text is stored: fixtures hold inputs-by-seed and outputs
(loss, sampled activations/gradients/parameters, LR values, loss curve, key manifest).

Inputs and weights are NOT stored: they come from oracle.mae_oracle.make_params /
make_volume / make_noise, a portable integer-hash stream, so the GPU box regenerates
them bit-exactly.
"""
import importlib.machinery
import json
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import mae_oracle as O  # noqa: E402


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install_standins():
    def _ntuple(n):
        def f(x):
            if isinstance(x, (list, tuple)):
                return tuple(x)
            return tuple([x] * n)
        return f

    timm = _mod("timm"); tm = _mod("timm.models"); tl = _mod("timm.models.layers")
    tl.to_2tuple, tl.to_3tuple = _ntuple(2), _ntuple(3)
    timm.models = tm; tm.layers = tl

    monai = _mod("monai"); mn = _mod("monai.networks"); ml = _mod("monai.networks.layers")
    mb = _mod("monai.networks.blocks"); mm = _mod("monai.networks.blocks.mlp")
    mu = _mod("monai.utils"); mum = _mod("monai.utils.module")

    class _Conv:
        CONV = "conv"
        def __getitem__(self, key):
            kind, dims = key
            assert kind == "conv" and dims == 3
            return nn.Conv3d
    ml.Conv = _Conv()

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        with torch.no_grad():
            return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)
    ml.trunc_normal_ = trunc_normal_

    class MLPBlock(nn.Module):  # MONAI 1.2/1.3 MLPBlock semantics
        def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
            super().__init__()
            self.linear1 = nn.Linear(hidden_size, mlp_dim)
            self.linear2 = nn.Linear(mlp_dim, hidden_size)
            self.fn = nn.GELU()
            self.drop1 = nn.Dropout(dropout_rate)
            self.drop2 = nn.Dropout(dropout_rate)
        def forward(self, x):
            return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))
    mm.MLPBlock = MLPBlock

    def ensure_tuple_rep(x, n):
        return tuple(x) if isinstance(x, (list, tuple)) else tuple([x] * n)
    def optional_import(module, name=""):
        try:
            m = __import__(module, fromlist=[name] if name else [])
            return (getattr(m, name) if name else m), True
        except Exception:
            return None, False
    def look_up_option(opt, supported):
        if opt not in supported:
            raise ValueError(f"unsupported option {opt}")
        return opt
    mu.ensure_tuple_rep = ensure_tuple_rep
    mu.optional_import = optional_import
    mum.look_up_option = look_up_option
    monai.networks = mn; mn.layers = ml; mn.blocks = mb; mb.mlp = mm; monai.utils = mu; mu.module = mum


def sample(t: torch.Tensor, n: int = 192):
    """Strided sample + norms of a tensor (keeps fixtures small)."""
    f = t.detach().double().flatten()
    idx = np.unique(np.linspace(0, f.numel() - 1, min(n, f.numel())).astype(np.int64))
    return dict(shape=list(t.shape), idx=idx.tolist(), val=f[idx].tolist(), l2=float(f.norm()), sum=float(f.sum()))


TRAIN_HP = dict(base_lr=1.5e-4 * 2 / 256 * 64, min_lr=1.5e-7, warmup=2, total=8, weight_decay=5e-3,
                beta1=0.9, beta2=0.95, grad_clip=3.0)


def build_reference_model(cfg: O.MAEConfig, params):
    from src.models.mae import MaskedAutoencoderViT
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    sd = m.state_dict()
    names = [n for n, _, _ in O.param_shapes(cfg)]
    assert list(sd.keys()) == names, (list(sd.keys())[:12], names[:12])
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(params[k].shape), k
    m.load_state_dict(params, strict=True)
    return m


def run_case(name: str, batch: int, seed: int, full: bool):
    cfg = O.CONFIGS[name]
    params = O.make_params(cfg, seed)
    x = O.make_volume(cfg, batch, seed)
    noise = O.make_noise(cfg, batch, seed)
    model = build_reference_model(cfg, params)
    model.train()

    # The reference draws noise with torch.rand(N, L) as the only RNG draw of forward (mae.py:206).
    # Inject our tie-free noise by patching torch.rand for the duration of the call.
    real_rand = torch.rand
    def fake_rand(*a, **k):
        assert tuple(a) == tuple(noise.shape), a
        return noise.clone()
    captured = {}
    hooks = []
    def cap(tag):
        def h(mod, inp, out):
            captured[tag] = (out[0] if isinstance(out, tuple) else out).detach()
        return h
    hooks.append(model.patch_embedding.register_forward_hook(cap("patch_embed")))
    for i, b in enumerate(model.blocks):
        hooks.append(b.register_forward_hook(cap(f"enc{i}.out")))
    for i, b in enumerate(model.decoder_blocks):
        hooks.append(b.register_forward_hook(cap(f"dec{i}.out")))
    hooks.append(model.norm.register_forward_hook(cap("latent")))
    hooks.append(model.decoder_pred.register_forward_hook(cap("pred_full")))
    torch.rand = fake_rand
    try:
        loss, _, _ = model(x)
    finally:
        torch.rand = real_rand
    loss.backward()
    for h in hooks:
        h.remove()
    pred = captured["pred_full"][:, 1:, :]
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    # ---- oracle vs reference (same inputs) ----
    o_loss, o_pred, o_mask, o_grads, o_inter = O.forward_backward(cfg, params, x, noise, want_inter=True)
    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    rep = {"loss": abs(float(o_loss) - float(loss.detach())) / abs(float(loss.detach())), "pred": rel(o_pred, pred)}
    for k in ("patch_embed", "latent"):
        rep[k] = rel(o_inter[k], captured[k])
    for k in captured:
        if k.endswith(".out"):
            rep[k] = rel(o_inter[k], captured[k])
    assert set(o_grads) == set(grads), set(o_grads) ^ set(grads)
    rep["grad_max"] = max(rel(o_grads[k], grads[k]) for k in grads)
    print(f"[{name}] oracle-vs-reference rel err:", {k: f"{v:.2e}" for k, v in rep.items() if not k.endswith('.out')},
          "blocks max", f"{max([v for k, v in rep.items() if k.endswith('.out')] or [0]):.2e}")
    assert max(rep.values()) < 2e-5, rep

    fx = dict(config=name, batch=batch, seed=seed, loss=float(loss),
              mask_sum=float(o_mask.sum()), oracle_vs_reference=rep,
              act={k: sample(v) for k, v in captured.items() if k != "pred_full"},
              pred=sample(pred),
              unpatchify_pred=sample(model.unpatchify(pred, x)),
              patchify_x=sample(model.patchify(x)),
              grads={k: sample(v, 96) for k, v in grads.items()})

    # ---- N-step training curve through the reference's own train_one_epoch ----
    import engine_pretrain_mae as E
    from src.utils.lr_sched import get_cosine_schedule_with_warmup
    hp = TRAIN_HP
    model = build_reference_model(cfg, params)
    opt = torch.optim.AdamW(model.parameters(), lr=hp["base_lr"], weight_decay=hp["weight_decay"],
                            betas=(hp["beta1"], hp["beta2"]))  # optimizers.py:354-360
    sched = get_cosine_schedule_with_warmup(opt, hp["warmup"], hp["total"], lr_end=hp["min_lr"])
    nsteps = 4
    batches = [O.make_volume(cfg, batch, seed + 10 + i) for i in range(nsteps)]
    noises = [O.make_noise(cfg, batch, seed + 10 + i) for i in range(nsteps)]
    it = iter(noises)
    def fake_rand2(*a, **k):
        return next(it).clone()
    class Cfg:  # the two attributes train_one_epoch reads (engine_pretrain_mae.py:47,66)
        class MODEL: NAME = "mae"
        class TRAIN: GRAD_CLIP = hp["grad_clip"]
    losses = []
    class L(logging.Logger):
        def info(self, msg, *a, **k):
            if "Loss:" in str(msg):
                losses.append(float(str(msg).split("Loss:")[1]))
    lrs = []
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None  # engine_pretrain_mae.py:73 (no GPU here)
    torch.rand = fake_rand2
    real_step = sched.step
    def step_spy(*a, **k):
        lrs.append(opt.param_groups[0]["lr"])
        return real_step(*a, **k)
    sched.step = step_spy
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            stats = E.train_one_epoch(Cfg, model, batches, opt, sched, 0, 1, logger=L("g"), device=torch.device("cpu"),
                                      use_amp=False, scaler=torch.amp.GradScaler(enabled=False), wandb_run=None)
    finally:
        torch.rand = real_rand
        torch.cuda.synchronize = real_sync
    # oracle curve
    st = O.TrainState({k: v.clone() for k, v in params.items()})
    o_losses, o_lrs = [], []
    for i in range(nsteps):
        l, lr, _, _ = O.train_step(cfg, st, batches[i], noises[i], **hp)
        o_losses.append(l); o_lrs.append(lr)
    ref_params = dict(model.named_parameters())
    perrs = {k: rel(st.params[k], ref_params[k].detach()) for k in ref_params}
    perr = max(perrs.values())
    print("   worst params:", sorted(perrs.items(), key=lambda kv: -kv[1])[:3])
    print(f"[{name}] train curve ref {losses} oracle {[round(v, 4) for v in o_losses]} lr {lrs} param-relerr {perr:.2e}")
    assert np.allclose(losses, o_losses, atol=6e-5), (losses, o_losses)
    assert np.allclose(lrs, o_lrs, rtol=1e-12), (lrs, o_lrs)
    # the K-third of attn.qkv.bias has a mathematically ZERO gradient (softmax is invariant to a key bias);
    # Adam normalises its round-off noise to +-lr, so that slice is compared loosely everywhere.
    assert perr < 5e-5
    fx["train"] = dict(hp=hp, steps=nsteps, logged_losses=losses, lrs=lrs, avg_loss=stats["loss"],
                       params_after={k: sample(v, 64) for k, v in ref_params.items()},
                       opt_state_keys=sorted(opt.state_dict().keys()),
                       sched_state_keys=sorted(sched.state_dict().keys()))
    fx["state_dict_manifest"] = [[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()]

    os.makedirs(HERE, exist_ok=True)
    with open(os.path.join(HERE, f"{name}_b{batch}_s{seed}.json"), "w") as f:
        json.dump(fx, f)
    if full:  # whole tensors for the micro case (small)
        np.savez_compressed(os.path.join(HERE, f"{name}_b{batch}_s{seed}_full.npz"),
                            loss=np.float32(float(loss)), pred=pred.numpy(), latent=captured["latent"].numpy(),
                            **{"grad." + k: v.numpy() for k, v in grads.items()})


def lr_schedule_fixture():
    """Values of the reference's LambdaLR for a canonical schedule (lr_sched.py:18-55)."""
    from src.utils.lr_sched import get_cosine_schedule_with_warmup
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1.5e-4)
    sched = get_cosine_schedule_with_warmup(opt, 5, 100, lr_end=1.5e-7)
    vals = []
    for _ in range(110):
        vals.append(opt.param_groups[0]["lr"])
        opt.step(); sched.step()
    with open(os.path.join(HERE, "lr_schedule.json"), "w") as f:
        json.dump(dict(base_lr=1.5e-4, min_lr=1.5e-7, warmup=5, total=100, lrs=vals), f)


def sincos_fixture():
    from src.utils.pos_embed import build_sincos_position_embedding
    out = {}
    for g, d in ((4, 48), (6, 768), (4, 192)):
        ref = build_sincos_position_embedding([g, g, g], d, 3).detach()
        mine = O.build_sincos_position_embedding_3d(g, d)
        assert torch.equal(ref, mine), (g, d, float((ref - mine).abs().max()))
        out[f"{g}_{d}"] = sample(ref, 256)
    with open(os.path.join(HERE, "sincos.json"), "w") as f:
        json.dump(out, f)


def vit_fixture():
    """ViT.forward (feature extraction, src/models/vit.py:144-173) with register tokens and a learnable position table."""
    from src.models.vit import ViT
    torch.manual_seed(11)
    kw = dict(in_chans=1, img_size=(32, 32, 32), patch_size=(16, 16, 16), hidden_size=192, mlp_dim=384, num_layers=2, num_heads=3,
              patch_embed="conv", pos_embed="learnable", classification=False, num_register_tokens=2, qkv_bias=False)
    ref = ViT(**kw).eval()
    params = O.make_vit_params({k: list(v.shape) for k, v in ref.state_dict().items()})  # cls / register tokens start as zeros
    ref.load_state_dict(params, strict=True)
    x = torch.from_numpy(O.hash_uniform(2 * 32 ** 3, 7).reshape(2, 1, 32, 32, 32).astype(np.float32)) * 0.5 + 0.5
    with torch.no_grad():
        out, hidden = ref(x)
    o_out, o_hidden = O.vit_forward(params, x, 16, 3, 2)
    err = max(float((out - o_out).abs().max()), *(float((a - b).abs().max()) for a, b in zip(hidden, o_hidden)))
    assert out.shape == (2, 1 + 2 + 8, 192) and err < 5e-6, err
    fx = dict(ctor=dict(kw, img_size=32, patch_size=16), max_abs_dev_oracle=err,
              state_dict={k: dict(shape=list(v.shape)) for k, v in ref.state_dict().items()},  # the reference's key order
              out=sample(out, 512), hidden=[sample(h, 256) for h in hidden])
    # the same model on a 48^3 volume: PatchEmbeddingBlock.forward resizes the position table (2^3 -> 3^3) for the call
    x48 = torch.from_numpy(O.hash_uniform(2 * 48 ** 3, 9).reshape(2, 1, 48, 48, 48).astype(np.float32)) * 0.5 + 0.5
    with torch.no_grad():
        out48, hidden48 = ref(x48)
    o_out48, o_hidden48 = O.vit_forward(params, x48, 16, 3, 2)
    err48 = max(float((out48 - o_out48).abs().max()), *(float((a - b).abs().max()) for a, b in zip(hidden48, o_hidden48)))
    assert out48.shape == (2, 1 + 2 + 27, 192) and err48 < 5e-6, err48
    fx["resized_48"] = dict(x_seed=9, max_abs_dev_oracle=err48, out=sample(out48, 512), hidden=[sample(h, 256) for h in hidden48])
    with open(os.path.join(HERE, "vit_features.json"), "w") as f:
        json.dump(fx, f)


def classifier_fixture():
    """Eval-mode LinearClassifier / AttentionClassifier (src/models/classifier.py) and ViT(classification=True); full outputs."""
    from src.models.classifier import AttentionClassifier, LinearClassifier
    from src.models.vit import ViT
    fx = {}
    # LinearClassifier
    ref = LinearClassifier(192, 3).eval()
    params = O.make_vit_params({k: list(v.shape) for k, v in ref.state_dict().items()}, seed0=300)
    ref.load_state_dict(params, strict=True)
    x = torch.from_numpy(O.hash_uniform(5 * 192, 31).reshape(5, 192).astype(np.float32))
    with torch.no_grad():
        out = ref(x)
    err = float((out - O.linear_classifier_forward(params, x)).abs().max())
    assert out.shape == (5, 3) and err < 2e-6, err
    fx["linear"] = dict(ctor=dict(dim=192, num_classes=3), seed0=300, x_seed=31, x_shape=[5, 192], max_abs_dev_oracle=err,
                        state_dict={k: dict(shape=list(v.shape)) for k, v in ref.state_dict().items()}, out=out.flatten().tolist())
    # the same head in training mode: one linear-probing step (batch statistics, CrossEntropyLoss, backward)
    ref = LinearClassifier(192, 3).train()
    ref.load_state_dict(params, strict=True)
    target = torch.tensor([0, 2, 1, 1, 0])
    logits = ref(x)
    loss = torch.nn.CrossEntropyLoss()(logits, target)
    loss.backward()
    o_logits, o_loss, o_grads, o_stats = O.linear_probe_step(params, x, target)
    sd = ref.state_dict()
    err = max(float((logits.detach() - o_logits).abs().max()), float((loss.detach() - o_loss).abs()),
              float((ref.linear.weight.grad - o_grads["linear.weight"]).abs().max()),
              float((ref.linear.bias.grad - o_grads["linear.bias"]).abs().max()),
              float((sd["bn.running_mean"] - o_stats["bn.running_mean"]).abs().max()),
              float((sd["bn.running_var"] - o_stats["bn.running_var"]).abs().max()))
    assert err < 2e-6 and int(sd["bn.num_batches_tracked"]) == 1, err
    fx["linear_probe_step"] = dict(ctor=dict(dim=192, num_classes=3), seed0=300, x_seed=31, x_shape=[5, 192], target=target.tolist(),
                                   max_abs_dev_oracle=err, state_dict=fx["linear"]["state_dict"],
                                   logits=logits.detach().flatten().tolist(), loss=float(loss.detach()),
                                   grad_weight=ref.linear.weight.grad.flatten().tolist(), grad_bias=ref.linear.bias.grad.tolist(),
                                   running_mean=sd["bn.running_mean"].tolist(), running_var=sd["bn.running_var"].tolist())
    # AttentionClassifier, one and several learnt queries
    for name, nq, bias, seed0 in (("attention_q1", 1, False, 320), ("attention_q3", 3, True, 340)):
        kw = dict(dim=192, num_classes=4, num_heads=3, qkv_bias=bias, num_queries=nq)
        ref = AttentionClassifier(**kw).eval()
        params = O.make_vit_params({k: list(v.shape) for k, v in ref.state_dict().items()}, seed0=seed0)
        params["cls_token"] = params["cls_token"] * 50.0  # logits of order one, so that the softmax is not flat
        ref.load_state_dict(params, strict=True)
        x = torch.from_numpy(O.hash_uniform(2 * 11 * 192, 33).reshape(2, 11, 192).astype(np.float32)) * 2.0
        with torch.no_grad():
            out = ref(x)
        err = float((out - O.attention_classifier_forward(params, x, 3, nq)).abs().max())
        assert out.shape == (2, 4) and err < 2e-6, (name, err)
        fx[name] = dict(ctor=kw, seed0=seed0, cls_token_gain=50.0, x_seed=33, x_shape=[2, 11, 192], x_gain=2.0, max_abs_dev_oracle=err,
                        state_dict={k: dict(shape=list(v.shape)) for k, v in ref.state_dict().items()}, out=out.flatten().tolist())
    # ViT with its own classification head (Linear + Tanh, and plain Linear)
    for name, post in (("vit_tanh", "Tanh"), ("vit_linear", "none")):
        kw = dict(in_chans=1, img_size=(32, 32, 32), patch_size=(16, 16, 16), hidden_size=192, mlp_dim=384, num_layers=2, num_heads=3,
                  patch_embed="conv", pos_embed="learnable", classification=True, num_classes=3, post_activation=post,
                  num_register_tokens=1, qkv_bias=True)
        ref = ViT(**kw).eval()
        params = O.make_vit_params({k: list(v.shape) for k, v in ref.state_dict().items()}, seed0=360)
        for k in params:
            if k.startswith("classification_head") and k.endswith("weight"):
                params[k] = params[k] * 10.0  # class scores of order one
        ref.load_state_dict(params, strict=True)
        x = torch.from_numpy(O.hash_uniform(2 * 32 ** 3, 7).reshape(2, 1, 32, 32, 32).astype(np.float32)) * 0.5 + 0.5
        with torch.no_grad():
            out, hidden = ref(x)
        o_out, _ = O.vit_forward(params, x, 16, 3, 2)
        err = float((out - o_out).abs().max())
        assert out.shape == (2, 3) and err < 5e-6, (name, err)
        fx[name] = dict(ctor=dict(kw, img_size=32, patch_size=16), seed0=360, head_weight_gain=10.0, max_abs_dev_oracle=err,
                        state_dict={k: dict(shape=list(v.shape)) for k, v in ref.state_dict().items()}, out=out.flatten().tolist())
    with open(os.path.join(HERE, "classifier_heads.json"), "w") as f:
        json.dump(fx, f)


def pos_interp_fixture():
    """interpolate_pos_embed (pos_embed.py:102-153) on a learnable table, up- and down-sampling; full outputs (small)."""
    from src.utils.pos_embed import interpolate_pos_embed
    out = {}
    for g_old, g_new, d, seed in ((4, 6, 12, 3), (6, 4, 12, 4), (2, 5, 6, 5), (6, 8, 24, 6)):
        table = torch.from_numpy(O.hash_uniform(g_old ** 3 * d, seed).reshape(1, g_old ** 3, d).astype(np.float32))
        fake = types.SimpleNamespace(patch_embedding=types.SimpleNamespace(
            n_patches=g_new ** 3, position_embeddings=torch.zeros(1, g_new ** 3, d)))
        ckpt = {"patch_embedding.position_embeddings": table.clone()}
        interpolate_pos_embed(fake, ckpt, spatial_dims=3)
        ref = ckpt["patch_embedding.position_embeddings"]
        mine = O.interpolate_pos_embed_3d(table, g_new, 0)
        err = float((ref - mine).abs().max())
        assert ref.shape == mine.shape and err < 2e-6, (g_old, g_new, err)
        out[f"{g_old}_{g_new}_{d}_{seed}"] = dict(max_abs_dev_oracle=err, ref=[float(v) for v in ref.flatten().tolist()])
    with open(os.path.join(HERE, "pos_interp.json"), "w") as f:
        json.dump(out, f)


def dino_fixture():
    """DINOLoss (src/losses/losses.py:46-102), _update_momentum_encoder (src/utils/misc.py:386-397), wd_cosine_scheduler
    (src/utils/wd_sched.py:3-15) and DINOHead (src/models/dino_head.py:7-41) of the reference on hash-generated inputs; the
    oracle restatement (oracle/dino_oracle.py) is asserted equal here, the reference's outputs are the fixture."""
    import torch.distributed as dist
    from oracle import dino_oracle as D
    from src.losses.losses import DINOLoss
    from src.models.dino_head import DINOHead
    from src.utils.misc import _update_momentum_encoder
    from src.utils.wd_sched import wd_cosine_scheduler
    out = {}
    V, B, K = 4, 3, 512
    u = lambda shape, seed, lo, hi: torch.from_numpy(O.hash_uniform(int(np.prod(shape)), seed)).float().reshape(shape) * (hi - lo) + lo
    student = u((V * B, K), 501, -3.0, 3.0).requires_grad_(True)
    teacher = u((2 * B, K), 502, -3.0, 3.0)
    center0 = u((1, K), 503, -0.5, 0.5)
    crit = DINOLoss(K, V, 0.04, 0.07, 3, 10, student_temp=0.1, center_momentum=0.9)
    crit.center.copy_(center0)
    real_ar, real_ws = dist.all_reduce, dist.get_world_size
    dist.all_reduce, dist.get_world_size = (lambda t, *a, **k: None), (lambda *a, **k: 1)
    try:
        loss = crit(student, teacher, 1)  # epoch 1 of the warm-up: teacher temperature 0.055
    finally:
        dist.all_reduce, dist.get_world_size = real_ar, real_ws
    loss.backward()
    temp = float(crit.teacher_temp_schedule[1])
    o_loss = D.dino_loss(student.detach(), teacher, center0, V, 0.1, temp)
    assert abs(float(o_loss) - float(loss)) < 1e-6 * abs(float(loss)), (float(o_loss), float(loss))
    assert torch.allclose(D.update_center(center0, teacher, 0.9), crit.center, rtol=0, atol=0)
    assert np.array_equal(D.teacher_temp_schedule(0.04, 0.07, 3, 10), crit.teacher_temp_schedule)
    out["loss"] = dict(V=V, B=B, K=K, student_seed=501, teacher_seed=502, center_seed=503, student_temp=0.1, teacher_temp=temp,
                       center_momentum=0.9, loss=float(loss), dstudent=sample(student.grad, 256), center_after=sample(crit.center, 128),
                       teacher_temp_schedule=crit.teacher_temp_schedule.tolist())
    # momentum teacher
    q = [u((5, 7), 510, -1, 1), u((12,), 511, -1, 1)]
    k = [u((5, 7), 512, -1, 1), u((12,), 513, -1, 1)]
    mq, mk = nn.ParameterList([nn.Parameter(t.clone()) for t in q]), nn.ParameterList([nn.Parameter(t.clone()) for t in k])
    _update_momentum_encoder(mq, mk, 0.996)
    ko = [t.clone() for t in k]
    D.update_momentum_encoder(q, ko, 0.996)
    assert all(torch.equal(a, b.data) for a, b in zip(ko, mk))
    out["ema"] = dict(m=0.996, q_seeds=[510, 511], k_seeds=[512, 513], shapes=[[5, 7], [12]], k_after=[t.data.flatten().tolist() for t in mk])
    # schedules
    wd = wd_cosine_scheduler(0.04, 0.4, 5, 7)
    mom = wd_cosine_scheduler(0.996, 1.0, 5, 7)
    assert np.array_equal(D.cosine_scheduler(0.04, 0.4, 5, 7), wd)
    out["schedules"] = dict(wd=dict(base=0.04, final=0.4, epochs=5, niter=7, values=wd.tolist()),
                            momentum=dict(base=0.996, final=1.0, epochs=5, niter=7, values=mom.tolist()),
                            warm=dict(base=1.0, final=0.1, epochs=4, niter=3, warmup_epochs=1, start=0.2,
                                      values=wd_cosine_scheduler(1.0, 0.1, 4, 3, 1, 0.2).tolist()))
    # projection head: forward + backward through every parameter (weight_g frozen: norm_last_layer)
    head = DINOHead(48, 128, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=64, bottleneck_dim=32)
    hp = {}
    for i, (n, prm) in enumerate(head.named_parameters()):
        lo, hi = (-0.2, 0.2) if prm.dim() > 1 else (-0.05, 0.05)
        if n.endswith("weight_g"):
            continue  # filled with 1 by the constructor (dino_head.py:27)
        prm.data.copy_(u(tuple(prm.shape), 520 + i, lo, hi))
    for n, prm in head.named_parameters():
        hp[n] = prm.detach().clone()
    xh = u((6, 48), 540, -1, 1).requires_grad_(True)
    dy = u((6, 128), 541, -1, 1)
    y = head(xh)
    (y * dy).sum().backward()
    xo = xh.detach().clone().requires_grad_(True)
    po = {n: t.clone().requires_grad_(not n.endswith("weight_g")) for n, t in hp.items()}
    yo = D.dino_head_forward(po, xo)
    (yo * dy).sum().backward()
    assert torch.allclose(yo, y, rtol=1e-6, atol=1e-7)
    assert torch.allclose(xo.grad, xh.grad, rtol=1e-5, atol=1e-7)
    out["head"] = dict(in_dim=48, out_dim=128, hidden=64, bottleneck=32, names=[n for n, _ in head.named_parameters()],
                       shapes={n: list(t.shape) for n, t in hp.items()}, param_seed0=520, x_seed=540, dy_seed=541,
                       requires_grad={n: bool(prm.requires_grad) for n, prm in head.named_parameters()},
                       y=sample(y, 256), dx=sample(xh.grad, 256),
                       grads={n: sample(prm.grad, 128) for n, prm in head.named_parameters() if prm.grad is not None})
    # the same head with use_bn=True (the default of config.py:86), training mode: batch statistics, running statistics after one
    # forward, gradients of every parameter incl. the BatchNorm affine pairs; then the eval-mode output with those running statistics
    hb = DINOHead(48, 128, use_bn=True, norm_last_layer=True, nlayers=3, hidden_dim=64, bottleneck_dim=32)
    for i, (n, prm) in enumerate(hb.named_parameters()):
        if n.endswith("weight_g"):
            continue
        lo, hi = (-0.2, 0.2) if prm.dim() > 1 else ((0.5, 1.5) if n in ("mlp.1.weight", "mlp.4.weight") else (-0.05, 0.05))
        prm.data.copy_(u(tuple(prm.shape), 560 + i, lo, hi))
    hbp = {n: t.detach().clone() for n, t in hb.state_dict().items()}
    xb = u((10, 48), 590, -1, 1).requires_grad_(True)
    dyb = u((10, 128), 591, -1, 1)
    hb.train()
    yb = hb(xb)
    (yb * dyb).sum().backward()
    after = {n: t.detach().clone() for n, t in hb.state_dict().items()}
    po = {n: (t.clone().float().requires_grad_(True) if (t.is_floating_point() and "running" not in n and not n.endswith("weight_g")) else t.clone()) for n, t in hbp.items()}
    xo = xb.detach().clone().requires_grad_(True)
    yo = D.dino_head_forward(po, xo, training=True)
    (yo * dyb).sum().backward()
    assert torch.allclose(yo, yb, rtol=1e-5, atol=1e-6), float((yo - yb).abs().max())
    assert float((xo.grad - xb.grad).norm() / xb.grad.norm()) < 2e-5, float((xo.grad - xb.grad).norm() / xb.grad.norm())
    for n, prm in hb.named_parameters():
        if prm.grad is None:
            continue
        if n in ("mlp.0.bias", "mlp.3.bias"):  # a bias in front of a BatchNorm has a mathematically zero gradient (the mean is removed): round-off only
            assert float(prm.grad.abs().max()) < 1e-5 and float(po[n].grad.abs().max()) < 1e-5, n
        else:
            assert float((po[n].grad - prm.grad).norm() / (prm.grad.norm() + 1e-30)) < 2e-5, n
    for n in ("mlp.1.running_mean", "mlp.1.running_var", "mlp.4.running_mean", "mlp.4.running_var"):
        assert torch.allclose(po[n], after[n], rtol=1e-5, atol=1e-7), n
    hb.eval()
    with torch.no_grad():
        ye = hb(xb.detach())
    ye_o = D.dino_head_forward({n: t.detach() for n, t in po.items()}, xb.detach(), training=False)
    assert torch.allclose(ye_o, ye, rtol=1e-5, atol=1e-6)
    out["head_bn"] = dict(in_dim=48, out_dim=128, hidden=64, bottleneck=32, keys=list(hbp.keys()), shapes={n: list(t.shape) for n, t in hbp.items()},
                          param_seed0=560, bn_weight_range=[0.5, 1.5], x_seed=590, dy_seed=591, rows=10,
                          y=sample(yb, 256), dx=sample(xb.grad, 256), y_eval=sample(ye, 256),
                          grads={n: sample(prm.grad, 128) for n, prm in hb.named_parameters() if prm.grad is not None},
                          running_after={n: after[n].flatten().tolist() for n in after if "running" in n},
                          num_batches_tracked=int(after["mlp.1.num_batches_tracked"]))
    # one whole DINO iteration with the reference's modules (engine_pretrain_dino.py:59-104: teacher on the two global crops,
    # student on all crops through MultiCropWrapper, DINOLoss, backward, last-layer gradients cancelled, centre update, momentum
    # teacher update), CPU, no AMP.  Four crops of 24^3 x 3 channels, patch 12, 4 register tokens, qkv bias, sincos table --
    # the structure of configs/dino/dino_HeadCT.yaml at toy width.
    from src.models.vit import ViT
    from src.utils.misc import MultiCropWrapper, cancel_gradients_last_layer
    kw = dict(in_chans=3, img_size=24, patch_size=12, hidden_size=48, mlp_dim=96, num_layers=2, num_heads=3, patch_embed="conv",
              pos_embed="sincos", classification=False, num_register_tokens=4, qkv_bias=True)
    hk = dict(in_dim=48, out_dim=128, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=64, bottleneck_dim=32)
    def build(seed_b, seed_h):
        b, h = ViT(**kw), DINOHead(**hk)
        pb = O.make_vit_params({k: list(v.shape) for k, v in b.state_dict().items()}, seed0=seed_b)
        b.load_state_dict(pb, strict=True)
        ph = D.make_head_params(48, 128, 64, 32, seed_h)
        assert list(h.state_dict().keys()) == list(ph.keys())
        h.load_state_dict(ph, strict=True)
        return MultiCropWrapper(b, h), pb, ph
    student, sb, sh = build(700, 750)
    teacher, tb, th = build(800, 850)
    Bc, V = 2, 4
    crops = [u((Bc, 3, 24, 24, 24), 900 + i, 0.0, 1.0) for i in range(V)]
    crit = DINOLoss(128, V, 0.04, 0.07, 3, 10)
    crit.center.copy_(u((1, 128), 910, -0.2, 0.2))
    center0 = crit.center.clone()
    student.train(); teacher.train()
    dist.all_reduce, dist.get_world_size = (lambda t, *a, **k: None), (lambda *a, **k: 1)
    try:
        with torch.no_grad():
            t_out = teacher(crops[:2])['dino_output']
        s_out = student(crops)['dino_output']
        loss = crit(s_out, t_out, 0)
    finally:
        dist.all_reduce, dist.get_world_size = real_ar, real_ws
    loss.backward()
    cancel_gradients_last_layer(0, student, 1)
    o_loss, o_gb, o_gh, o_center = D.dino_step(sb, sh, tb, th, crops, center0, patch=12, heads=3, layers=2, student_temp=0.1,
                                               teacher_temp=float(crit.teacher_temp_schedule[0]))
    assert abs(float(o_loss) - float(loss)) < 2e-6 * abs(float(loss)), (float(o_loss), float(loss))
    ref_g = {n: p.grad for n, p in student.named_parameters() if p.grad is not None}
    for n, g_ in ref_g.items():
        o = o_gb[n[len("backbone."):]] if n.startswith("backbone.") else o_gh[n[len("head."):]]
        assert float((o - g_).norm() / (g_.norm() + 1e-30)) < 2e-5, n
    assert not any("last_layer" in n for n in ref_g) and not any("last_layer" in n for n in o_gh)
    assert torch.allclose(o_center, crit.center, rtol=1e-4, atol=1e-6)  # the oracle's teacher logits differ in the last bits
    _update_momentum_encoder(student, teacher, 0.996)
    out["step"] = dict(vit=kw, head=hk, batch=Bc, crops=V, backbone_seed=[700, 800], head_seed=[750, 850], crop_seed0=900, center_seed=910,
                       teacher_temp=float(crit.teacher_temp_schedule[0]), loss=float(loss), logits=sample(s_out, 256),
                       grads={n: sample(g_, 96) for n, g_ in ref_g.items()}, center_after=sample(crit.center, 128),
                       teacher_after={n: sample(p, 32) for n, p in list(teacher.named_parameters())[:6]},
                       backbone_keys=[k for k in student.backbone.state_dict().keys()])
    with open(os.path.join(HERE, "dino.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference not mounted: fixtures can only be regenerated in the build container"
    sys.dont_write_bytecode = True
    install_standins()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "dino" in sys.argv[1:]:  # python tests/golden/make_golden.py dino : only tests/golden/dino.json
        dino_fixture()
        print("dino fixture written")
        sys.exit(0)
    sincos_fixture()
    lr_schedule_fixture()
    pos_interp_fixture()
    vit_fixture()
    classifier_fixture()
    dino_fixture()
    run_case("micro", 2, 0, full=True)
    run_case("yaml_cut", 2, 1, full=False)
    run_case("tiny", 2, 0, full=False)
    run_case("vitb_cut", 2, 0, full=False)
    print("golden fixtures written to", HERE)
