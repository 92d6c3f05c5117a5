"""GPU: the drop-in engine / entry point on the HIP path (BASELINE config #1 plumbing run: ViT-Tiny, 2-layer decoder,
64^3 synthetic volumes, patch 16, mask 0.75, batch 2), checkpoint layout and resume."""
import argparse
import logging
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_main_pretrain_mae_plumbing_run(cuda, tmp_path):
    """python -m torch.distributed.run --nproc-per-node 1 main_pretrain_mae.py ... (2 epochs, val, checkpoint, test)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "main_pretrain_mae.py"), "--local_rank", "0", "--model_name", "mae",
           "--batch_size", "2", "--max_epochs", "2", "--base_lr", "1.5e-4", "--cfg", os.path.join(ROOT, "configs/mae/mae_tiny_plumbing.yaml"),
           "--optimizer", "AdamW", "--scheduler", "cosine", "--weight_decay", "5e-3", "--grad_clip", "3.0",
           "--opts", "MODEL.DIR", str(tmp_path / "ckpt"), "LOG.OUTPUT_DIR", str(tmp_path / "log"), "OUTPUT", str(tmp_path / "json")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Train completed" in r.stdout and "Test completed" in r.stdout
    ck = torch.load(tmp_path / "ckpt" / "latest_mae_tiny.pt", map_location="cpu", weights_only=True)
    assert sorted(ck.keys()) == ["best_loss", "epoch", "momentum_model_state_dict", "optimizer", "scheduler", "state_dict"]
    assert ck["momentum_model_state_dict"] is None and ck["epoch"] == 1
    assert all(k.startswith("module.") for k in ck["state_dict"])  # saved from the DDP wrapper (misc.py:38)
    assert "module.blocks.0.attn.qkv.weight" in ck["state_dict"]
    assert (tmp_path / "ckpt" / "best_mae_tiny.pt").exists()
    st = ck["optimizer"]["state"]
    assert set(st[0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st[0]["step"]) == 8.0  # 4 it/epoch x 2


def test_async_loss_readback_logs_what_the_synchronous_loop_logs(lib, cuda, monkeypatch):
    """train_one_epoch reads the loss back one iteration late through a pinned buffer (no per-iteration device synchronisation);
    HCT_SYNC_LOSS=1 is the reference's synchronize + .item() per iteration (engine_pretrain_mae.py:73-74).  Same log lines in the
    same order, same epoch statistics (fp32 path: bit-identical runs)."""
    import engine_pretrain_mae as E
    from oracle import mae_oracle as O
    from headct_foundation_amd import MaskedAutoencoderViT
    from headct_foundation_amd.cfgnode import CfgNode
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    from headct_foundation_amd.optim import HipAdamW
    cfg = O.CONFIGS["micro"]
    ecfg = CfgNode()
    ecfg.MODEL = CfgNode(); ecfg.MODEL.NAME = "mae"
    ecfg.TRAIN = CfgNode(); ecfg.TRAIN.GRAD_CLIP = 3.0

    class Lines(logging.Handler):
        def __init__(self):
            super().__init__()
            self.lines = []

        def emit(self, record):
            self.lines.append(record.getMessage())

    def run(sync):
        monkeypatch.setenv("HCT_SYNC_LOSS", "1" if sync else "0")
        torch.manual_seed(0)
        m = MaskedAutoencoderViT(**cfg.ctor_kwargs(), compute_dtype="fp32")
        m.load_state_dict(O.make_params(cfg, 0))
        m = m.to(cuda)
        opt = HipAdamW(m, lr=1e-3, weight_decay=5e-3, betas=(0.9, 0.95))
        sch = get_cosine_schedule_with_warmup(opt, 2, 10, lr_end=1e-6)
        batches = [O.make_volume(cfg, 2, 30 + i) for i in range(5)]
        log = logging.getLogger(f"tap{int(sync)}")
        log.setLevel(logging.INFO); log.propagate = False
        h = Lines(); log.addHandler(h)
        torch.manual_seed(123)  # the masking noise is drawn inside forward
        stats = E.train_one_epoch(ecfg, m, batches, opt, sch, 0, 1, logger=log, device=cuda)
        return h.lines, stats

    a_lines, a_stats = run(False)
    s_lines, s_stats = run(True)
    assert len(a_lines) == 6 and a_lines == s_lines  # 5 iteration lines + "Averaged stats"
    assert a_stats == s_stats and set(a_stats) == {"loss", "lr"}


def test_resume_reproduces_uninterrupted_run(lib, cuda, tmp_path):
    """save_checkpoint -> new model/optimizer -> load -> continue == uninterrupted training (bit-exact fp32 path)."""
    from oracle import mae_oracle as O
    from headct_foundation_amd import MaskedAutoencoderViT
    from headct_foundation_amd.lr_sched import get_cosine_schedule_with_warmup
    from headct_foundation_amd.misc import load_optimizer, save_checkpoint
    from headct_foundation_amd.optim import HipAdamW, clip_gradients
    cfg = O.CONFIGS["micro"]

    def fresh():
        m = MaskedAutoencoderViT(**cfg.ctor_kwargs(), compute_dtype="fp32")
        m.load_state_dict(O.make_params(cfg, 0))
        m = m.to(cuda)
        opt = HipAdamW(m, lr=1e-3, weight_decay=5e-3, betas=(0.9, 0.95))
        sch = get_cosine_schedule_with_warmup(opt, 2, 10, lr_end=1e-6)
        return m, opt, sch

    def steps(m, opt, sch, idx):
        out = []
        for i in idx:
            opt.zero_grad()
            loss, _, _ = m(O.make_volume(cfg, 2, i).to(cuda), noise=O.make_noise(cfg, 2, i).to(cuda))
            loss.backward()
            clip_gradients(m, 3.0)
            opt.step(); sch.step()
            out.append(float(loss.detach()))
        return out

    m, opt, sch = fresh()
    ref = steps(m, opt, sch, range(6))
    m1, opt1, sch1 = fresh()
    first = steps(m1, opt1, sch1, range(3))
    save_checkpoint(m1, None, 0, opt1, sch1, filename="c.pt", best_loss=1.0, dir_add=str(tmp_path), logger=logging.getLogger("t"))
    ck = torch.load(tmp_path / "c.pt", map_location="cpu", weights_only=True)
    m2, opt2, sch2 = fresh()
    m2.load_state_dict(ck["state_dict"])
    load_optimizer(opt2, sch2, ck, logging.getLogger("t"))
    second = steps(m2, opt2, sch2, range(3, 6))
    assert first + second == ref
    for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n


def test_grad_accumulation_and_foreign_optimizer(lib, cuda):
    """Two backwards without zero_grad accumulate; torch.optim.AdamW on the HIP model's parameters also works."""
    from oracle import mae_oracle as O
    from headct_foundation_amd import MaskedAutoencoderViT
    cfg = O.CONFIGS["micro"]
    params = O.make_params(cfg, 0)
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs(), compute_dtype="fp32")
    m.load_state_dict(params)
    m = m.to(cuda)
    x, nz = O.make_volume(cfg, 2, 0).to(cuda), O.make_noise(cfg, 2, 0).to(cuda)
    m(x, noise=nz)[0].backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m(x, noise=nz)[0].backward()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, 2 * g1[n], rtol=1e-6, atol=1e-9), n
    # GradScaler-style scaled loss: gradients scale with the backward seed
    m.zero_grad()
    (m(x, noise=nz)[0] * 8.0).backward()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, 8 * g1[n], rtol=1e-5, atol=1e-8), n
    # foreign optimizer vs oracle AdamW step
    m.zero_grad()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=5e-3)
    loss = m(x, noise=nz)[0]
    loss.backward()
    opt.step()
    st = O.TrainState({k: v.clone() for k, v in params.items()})
    O.train_step(cfg, st, x.cpu(), nz.cpu(), base_lr=1e-3, min_lr=1e-9, warmup=0, total=10**9, weight_decay=5e-3)
    # Adam's first step is lr*sign(g): elements whose gradient is at round-off level may flip by 2*lr, so require
    # agreement on all but a vanishing fraction of elements instead of element-wise closeness.
    for n, p in m.named_parameters():
        if not n.endswith("qkv.bias"):
            bad = ((p.detach().cpu() - st.params[n]).abs() > 1e-5).float().mean().item()
            assert bad < 2e-3, (n, bad)
    loss2 = m(x, noise=nz)[0]  # bf16/f32 working copies refresh after a foreign update
    assert float(loss2) < float(loss)
