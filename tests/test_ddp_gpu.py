"""GPU, world_size 2 on ONE card (gloo backend moving device tensors): the real HIP model under the data-parallel wrapper.

The 8-GPU RCCL run is the driver's; here the same code path (constructor broadcast, backward seeded with dLoss/world,
staged bucket all-reduce of the flat gradient buffer launched from the native backward's stage callbacks, CU reserve,
optimizer step on the wrapped module) runs with two ranks sharing cuda:0 and is checked against a single-process step over
the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dtype, cap_mb, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mae_oracle as O
        from tests.util import build_hip_model, grads_by_name, rel_err
        from headct_foundation_amd.ddp import DistributedDataParallel
        from headct_foundation_amd.optim import HipAdamW, clip_gradients
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = O.CONFIGS["tiny"]
        B = 2
        params = O.make_params(cfg, 0)
        # rank 1 starts from perturbed weights: the constructor must broadcast rank 0's
        start = {k: (v if rank == 0 else v + 0.01) for k, v in params.items()}
        model = build_hip_model(cfg, start, dev, dtype).train()
        ddp = DistributedDataParallel(model, device_ids=[dev], bucket_cap_mb=cap_mb)
        xs = [O.make_volume(cfg, B, 20 + r).to(dev) for r in range(world)]
        ns = [O.make_noise(cfg, B, 20 + r).to(dev) for r in range(world)]
        opt = HipAdamW(ddp, lr=1e-3, weight_decay=5e-3, betas=(0.9, 0.95))
        opt.zero_grad()
        loss, _, _ = ddp(xs[rank], noise=ns[rank])
        loss.backward()
        torch.cuda.synchronize()
        g = grads_by_name(model)
        nb = len(ddp.launched)
        # reference: one process, both batches at once (every sample masks the same number of patches, so the loss over
        # 2B samples is the mean of the ranks' losses and its gradient the mean of the ranks' gradients)
        single = build_hip_model(cfg, params, dev, dtype).train()
        ls, _, _ = single(torch.cat(xs), noise=torch.cat(ns))
        ls.backward()
        torch.cuda.synchronize()
        gs = grads_by_name(single)
        tol = 2e-4 if dtype == "fp32" else 2e-2
        worst = max((rel_err(g[k], gs[k]), k) for k in g if not k.endswith("qkv.bias"))
        assert worst[0] < tol, worst
        # gradient accumulation: a second backward without zero_grad() is reduced as well (every backward is, as in torch's
        # DDP) and lands on top of the already averaged first one -> identical on both ranks, equal to the single-process sum
        xs2 = [O.make_volume(cfg, B, 30 + r).to(dev) for r in range(world)]
        ns2 = [O.make_noise(cfg, B, 30 + r).to(dev) for r in range(world)]
        ddp(xs2[rank], noise=ns2[rank])[0].backward()
        single(torch.cat(xs2), noise=torch.cat(ns2))[0].backward()
        torch.cuda.synchronize()
        g2, gs2 = grads_by_name(model), grads_by_name(single)
        worst2 = max((rel_err(g2[k], gs2[k]), k) for k in g2 if not k.endswith("qkv.bias"))
        assert worst2[0] < tol, worst2
        both = [torch.empty_like(model._flat_grad) for _ in range(world)]
        dist.all_gather(both, model._flat_grad.detach().clone())
        assert torch.equal(both[0], both[1]), "accumulated gradients differ between ranks"
        # zero_grad(set_to_none=False) from a foreign optimizer keeps the .grad tensors: the next backward still reduces
        torch.optim.SGD(model.parameters(), lr=0.0).zero_grad(set_to_none=False)
        ddp(xs[rank], noise=ns[rank])[0].backward()
        torch.cuda.synchronize()
        g3 = grads_by_name(model)
        worst3 = max((rel_err(g3[k], gs[k]), k) for k in g3 if not k.endswith("qkv.bias"))
        assert worst3[0] < tol, worst3
        clip_gradients(ddp, 3.0)
        opt.step()
        torch.cuda.synchronize()
        flat = model._flat.detach().clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged after one step"
        assert all(k.startswith("module.") for k in ddp.state_dict())
        if rank == 0:
            out.put((nb, float(worst[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,cap_mb", [("fp32", 1e-9), ("bf16", 8.0)])
def test_two_ranks_one_card(cuda, dtype, cap_mb):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype, cap_mb, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    nb, worst = out.get(timeout=5)
    assert nb >= 1
    if cap_mb < 1e-6:
        assert nb > 3  # one bucket per backward stage with a zero cap


def test_bench_two_ranks_rehearsal(cuda):
    """The N > 1 path of bench.py itself (rendezvous, per-rank seeds, barrier + synchronize fences, MAX-reduce of the elapsed time, one
    JSON line from rank 0): `--gpus 2` launched the way the driver launches it, with HCT_BENCH_REHEARSAL=1 putting both ranks on this
    one card over gloo (the driver's runs use one rank per GPU over RCCL).  Small batch: only the code path is under test."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, HCT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    assert out["config"]["parallelism"] == "dp2" and out["value"] > 0 and abs(out["value"] - 16 * 1e3 / out["ms_per_step"]) < 0.01 * out["value"]
    assert "cpu_baseline" not in out and out["roofline"] is not None


@pytest.mark.gpu
def test_bench_rccl_world1(cuda):
    """bench.py at world size ONE over the real "nccl" backend (= RCCL) with the data-parallel wrapper forced on (HCT_BENCH_RCCL1=1):
    communicator set-up, the bucketed asynchronous all-reduce on RCCL's stream, the CU reserve and the final wait execute on
    hardware (the driver's N > 1 runs are the only other place they do).  The sums are those of one rank, so the loss curve must be
    the plain run's."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for tag, extra in (("plain", {}), ("rccl1", {"HCT_BENCH_RCCL1": "1", "MASTER_PORT": "29571"})):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                           capture_output=True, text=True, env=env, cwd=root, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert sum(lines["rccl1"]["rccl_buckets_per_step"]) > 0 and len(lines["rccl1"]["rccl_buckets_per_step"]) >= 2
    assert abs(lines["rccl1"]["loss_last"] - lines["plain"]["loss_last"]) < 1e-4 * abs(lines["plain"]["loss_last"]) + 1e-6
