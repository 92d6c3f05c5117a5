"""CPU, world_size 2, gloo: the data-parallel wrapper's bucketing / all-reduce / broadcast logic on a stand-in flat-buffer
model that replays the native plan's stage ranges (the HIP model itself cannot run without a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeFlatModel(torch.nn.Module):
    """Mimics the HIP model's contract with DistributedDataParallel: flat buffers + staged backward hooks."""

    def __init__(self, total, ranges):
        super().__init__()
        self._flat = torch.arange(total, dtype=torch.float32) * 0.0
        self._flat_grad = torch.zeros(total)
        self._bucket_hook = None
        self._post_backward_hook = None
        self._grad_prescale = 1.0
        self.ranges = ranges
        self.marked = 0

    def mark_weights_updated(self, plain_bf16_fresh=False):
        self.marked += 1

    def backward(self, rank):
        for s, (b, e) in enumerate(self.ranges):  # stage s finishes [b, e) (descending ranges)
            self._flat_grad[b:e] = (torch.arange(b, e, dtype=torch.float32) + 1.0) * (rank + 1) * self._grad_prescale
            if self._bucket_hook:
                self._bucket_hook(s, b, e)
        if self._post_backward_hook:
            self._post_backward_hook()


def _worker(rank, world, port, total, ranges, cap_mb, out, grad_dtype="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from headct_foundation_amd.ddp import DistributedDataParallel
        m = _FakeFlatModel(total, ranges)
        m._flat += float(rank + 1)  # ranks start different; ctor must broadcast rank 0's parameters
        ddp = DistributedDataParallel(m, bucket_cap_mb=cap_mb, grad_dtype=grad_dtype)
        assert torch.all(m._flat == 1.0) and m.marked == 1
        assert "module._dummy" not in ddp.state_dict()
        for _ in range(2):
            m._flat_grad.zero_()
            m.backward(rank)
            want = (torch.arange(total, dtype=torch.float32) + 1.0) * sum(r + 1 for r in range(world)) / world
            covered = torch.zeros(total, dtype=torch.bool)
            for b, e in ranges:
                covered[b:e] = True
            # bf16 buckets: every rank's gradient and the sum are rounded to 8 significant bits
            assert torch.allclose(m._flat_grad[covered], want[covered], rtol=1e-5 if grad_dtype == "fp32" else 1.2e-2)
            assert m._flat_grad.dtype == torch.float32
            # launched buckets tile the stage ranges exactly, in descending order
            spans = sorted(ddp.launched)
            assert spans[0][0] == min(b for b, _ in ranges) and spans[-1][1] == max(e for _, e in ranges)
            assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        if rank == 0:
            out.put(len(ddp.launched))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("cap_mb,expect_buckets", [(1e-9, 5), (0.004, 3), (1000.0, 1)])
def test_ddp_bucketed_allreduce_world2(cap_mb, expect_buckets):
    total = 5 * 1024
    ranges = [(4096, 5120), (3072, 4096), (2048, 3072), (1024, 2048), (0, 1024)]  # like the plan: end -> start
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, ranges, cap_mb, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == expect_buckets


def test_ddp_bf16_gradient_buckets_world2():
    """grad_dtype="bf16" (298 MB instead of 595 MB per step over the links): buckets are cast to bfloat16, summed by the
    collective, cast back into the fp32 gradient; off by default."""
    total = 5 * 1024
    ranges = [(4096, 5120), (3072, 4096), (2048, 3072), (1024, 2048), (0, 1024)]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, ranges, 0.004, out, "bf16")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 3


def test_plan_stage_ranges_are_contiguous_and_descending(lib):
    """The native plan's backward stages finish contiguous parameter ranges from the end of the flat buffer to its start."""
    import ctypes as C
    from headct_foundation_amd import MaskedAutoencoderViT, _lib
    from oracle import mae_oracle as O
    cfg = O.CONFIGS["tiny"]
    m = MaskedAutoencoderViT(**cfg.ctor_kwargs())
    h = lib.hct_mae_plan_create(C.byref(m._ccfg), 2, m._dt)
    try:
        n = lib.hct_mae_num_backward_stages(h)
        assert n == cfg.encoder_depth + cfg.decoder_depth + 3
        prev_begin = lib.hct_mae_plan_param_elems(h)
        for s in range(n):
            b, e = C.c_int64(), C.c_int64()
            _lib.check(lib.hct_mae_backward_stage_range(h, s, C.byref(b), C.byref(e)), "range")
            assert e.value == prev_begin and b.value < e.value and b.value % 1024 == 0
            prev_begin = b.value
        assert prev_begin == 0
    finally:
        lib.hct_mae_plan_destroy(h)


class _FakeHead:
    """The DINO head's contract with DinoDataParallel: flat parameter / gradient buffers."""

    def __init__(self, n, rank):
        self._flat = torch.full((n,), float(rank + 1))
        self._flat_grad = torch.zeros(n)
        self.marked = 0

    def mark_weights_updated(self, plain_bf16_fresh=False):
        self.marked += 1


class _FakeMultiCrop(torch.nn.Module):
    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def forward(self, x):
        return x


def _dino_worker(rank, world, port, total, ranges, n_head):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from headct_foundation_amd.dino import DinoDataParallel
        backbone = _FakeFlatModel(total, ranges)
        backbone._flat += float(rank + 1)
        head = _FakeHead(n_head, rank)
        ddp = DinoDataParallel(_FakeMultiCrop(backbone, head), bucket_cap_mb=0.004)
        # rank 0's backbone and head parameters everywhere
        assert torch.all(backbone._flat == 1.0) and backbone.marked == 1
        assert torch.all(head._flat == 1.0) and head.marked == 1
        assert ddp.module.head is head and ddp(3) == 3
        for _ in range(2):  # one iteration of the engine: backbone buckets during its backward, the head's gradient right after
            backbone._flat_grad.zero_()
            backbone.backward(rank)
            head._flat_grad = (torch.arange(n_head, dtype=torch.float32) + 1.0) * (rank + 1)
            ddp.reduce_head_gradients()
            mean_factor = sum(r + 1 for r in range(world)) / world
            assert torch.allclose(backbone._flat_grad, (torch.arange(total, dtype=torch.float32) + 1.0) * mean_factor)
            assert torch.allclose(head._flat_grad, (torch.arange(n_head, dtype=torch.float32) + 1.0) * mean_factor)
    finally:
        dist.destroy_process_group()


def test_dino_data_parallel_world2():
    """DINO (config #5) over two ranks: backbone gradient in buckets through its staged backward, head gradient in one all-reduce,
    both averaged; rank 0's parameters broadcast at construction (engine_pretrain_dino.train_one_epoch's data-parallel calls)."""
    total = 4 * 1024
    ranges = [(3072, 4096), (2048, 3072), (1024, 2048), (0, 1024)]
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_dino_worker, args=(r, 2, port, total, ranges, 777)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
