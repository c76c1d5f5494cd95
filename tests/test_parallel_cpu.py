"""Data-parallel engine on CPU: world_size 2, gloo.  Checks that bucketed, overlapped gradient
all-reduce + AdamW equals one process stepping on the mean gradient, that accumulation windows
exchange once, and that the num_masks normaliser is summed over ranks (HF:781-794)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.c = nn.Linear(16, 4)
        self.unused = nn.Linear(3, 3)  # never receives a gradient: exercises the straggler path

    def forward(self, x):
        h = torch.relu(self.a(x))
        return self.c(torch.relu(self.b(h)) + self.b(h))  # self.b used twice


def _data(rank, step):
    g = torch.Generator().manual_seed(100 * step + rank)
    return torch.randn(5, 8, generator=g), torch.randn(5, 4, generator=g)


def _worker(rank, world, port, accumulation, outdir, exchange="all_reduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from weed_instance_segmentation_amd.parallel import DataParallelEngine
    torch.manual_seed(1234 + rank)  # replicas start DIFFERENT; the engine must broadcast rank 0's
    model = Tiny()
    eng = DataParallelEngine(model, lr=1e-2, accumulation=accumulation, bucket_bytes=600, exchange=exchange)  # several buckets
    assert len(eng.buckets.buckets) > 2
    stepped = []
    for step in range(4):
        x, y = _data(rank, step)
        loss = ((model(x) - y) ** 2).mean()
        stepped.append(eng.backward_and_step(loss))
    n, w = eng.reduce_num_masks(torch.tensor(float(3 + rank)))
    torch.save((rank, [p.detach().clone() for p in model.parameters()], stepped, float(n), w),
               os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _reference(world, accumulation):
    torch.manual_seed(1234)  # rank 0's initial weights
    model = Tiny()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    for step in range(4):
        for r in range(world):
            x, y = _data(r, step)
            (((model(x) - y) ** 2).mean() / accumulation / world).backward()
        if (step + 1) % accumulation == 0:
            opt.step()
            for p in model.parameters():
                p.grad.zero_()
    return [p.detach().clone() for p in model.parameters()]


def _run(accumulation, outdir, exchange="all_reduce"):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, accumulation, str(outdir), exchange)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(outdir, f"r{r}.pt")) for r in range(world)]
    ref = _reference(world, accumulation)
    for (rank, params, stepped, n, w) in res:
        assert stepped == [(s + 1) % accumulation == 0 for s in range(4)]
        assert n == 3 + 4 and w == 2
        for a, b in zip(params, ref):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)  # replicas stay bit-identical


def test_ddp_gloo_world2(tmp_path):
    _run(1, tmp_path)


def test_ddp_gloo_world2_accumulation2(tmp_path):
    _run(2, tmp_path)  # the reference's GRADIENT_ACCUMULATION = 2 (config.py:8)


def test_ddp_gloo_world2_reduce_scatter_form(tmp_path):
    """exchange="reduce_scatter": buckets padded to whole shards; reduce_scatter_tensor (in place, the rank's shard a view of
    the bucket) -> mean on the shard -> all_gather_into_tensor, the code RCCL runs -- including a bucket whose parameter got
    no gradient (Tiny.unused: launched from finish()) -- gives the same parameters as the all-reduce form."""
    _run(2, tmp_path, exchange="reduce_scatter")


def test_reduce_scatter_exchange_calls_the_two_collectives(tmp_path, monkeypatch):
    """Guard against the exchange silently taking another route: a one-rank gloo group, the collectives counted."""
    import torch.distributed as dist
    from weed_instance_segmentation_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        calls = []
        rs, ag = dist.reduce_scatter_tensor, dist.all_gather_into_tensor
        monkeypatch.setattr(dist, "reduce_scatter_tensor", lambda *a, **k: (calls.append("rs"), rs(*a, **k))[1])
        monkeypatch.setattr(dist, "all_gather_into_tensor", lambda *a, **k: (calls.append("ag"), ag(*a, **k))[1])
        flat = torch.arange(6, dtype=torch.float32)
        h = parallel._ScatterGather(flat, 1, None)
        h.wait()
        assert calls == ["rs", "ag"] and torch.equal(flat, torch.arange(6, dtype=torch.float32))
    finally:
        dist.destroy_process_group()
