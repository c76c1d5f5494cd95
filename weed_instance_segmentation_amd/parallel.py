"""Batch-sharded data parallelism: one process per MI355X, `torch.distributed` over RCCL / xGMI.

The reference is single-process (models/mask2former/train.py:187-205: AdamW lr 5e-5, gradient
accumulation 2, loss / 2).  This engine keeps that step arithmetic per rank and adds the one exchange
the path needs when the image batch is sharded over GPUs:

  * gradients live in a few large FLAT buckets (default 64 MiB; parameters' .grad are views into
    them), filled in reverse parameter order.  A bucket is all-reduced (SUM) asynchronously the
    moment its last gradient has been accumulated, so the exchange overlaps the rest of backward.
    Few, large messages: xGMI is point-to-point (7 links x ~153 GB/s per GPU) and ring collectives
    are per-link bound, so per-message latency is what small buckets would waste;
  * the loss normaliser `num_masks` is all-reduced like the dependency does under a distributed
    launch (HF:781-794): one scalar, before the per-level losses.

Two forms of the bucket exchange (`exchange=`):
  "all_reduce"      one `all_reduce(SUM)` per bucket (default; RCCL picks ring / direct by size);
  "reduce_scatter"  `reduce_scatter_tensor` + `all_gather_into_tensor` per bucket, both IN PLACE on the flat bucket (the
                    rank's shard is a view of it) -- the two halves of an all-reduce as separate collectives, each moving
                    S/N per peer pair over all 7 xGMI links at once; the mean is taken on the 1/N shard between
                    the two.  The same code runs under gloo (tests/test_parallel_cpu.py executes exactly these two
                    collectives on CPU tensors) and RCCL; on RCCL / xGMI it is UNMEASURED (no multi-GPU box in the build
                    loop) -- experimental until the driver's scaling run has covered it.

Works with any backend (`nccl` == RCCL on ROCm; `gloo` for the tests: CPU tensors directly, GPU tensors through a host
copy of the bucket -- the same collectives on that copy -- so that two ranks can share ONE GPU in
tests/test_parallel_gpu.py; RCCL refuses two ranks on a device).
"""
from __future__ import annotations

from typing import Callable, Iterable

import torch
import torch.distributed as dist
from torch import nn


class _Done:
    def wait(self):
        return True


def _host_staged(t: torch.Tensor, group) -> bool:
    """A GPU tensor on a gloo group goes through a host copy (tests only); CPU tensors and RCCL take the collective directly."""
    return t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of `t` in place.  gloo + a GPU tensor: staged through host memory (tests only)."""
    if _host_staged(t, group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return w if async_op else _Done()


def broadcast_from(t: torch.Tensor, src: int = 0, group=None):
    if _host_staged(t, group):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
        return
    dist.broadcast(t, src=src, group=group)


class _ScatterGather:
    """reduce_scatter_tensor -> mean on the shard -> all_gather_into_tensor, as one waitable exchange of a flat bucket
    (its length is a multiple of the world size; both collectives in place: the rank's shard is a view of the bucket).
    A GPU bucket on a gloo group (tests: two ranks on one card) runs the same two collectives on a host copy."""

    def __init__(self, flat, world, group):
        self.flat, self.world, self.group = flat, world, group
        self.buf = flat.detach().cpu() if _host_staged(flat, group) else flat
        rank = dist.get_rank(group)
        n = self.buf.numel() // world
        assert n * world == self.buf.numel(), "bucket length must be a multiple of the world size"
        self.shard = self.buf[rank * n:(rank + 1) * n]
        self.work = dist.reduce_scatter_tensor(self.shard, self.buf, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def wait(self):
        self.work.wait()
        self.shard.div_(self.world)
        dist.all_gather_into_tensor(self.buf, self.shard, group=self.group)
        if self.buf is not self.flat:
            self.flat.copy_(self.buf)
        return True


class GradBuckets:
    """Flat gradient storage with per-bucket async exchange.

    Parameters keep `.grad = None` while backward runs, so autograd hands every gradient over as the tensor it computed
    (no per-parameter add into a zeroed buffer: 647 add launches = 8 - 11 ms of a config-2 train step, profiles/r02_train_*);
    a post-accumulate hook collects it, and when the last gradient of a bucket has arrived the whole bucket is folded into
    its flat storage by ONE multi-tensor copy (or add, on the later micro-steps of an accumulation window) and -- on the last
    micro-step -- its exchange is launched.  `finish()` leaves `.grad` = the (averaged) view into the flat bucket for the
    optimiser; `release()` (after the optimiser step) sets `.grad` back to None."""

    def __init__(self, params: Iterable[nn.Parameter], bucket_bytes: int = 64 << 20, process_group=None,
                 exchange: str = "all_reduce"):
        if exchange not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"exchange={exchange!r}")
        self.exchange = exchange
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets: list[dict] = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):  # backward produces the last parameters' gradients first
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._close(cur)
        self._handles: list = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b["params"]):
                p.grad = None
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, pi)))
        self.sync_enabled = True

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        n = (n + self.world - 1) // self.world * self.world  # whole shards for the reduce-scatter form
        flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.buckets.append(dict(params=ps, flat=flat, views=views, fresh=[None] * len(ps), pending=len(ps), has_data=False,
                                 sent=False))

    def _make_hook(self, bi, pi):
        def hook(p):
            b = self.buckets[bi]
            if b["fresh"][pi] is None:
                b["pending"] -= 1
                b["fresh"][pi] = p.grad
            else:  # a second backward inside one micro-step (not the engine's own use): keep the sum
                b["fresh"][pi] = b["fresh"][pi] + p.grad
            p.grad = None
            if b["pending"] == 0:
                self._fold(b)
                if self.sync_enabled:
                    self._launch(b)
        return hook

    def _fold(self, b):
        """This micro-step's gradients of one bucket -> its flat storage (multi-tensor copy / add); parameters that got no
        gradient contribute zeros."""
        got = [(v, g) for v, g in zip(b["views"], b["fresh"]) if g is not None]
        if got:
            vs, gs = [v for v, _ in got], [g if g.dtype == v.dtype else g.to(v.dtype) for v, g in got]
            if b["has_data"]:
                torch._foreach_add_(vs, gs)
            else:
                torch._foreach_copy_(vs, gs)
        if not b["has_data"]:
            for v, g in zip(b["views"], b["fresh"]):
                if g is None:
                    v.zero_()
        b["has_data"] = True
        b["fresh"] = [None] * len(b["params"])
        b["pending"] = len(b["params"])
        b["folded"] = True

    def end_micro_step(self):
        """After a backward that is NOT the last of its accumulation window: fold the buckets whose hooks did not all fire."""
        for b in self.buckets:
            if not b.get("folded"):
                self._fold(b)
            b["folded"] = False

    def _launch(self, b):
        b["sent"] = True
        if self.world == 1:
            return
        if self.exchange == "reduce_scatter":
            self._handles.append(_ScatterGather(b["flat"], self.world, self.group))
        else:
            self._handles.append(all_reduce_sum(b["flat"], self.group, async_op=True))

    def finish(self):
        """Fold and exchange what is still outstanding (a parameter without a gradient this step never fires its hook), wait,
        turn sums into means and hand the flat views to the parameters as `.grad`."""
        for b in self.buckets:
            if not b.get("folded"):
                self._fold(b)
            b["folded"] = False
            if not b["sent"]:
                self._launch(b)
        for h in self._handles:
            h.wait()
        if self.world > 1 and self.exchange == "all_reduce":
            for b in self.buckets:
                b["flat"].div_(self.world)
        self._handles.clear()
        for b in self.buckets:
            b["sent"] = False
            for p, v in zip(b["params"], b["views"]):
                p.grad = v

    def release(self):
        """After the optimiser step: `.grad` = None again (the next backward's gradients arrive as fresh tensors)."""
        for b in self.buckets:
            b["has_data"] = False
            for p in b["params"]:
                p.grad = None

    def nbytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)


class DataParallelEngine:
    """Replicated model + sharded batch.  `train_step` = forward, backward (overlapped gradient
    all-reduce), optimiser step every `accumulation` calls -- the loop body of train.py:190-202."""

    def __init__(self, model: nn.Module, lr: float = 5e-5, accumulation: int = 1, bucket_bytes: int = 64 << 20,
                 optimizer: torch.optim.Optimizer | None = None, process_group=None, exchange: str = "all_reduce"):
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.accumulation = max(1, int(accumulation))
        if self.world > 1:  # identical replicas: rank 0's weights and buffers win
            for t in list(model.parameters()) + list(model.buffers()):
                broadcast_from(t.data, 0, process_group)
        self.buckets = GradBuckets(model.parameters(), bucket_bytes, process_group, exchange)
        if optimizer is None:  # train.py:174 (AdamW, lr 5e-5).  On a GPU the fused form: ONE kernel over all parameters instead of
            # the multi-tensor form's ~14 launches per 1k tensors (1.7 -> 0.3 ms per config-2 step); same update rule
            params = list(model.parameters())
            fused = bool(params) and all(p.is_cuda and p.is_floating_point() for p in params)
            optimizer = torch.optim.AdamW(params, lr=lr, fused=True) if fused else torch.optim.AdamW(params, lr=lr)
        self.optimizer = optimizer
        self._micro = 0
        crit = getattr(model, "criterion", None)
        if crit is not None and hasattr(crit, "world_size_fn"):
            crit.world_size_fn = self.reduce_num_masks

    def reduce_num_masks(self, n: torch.Tensor):
        """SUM over ranks of the per-rank target count, and the world size (HF:781-794)."""
        if self.world > 1:
            all_reduce_sum(n, self.group)
        return n, self.world

    def backward_and_step(self, loss: torch.Tensor) -> bool:
        """Returns True when the optimiser stepped."""
        self._micro += 1
        last = self._micro % self.accumulation == 0
        self.buckets.sync_enabled = last  # exchange only on the last micro-step of an accumulation window
        (loss / self.accumulation).backward()
        if not last:
            self.buckets.end_micro_step()
            return False
        self.buckets.finish()
        self.optimizer.step()
        self.buckets.release()
        return True

    def train_step(self, pixel_values, mask_labels, class_labels):
        out = self.model(pixel_values=pixel_values, mask_labels=mask_labels, class_labels=class_labels)
        self.backward_and_step(out.loss)
        return out.loss.detach()
