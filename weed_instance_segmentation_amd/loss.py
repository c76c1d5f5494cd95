"""Hungarian matcher + Mask2Former loss on the GPU.

Mirrors `Mask2FormerHungarianMatcher` (HF:378-481) and `Mask2FormerLoss` (HF:485-794):
same sampling scheme, same weights, same result, but organised for one MI355X per process:

  * the cost matrices of EVERY prediction level and EVERY image come from one batched kernel
    call (K4, ops.matcher_cost); the reference syncs once per (image, level) -- B x 10 times per
    step (HF:474);
  * the linear sum assignment runs on the device too (ops.lsa_batched: scipy's own algorithm, arithmetic
    and tie rule -- bit-identical indices), so the loss has NO host synchronisation: the index tensors
    of every loss term are built by device ops from the solver's output (`DeviceIndices`).  scipy on a
    host copy remains the route for matrices beyond the kernel's size and for the per-level loss path;
  * point sampling of predictions / targets uses the HIP sampler (ops.point_sample).

Random points: the dependency draws from torch's global generator in call order (HF:455, :705,
:721).  Here a `point_provider` supplies them by ROLE, so tests can replay recorded draws;
the default provider draws on the device.
"""
from __future__ import annotations


from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from . import ops


class DevicePointProvider:
    """Default: fresh uniform points on the device.  `level` counts prediction levels in decoder order."""

    def __init__(self, device, generator: torch.Generator | None = None):
        self.device, self.generator = device, generator

    def _rand(self, *shape):
        return torch.rand(*shape, device=self.device, generator=self.generator)

    def matcher_points(self, n_levels: int, batch: int, num_points: int) -> torch.Tensor:
        return self._rand(n_levels, batch, num_points, 2)

    def oversample_points(self, level: int, n_masks: int, n: int) -> torch.Tensor:
        return self._rand(n_masks, n, 2)

    def random_points(self, level: int, n_masks: int, n: int) -> torch.Tensor:
        return self._rand(n_masks, n, 2)


class ReplayPointProvider:
    """Replays draws recorded from the dependency (tests): its order is, per criterion call
    (final level first, then auxiliary levels 0..n-2): B matcher draws (1,P,2), one oversample
    draw (M,3P,2), one random draw (M,P/4,2)."""

    def __init__(self, draws: Sequence[torch.Tensor], n_levels: int, batch: int, device):
        self.draws, self.n_levels, self.batch, self.device = list(draws), n_levels, batch, device
        self.per_level = batch + 2

    def _slot(self, level: int) -> int:
        order = [self.n_levels - 1] + list(range(self.n_levels - 1))
        return order.index(level) * self.per_level

    def matcher_points(self, n_levels, batch, num_points):
        out = [torch.cat([self.draws[self._slot(l) + b] for b in range(batch)], 0) for l in range(n_levels)]
        return torch.stack(out).to(self.device)

    def oversample_points(self, level, n_masks, n):
        return self.draws[self._slot(level) + self.batch].to(self.device)

    def random_points(self, level, n_masks, n):
        return self.draws[self._slot(level) + self.batch + 1].to(self.device)


_CONST_CACHE: dict = {}


def _dev_const(values, device, dtype=torch.int64) -> torch.Tensor:
    """Small host-known integer tables (level order, per-image offsets and counts, the image / slot of every matched pair) as
    device tensors, cached by value: a train loop sees the same few target-count patterns again and again, and a host-to-device
    copy per step -- however small -- is a point where the host can wait for the stream."""
    key = (tuple(int(v) for v in values), str(device), dtype)
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 512:
            _CONST_CACHE.clear()
        t = torch.tensor(key[0], dtype=dtype).to(device)
        _CONST_CACHE[key] = t
    return t


class DeviceIndices:
    """The assignment of every (level, image) as the device solver left it: rows / cols (NL, B, t_cap) int32 -- the matched
    (query, target) pairs sorted by query, `matched[b]` = min(Q, T_b) of them valid -- plus the flat views the loss terms
    index with, all built by device ops (no synchronisation).  Iterating / indexing it like the host form (`[level][image]` ->
    (rows, cols) int64 CPU tensors) copies to the host on first use: for callers and tests, not for the loss."""

    def __init__(self, rows: torch.Tensor, cols: torch.Tensor, matched: list[int]):
        self.rows, self.cols, self.matched = rows, cols, [int(m) for m in matched]
        NL, B, cap = rows.shape
        dev = rows.device
        b_list = [i for i, m in enumerate(self.matched) for _ in range(m)]
        s_list = [k for m in self.matched for k in range(m)]
        self.b_idx, self.slot = _dev_const(b_list, dev), _dev_const(s_list, dev)  # (M,) image / position in the image's matched list
        flat = _dev_const([b * cap + k for b, k in zip(b_list, s_list)], dev)
        self.q = rows.reshape(NL, B * cap)[:, flat].long()      # (NL, M) matched query
        self.t = cols.reshape(NL, B * cap)[:, flat].long()      # (NL, M) matched target (index inside its image)
        self._host = None

    @property
    def M(self) -> int:
        return int(self.b_idx.numel())

    def host(self):
        if self._host is None:
            r, c = self.rows.cpu().long(), self.cols.cpu().long()
            self._host = [[(r[l, b, :m].clone(), c[l, b, :m].clone()) for b, m in enumerate(self.matched)] for l in range(r.shape[0])]
        return self._host

    def __len__(self):
        return self.rows.shape[0]

    def __getitem__(self, level):
        return self.host()[level]

    def __iter__(self):
        return iter(self.host())


class _LevelView:
    """`indices[level]` of a DeviceIndices for callers that iterate images (the model's `matched_indices`): lazy host copy."""

    def __init__(self, di: DeviceIndices, level: int):
        self.di, self.level = di, level

    def __iter__(self):
        return iter(self.di.host()[self.level])

    def __len__(self):
        return len(self.di.matched)

    def __getitem__(self, i):
        return self.di.host()[self.level][i]


class Mask2FormerLoss(nn.Module):
    def __init__(self, config, weight_dict):
        super().__init__()
        self.num_labels = config.num_labels
        self.weight_dict = weight_dict
        self.eos_coef = config.no_object_weight
        ew = torch.ones(self.num_labels + 1)
        ew[-1] = self.eos_coef
        self.register_buffer("empty_weight", ew)
        self.num_points = config.train_num_points
        self.oversample_ratio = config.oversample_ratio
        self.importance_sample_ratio = config.importance_sample_ratio
        self.batched_levels = True  # loss_masks_all_levels; False = one pass per level (kept for A/B and tests)
        # mask losses on the matched queries' rows recomputed by one einsum (matched_row_logits) instead of the dense
        # predictions; this attribute = False restores the dense route (A/B runs and tests)
        self.matched_row_masks = True
        self.device_lsa = True  # ops.lsa_batched instead of scipy on a host copy (same indices); False: the host route
        self.cost_class, self.cost_mask, self.cost_dice = config.class_weight, config.mask_weight, config.dice_weight
        self.world_size_fn = None  # set by parallel.DataParallelEngine: all-reduces num_masks (HF:781-794)

    # ---------------------------------------------------------------- matcher (all levels at once)
    @torch.no_grad()
    def match(self, all_masks, all_classes, tgt, counts, cls, points):
        """Returns indices[level][image] = (rows int64, cols int64) -- HF:413-481."""
        NL, B = len(all_masks), all_masks[0].shape[0]
        if sum(counts) == 0:
            e = torch.zeros(0, dtype=torch.int64)
            return [[(e, e) for _ in range(B)] for _ in range(NL)]
        ml = [m.detach().float() for m in all_masks]  # used where they are: a stacked copy is 10 x 210 MB at config 2
        cl = torch.stack([c.detach() for c in all_classes]) if NL > 1 else all_classes[0].detach()[None]
        # (the cost sums over the P random points of an (image, level) do not depend on their order: wm2f_matcher_cost groups
        # them by map band itself and samples from LDS -- csrc/matcher.hip, band form; a sort of the points by pixel used to
        # stand here, 1.4 ms per config-2 step)
        cost = ops.matcher_cost(ml if NL <= 16 else torch.stack(ml), cl.float(), tgt, counts, cls, points, self.cost_class,
                                self.cost_mask, self.cost_dice)
        Q = cost.shape[2]
        if self.device_lsa and cost.is_cuda and Q <= 1024 and cost.shape[3] <= 1024 and min(counts) > 0:
            matched = [min(Q, c) for c in counts]
            rows, cols = ops.lsa_batched(cost, _dev_const(counts, cost.device, torch.int32), max(matched))
            return DeviceIndices(rows, cols, matched)
        cost = cost.cpu().numpy()  # the host route: ONE device->host sync of the step
        indices = []
        for l in range(NL):
            per = []
            for b in range(B):
                r, c = linear_sum_assignment(cost[l, b, :, :counts[b]])
                per.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
            indices.append(per)
        return indices

    @staticmethod
    def _uncertain_points(unc, pc, n_unc, P):
        """HF:688-704: the n_unc most uncertain of the oversampled points, as the first n_unc entries of a (rows, P, 2) tensor
        (the rest is the caller's: random points).  On the GPU a radix selection (ops.select_top_points: the set topk returns,
        in index order -- the losses sum over points); on the CPU the dependency's topk + gather."""
        if unc.is_cuda and n_unc > 0:
            return ops.select_top_points(unc.float().contiguous(), pc.float().contiguous(), n_unc, P)
        pts = torch.empty(unc.shape[0], P, 2, device=unc.device, dtype=pc.dtype)
        if n_unc > 0:
            idx = torch.topk(unc, k=n_unc, dim=1)[1]
            pts[:, :n_unc] = torch.gather(pc, 1, idx[..., None].expand(-1, -1, 2))
        return pts

    # ---------------------------------------------------------------- per-level losses
    def _num_masks(self, counts, device):
        n = torch.full((), float(sum(counts)), dtype=torch.float, device=device)  # (a fill, not a host-to-device copy)
        world = 1
        if self.world_size_fn is not None:
            n, world = self.world_size_fn(n)
        return torch.clamp(n / world, min=1)

    def loss_labels(self, classes, cls, offsets, indices):
        """HF:546-578."""
        B, Q, _ = classes.shape
        dev = classes.device
        bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)]).to(dev)
        si = torch.cat([s for s, _ in indices]).to(dev)
        ti = torch.cat([t + offsets[i] for i, (_, t) in enumerate(indices)]).to(dev)
        target = torch.full((B, Q), self.num_labels, dtype=torch.int64, device=dev)
        target[bi, si] = cls[ti]
        return F.cross_entropy(classes.transpose(1, 2), target, weight=self.empty_weight)

    def loss_labels_all_levels(self, all_classes, cls, offsets, indices, order):
        """`loss_labels` (HF:546-578) for every level of `order` with one cross-entropy call: (len(order),) tensor of the
        weighted means sum(w_i * nll_i) / sum(w_i) that nn.CrossEntropyLoss(weight=...) returns per level."""
        B, Q, C1 = all_classes[0].shape
        dev = all_classes[0].device
        logits = torch.stack([all_classes[lvl] for lvl in order])  # (NL, B, Q, C1): tiny
        target = torch.full((len(order), B, Q), self.num_labels, dtype=torch.int64, device=dev)
        if isinstance(indices, DeviceIndices):  # device ops only
            od = _dev_const(order, dev)
            M = indices.M
            li = _dev_const(range(len(order)), dev)[:, None].expand(-1, M)
            off = _dev_const(offsets[:-1], dev)[indices.b_idx]
            target[li, indices.b_idx[None].expand(len(order), -1), indices.q[od]] = cls[indices.t[od] + off[None]]
        else:
            li = torch.cat([torch.full((sum(int(s.numel()) for s, _ in indices[lvl]),), n, dtype=torch.int64) for n, lvl in enumerate(order)])
            bi = torch.cat([torch.full_like(s, i) for lvl in order for i, (s, _) in enumerate(indices[lvl])])
            si = torch.cat([s for lvl in order for s, _ in indices[lvl]])
            ti = torch.cat([t + offsets[i] for lvl in order for i, (_, t) in enumerate(indices[lvl])])
            target[li.to(dev), bi.to(dev), si.to(dev)] = cls[ti.to(dev)]
        nll = F.cross_entropy(logits.reshape(-1, C1).float(), target.reshape(-1), weight=self.empty_weight, reduction="none")
        wts = self.empty_weight[target.reshape(-1)]
        return nll.view(len(order), -1).sum(1) / wts.view(len(order), -1).sum(1)

    def loss_masks(self, masks, tgt, offsets, indices, num_masks, level, provider):
        """HF:580-640 with the uncertainty sampling of HF:671-724.  Matched prediction and target
        maps are sampled IN PLACE through an index (no (M,H,W) gathered copies, HF:602-609)."""
        dev = masks.device
        B, Q, h, w = masks.shape
        pred_idx = torch.cat([s + i * Q for i, (s, _) in enumerate(indices)]).to(device=dev, dtype=torch.int32)
        tgt_idx = torch.cat([t + offsets[i] for i, (_, t) in enumerate(indices)]).to(device=dev, dtype=torch.int32)
        M, P = int(pred_idx.shape[0]), self.num_points
        if M == 0:
            z = masks.sum() * 0.0
            return z, z
        maps = masks.reshape(B * Q, h, w).float()
        n_over = int(P * self.oversample_ratio)
        n_unc = int(self.importance_sample_ratio * P)
        with torch.no_grad():
            pc = provider.oversample_points(level, M, n_over)
            unc = -ops.point_sample(maps.detach(), pc, pred_idx).abs()
            pts = self._uncertain_points(unc, pc, n_unc, P)
            if P - n_unc > 0:
                pts[:, n_unc:] = provider.random_points(level, M, P - n_unc)
            point_labels = ops.point_sample(tgt, pts, tgt_idx)
        point_logits = ops.point_sample(maps, pts, pred_idx)
        bce = F.binary_cross_entropy_with_logits(point_logits, point_labels, reduction="none")
        loss_mask = bce.mean(1).sum() / num_masks  # HF:308-324
        probs = point_logits.sigmoid()
        num = 2 * (probs * point_labels).sum(-1)
        den = probs.sum(-1) + point_labels.sum(-1)
        loss_dice = (1 - (num + 1) / (den + 1)).sum() / num_masks  # HF:278-305
        return loss_mask, loss_dice

    def matched_row_logits(self, rows, indices, order):
        """The mask logits of the MATCHED queries only, for every level of `order`, from ONE einsum: rows =
        (inter, mask_embedder, mask_features) -- the decoder's normalised states per level (B, Q, d), the mask-embedding
        MLP and the pixel features (HF:2040-2046).  The dependency takes these rows out of the dense (B, Q, H, W)
        predictions (HF:602-609), whose backward is then a dense einsum backward per level over 100 query rows of which
        at most `targets` are non-zero, plus nine accumulations of the 0.5 GB pixel-feature gradient.  Recomputing the
        matched rows -- (B, levels x T_max, C) embeddings against the same pixel features, same kernel, same per-row sums --
        makes that backward ONE einsum backward over levels x T_max rows; the dense predictions keep feeding the matcher
        and the caller and receive no gradient from this loss.  Returns (maps (B * NL * T_max, h, w) fp32, index (NL, M))."""
        inter, embedder, pix = rows
        B = pix.shape[0]
        NL = len(order)
        dev = pix.device
        dev_idx = isinstance(indices, DeviceIndices)
        counts = indices.matched if dev_idx else [int(s.numel()) for s, _ in indices[order[0]]]
        # whole multiples of 4 rows per level: the hand-written K3 backward wants Q % 4 == 0 (an odd count would silently take
        # the library bmm pair); the extra rows are zero embeddings that no index points at
        t_max = (max(max(counts), 1) + 3) // 4 * 4
        if dev_idx:
            b_idx, t_idx = indices.b_idx, indices.slot
            q_idx = indices.q[_dev_const(order, dev)]  # (NL, M)
        else:
            b_idx = torch.cat([torch.full((c,), i, dtype=torch.long) for i, c in enumerate(counts)]).to(dev)
            t_idx = torch.cat([torch.arange(c) for c in counts]).to(dev)
            q_idx = torch.stack([torch.cat([s for s, _ in indices[lvl]]) for lvl in order]).to(dev)  # (NL, M)
        h_m = torch.stack([inter[lvl][b_idx, q_idx[n]] for n, lvl in enumerate(order)])            # (NL, M, d)
        emb_m = embedder(h_m)                                                                     # (NL, M, C)
        C = emb_m.shape[-1]
        e_all = emb_m.new_zeros(B, NL * t_max, C)
        slot = (torch.arange(NL, device=dev)[:, None] * t_max + t_idx[None, :])                     # (NL, M)
        e_all = e_all.index_put((b_idx[None, :].expand(NL, -1), slot), emb_m)
        if pix.dtype == torch.bfloat16 and pix.shape[1] % 32 == 0 and pix.shape[1] <= 512:
            pix_c = pix.contiguous()
            pix_t = ops.nchw_to_pixel_major_bf16(pix_c)
            per = max(1, 112 // t_max) * t_max  # whole levels per call, at most 112 rows: the bf16 backward kernel's limit
            parts = [ops.mask_einsum_bf16(e_all[:, r0:r0 + per].contiguous(), pix_c, pix_t) for r0 in range(0, NL * t_max, per)]
            logits = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        else:
            logits = ops.mask_einsum(e_all.float(), pix.float())
        index = (b_idx[None, :] * (NL * t_max) + slot).to(torch.int32)
        return logits.reshape(B * NL * t_max, logits.shape[2], logits.shape[3]), index

    def loss_masks_all_levels(self, all_masks, tgt, offsets, indices, num_masks, order, provider, rows=None):
        """`loss_masks` for every level of `order` in one pass: the level tensors are sampled where
        they are (pointer table), the top-k, the target sampling and the BCE / dice reductions run once over
        (levels x matched masks) rows.  Returns two (len(order),) tensors: loss_mask, loss_dice per level.
        rows: see `matched_row_logits` -- the predictions are then sampled from the matched-row maps."""
        dev = all_masks[0].device
        B, Q, h, w = all_masks[0].shape
        NL, P = len(order), self.num_points
        if isinstance(indices, DeviceIndices):  # device ops only
            od = _dev_const(order, dev)
            pred_idx = indices.q[od] + indices.b_idx[None] * Q
            tgt_idx = indices.t[od] + _dev_const(offsets[:-1], dev)[indices.b_idx][None]
        else:
            pred_idx = torch.stack([torch.cat([s + i * Q for i, (s, _) in enumerate(indices[lvl])]) for lvl in order])
            tgt_idx = torch.stack([torch.cat([t + offsets[i] for i, (_, t) in enumerate(indices[lvl])]) for lvl in order])
        M = int(pred_idx.shape[1])
        if M == 0:
            z = torch.stack([all_masks[lvl].sum() * 0.0 for lvl in order])
            return z, z
        pred_idx = pred_idx.to(device=dev, dtype=torch.int32)
        tgt_idx = tgt_idx.to(device=dev, dtype=torch.int32)
        if rows is not None:
            compact, cidx = self.matched_row_logits(rows, indices, order)
            maps, pred_idx, lv_shape = [compact], cidx.view(1, NL * M), (1, NL * M)
        else:
            maps, lv_shape = [all_masks[lvl].reshape(B * Q, h, w).float() for lvl in order], (NL, M)
        n_over = int(P * self.oversample_ratio)
        n_unc = int(self.importance_sample_ratio * P)
        with torch.no_grad():
            pc = torch.stack([provider.oversample_points(lvl, M, n_over) for lvl in order])  # (NL, M, n_over, 2)
            unc = ops.point_sample_levels([m.detach() for m in maps], pc.view(*lv_shape, n_over, 2), pred_idx, neg_abs=True)
            pts = self._uncertain_points(unc.view(NL * M, n_over), pc.view(NL * M, n_over, 2), n_unc, P)
            if P - n_unc > 0:
                pts[:, n_unc:] = torch.stack([provider.random_points(lvl, M, P - n_unc) for lvl in order]).view(NL * M, P - n_unc, 2)
            point_labels = ops.point_sample(tgt, pts, tgt_idx.view(-1))
        point_logits = ops.point_sample_levels(maps, pts.view(*lv_shape, P, 2), pred_idx, unique_index=True)  # a one-to-one assignment: no map twice
        bce, dice = ops.mask_loss_rows(point_logits.view(NL * M, P), point_labels)
        return bce.view(NL, M).sum(1) / num_masks, dice.view(NL, M).sum(1) / num_masks

    def forward(self, all_masks, all_classes, mask_labels, class_labels, point_provider=None, matched_rows=None):
        """all_masks / all_classes: per-level lists in decoder order, LAST = final prediction.
        matched_rows = (inter, mask_embedder, mask_features), lists in the same level order: the mask losses then read the
        matched queries' logits from one recomputed einsum (`matched_row_logits`) instead of the dense predictions.
        Returns (weighted loss dict with the dependency's key names, indices of the final level)."""
        NL, B = len(all_masks), all_masks[0].shape[0]
        dev = all_masks[0].device
        provider = point_provider or DevicePointProvider(dev)
        counts = [int(m.shape[0]) for m in mask_labels]
        offsets = [0]
        for c in counts:
            offsets.append(offsets[-1] + c)
        if len({tuple(m.shape[1:]) for m in mask_labels}) != 1:
            raise NotImplementedError("mask_labels of different sizes in one batch (collate_fn stacks equal sizes)")
        tgt = torch.cat([m.to(dev) for m in mask_labels], 0)  # once per step, shared by matcher and every level
        if tgt.dtype == torch.bool:
            tgt = tgt.view(torch.uint8)
        elif tgt.dtype not in (torch.uint8, torch.float32):
            tgt = tgt.float()
        cls = torch.cat([c.to(dev) for c in class_labels], 0).to(torch.int64)
        points = provider.matcher_points(NL, B, self.num_points)
        indices = self.match(all_masks, all_classes, tgt, counts, cls, points)
        num_masks = self._num_masks(counts, dev)
        losses = {}
        order = [NL - 1] + list(range(NL - 1))  # HF:762-777: final level first, then aux 0..n-2
        dev_idx = isinstance(indices, DeviceIndices)
        same_m = dev_idx or len({sum(int(s.numel()) for s, _ in indices[lvl]) for lvl in order}) == 1
        batched = self.batched_levels and same_m and NL <= 16
        if batched:
            rows = matched_rows if (matched_rows is not None and self.matched_row_masks and all_masks[0].is_cuda) else None
            m_total = indices.M if dev_idx else sum(int(s.numel()) for s, _ in indices[order[0]])
            if rows is not None and NL * m_total >= 65536:
                # the matched-row route folds the levels into ONE map list of NL * M rows, and the point samplers carry the
                # row in grid.y (< 65536): beyond that (33+ fully matched 200-query images per rank at 10 levels) the dense
                # route, whose limit is per level, runs instead
                rows = None
            lm_all, ld_all = self.loss_masks_all_levels(all_masks, tgt, offsets, indices, num_masks, order, provider, rows=rows)
            lc_all = self.loss_labels_all_levels(all_classes, cls, offsets, indices, order)
        for n, lvl in enumerate(order):
            if batched:
                lm, ld, lc = lm_all[n], ld_all[n], lc_all[n]
            else:
                lm, ld = self.loss_masks(all_masks[lvl], tgt, offsets, indices[lvl], num_masks, lvl, provider)
                lc = self.loss_labels(all_classes[lvl], cls, offsets, indices[lvl])
            suffix = "" if n == 0 else f"_{lvl}"
            losses["loss_mask" + suffix] = lm * self.weight_dict["loss_mask"]
            losses["loss_dice" + suffix] = ld * self.weight_dict["loss_dice"]
            losses["loss_cross_entropy" + suffix] = lc * self.weight_dict["loss_cross_entropy"]
        return losses, (_LevelView(indices, NL - 1) if dev_idx else indices[NL - 1])
