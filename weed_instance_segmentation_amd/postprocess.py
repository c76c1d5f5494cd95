"""Device-side `post_process_instance_segmentation`.

Mirrors `Mask2FormerImageProcessor.post_process_instance_segmentation` of transformers 5.15.0
(models/mask2former/image_processing_mask2former.py:627-746) -- same arguments, same return structure -- for the
reference's call sites `models/metrics.py:58-63` and `models/mask2former/inference.py:30`:

    processor = Mask2FormerInstancePostProcessor()
    preds = processor.post_process_instance_segmentation(outputs, threshold=0.5, mask_threshold=0.5,
                                                         target_sizes=batch["target_sizes"])

What changes is where the work runs.  The dependency resizes all Q masks to 384 x 384, then loops over the queries in
Python with one `.item()` sync each.  Here the (query, class) selection of :695-701 -- a top-k over Q * C numbers --
runs on the host with the same torch routine the reference's CPU path uses (`topk(sorted=False)`: its order decides
which instance is painted last, and only the same routine reproduces it), everything that touches pixels runs in
the HIP kernels of csrc/postprocess.hip, and ONE device-to-host copy of (B, Q) scores / labels / flags builds the
`segments_info` dictionaries.  `mask_threshold` and `overlap_mask_area_threshold` are accepted and unused, exactly
as in the dependency's instance path.
"""
from __future__ import annotations

import torch

from . import ops


def binary_mask_to_rle(mask: torch.Tensor) -> list[int]:
    """COCO-style run lengths of a binary (H, W) mask, row-major (image_processing_mask2former.py:77-97)."""
    pixels = mask.flatten()
    zero = torch.zeros(1, device=pixels.device, dtype=pixels.dtype)
    pixels = torch.cat([zero, pixels, zero])
    runs = torch.where(pixels[1:] != pixels[:-1])[0] + 1
    runs[1::2] -= runs[::2]
    return runs.tolist()


def convert_segmentation_to_rle(segmentation: torch.Tensor) -> list[list[int]]:
    """One run-length list per distinct id of the map, background (-1) included (:100-118)."""
    return [binary_mask_to_rle(torch.where(segmentation == idx, 1, 0)) for idx in torch.unique(segmentation)]


class Mask2FormerInstancePostProcessor:
    """Stands where the reference holds its `AutoImageProcessor` for post-processing."""

    def post_process_instance_segmentation(self, outputs, threshold: float = 0.5, mask_threshold: float = 0.5,
                                           overlap_mask_area_threshold: float = 0.8, target_sizes=None,
                                           return_coco_annotation: bool = False, return_binary_maps: bool = False):
        if return_coco_annotation and return_binary_maps:
            raise ValueError("return_coco_annotation and return_binary_maps can not be both set to True.")
        cls = outputs.class_queries_logits
        logits = outputs.masks_queries_logits
        if not logits.is_cuda:
            from ._lib import Wm2fError
            raise Wm2fError(f"masks_queries_logits is on {logits.device}: the wm2f kernels run on a GPU only (no CPU fallback)")
        logits = logits.float().contiguous()
        B, Q = cls.shape[0], cls.shape[1]
        C = cls.shape[-1] - 1
        if target_sizes is not None and len(target_sizes) != B:
            raise ValueError("Make sure that you pass in as many target sizes as the batch dimension of the logits")

        # ---- (query, class) selection, :695-701, on the host (see the module docstring)
        cls_cpu = cls.detach().float().cpu()
        sel_scores, sel_labels, sel_q = [], [], []
        for i in range(B):
            scores = torch.nn.functional.softmax(cls_cpu[i], dim=-1)[:, :-1]
            s, idx = scores.flatten(0, 1).topk(Q, sorted=False)
            sel_scores.append(s)
            sel_labels.append(idx % C)
            sel_q.append(torch.div(idx, C, rounding_mode="floor"))
        dev = logits.device
        sel_scores = torch.stack(sel_scores).to(dev)
        labels_cpu = torch.stack(sel_labels)
        qidx = torch.stack(sel_q).to(torch.int32).to(dev)

        # ---- mask quality on the 384 x 384 grid, :703-709
        sum_sig, cnt = ops.instance_scores(logits, qidx)
        pred_scores = sel_scores * (sum_sig / (cnt + 1e-6))
        cand = pred_scores >= threshold

        sizes = [tuple(int(v) for v in t) for t in target_sizes] if target_sizes is not None else [ops._GRID] * B
        seg_out: list = [None] * B
        keep_all = torch.zeros(B, Q, dtype=torch.bool, device=dev)
        kept_q_all = torch.zeros(B, Q, dtype=torch.int32, device=dev)
        for size in dict.fromkeys(sizes):  # one launch group per distinct target size
            rows = [i for i in range(B) if sizes[i] == size]
            ridx = torch.tensor(rows, device=dev)
            lg, qi = logits[ridx], qidx[ridx]
            if size[0] >= ops._GRID[0] and size[1] >= ops._GRID[1]:
                nonempty = cnt[ridx] > 0  # `nearest` up-sampling keeps every grid pixel
            else:
                nonempty = ops.instance_any(lg, qi, cand[ridx].to(torch.uint8), size) > 0
            keep = cand[ridx] & nonempty  # :724
            # kept instances in query order -> ids 0 .. n-1 (:725-735)
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
            kept_q = torch.gather(qi, 1, order).contiguous()
            n_kept = keep.sum(1).to(torch.int32)
            seg = ops.instance_segmentation(lg, kept_q, n_kept, size)
            keep_all[ridx] = keep
            kept_q_all[ridx] = kept_q
            for j, i in enumerate(rows):
                seg_out[i] = seg[j]

        # ---- the one device-to-host copy
        keep_cpu, score_cpu = keep_all.cpu(), pred_scores.cpu()
        results = []
        for i in range(B):
            ks = torch.nonzero(keep_cpu[i]).flatten().tolist()
            segments = [{"id": r, "label_id": int(labels_cpu[i, j]), "was_fused": False, "score": round(float(score_cpu[i, j]), 6)}
                        for r, j in enumerate(ks)]
            segmentation = seg_out[i]
            if return_coco_annotation:
                segmentation = convert_segmentation_to_rle(segmentation)
            if return_binary_maps and ks:
                segmentation = ops.instance_maps(logits[i], kept_q_all[i], len(ks), sizes[i])
            results.append({"segmentation": segmentation, "segments_info": segments})
        return results
