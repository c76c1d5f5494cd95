"""Instance post-processing: the oracle against the dependency's own outputs (CPU), the HIP path
against both (GPU)."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m2f_oracle as O

T = torch.from_numpy


def _fixture():
    g = load_golden("postprocess_instances.npz")
    return g, json.loads(str(g["info_json"]))


def _same_info(a, b, tol=2e-6):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["id"] == y["id"] and x["label_id"] == y["label_id"] and x["was_fused"] == y["was_fused"]
        assert abs(x["score"] - y["score"]) <= tol


@pytest.mark.parametrize("tag", ["none", "mixed", "small"])
def test_oracle_matches_dependency(tag):
    g, info = _fixture()
    ts = info[tag]["target_sizes"]
    res = O.post_process_instance_segmentation(T(g["class_logits"]), T(g["mask_logits"]), 0.5, ts)
    for i, r in enumerate(res):
        _same_info(r["segments_info"], info[tag]["segments_info"][i], tol=0)
        assert torch.equal(r["segmentation"].to(torch.int16), T(g[f"seg_{tag}_{i}"]))


def test_oracle_binary_maps():
    g, info = _fixture()
    res = O.post_process_instance_segmentation(T(g["class_logits"]), T(g["mask_logits"]), 0.5, info["maps"]["target_sizes"],
                                               return_binary_maps=True)
    for i, r in enumerate(res):
        assert torch.equal(r["segmentation"].to(torch.int16), T(g[f"maps_{i}"]))


def test_rle_helpers_match_dependency():
    from weed_instance_segmentation_amd.postprocess import convert_segmentation_to_rle
    g, info = _fixture()
    res = O.post_process_instance_segmentation(T(g["class_logits"]), T(g["mask_logits"]), 0.5, info["rle"]["target_sizes"])
    for i, r in enumerate(res):
        assert convert_segmentation_to_rle(r["segmentation"]) == info["rle"]["segmentation"][i]


def _outputs(g):
    return SimpleNamespace(class_queries_logits=T(g["class_logits"]).cuda(), masks_queries_logits=T(g["mask_logits"]).cuda())


def _close_maps(a, b, max_frac=2e-4):
    """Pixel-exact up to sign flips of logits within float rounding of zero (the bilinear resize is re-evaluated on
    the GPU; a flip changes one pixel)."""
    a, b = a.cpu().to(torch.int16), b.to(torch.int16)
    assert a.shape == b.shape
    assert (a != b).float().mean().item() <= max_frac


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["none", "mixed", "small"])
def test_hip_matches_dependency(tag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
    g, info = _fixture()
    ts = info[tag]["target_sizes"]
    res = Mask2FormerInstancePostProcessor().post_process_instance_segmentation(_outputs(g), threshold=0.5, mask_threshold=0.5,
                                                                                target_sizes=ts)
    for i, r in enumerate(res):
        _same_info(r["segments_info"], info[tag]["segments_info"][i])
        assert r["segmentation"].is_cuda
        _close_maps(r["segmentation"], T(g[f"seg_{tag}_{i}"]))


@pytest.mark.gpu
def test_hip_binary_maps_and_rle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
    g, info = _fixture()
    p = Mask2FormerInstancePostProcessor()
    res = p.post_process_instance_segmentation(_outputs(g), threshold=0.5, target_sizes=info["maps"]["target_sizes"], return_binary_maps=True)
    for i, r in enumerate(res):
        _same_info(r["segments_info"], info["maps"]["segments_info"][i])
        _close_maps(r["segmentation"], T(g[f"maps_{i}"]))
    res = p.post_process_instance_segmentation(_outputs(g), threshold=0.5, target_sizes=info["rle"]["target_sizes"], return_coco_annotation=True)
    for i, r in enumerate(res):
        assert r["segmentation"] == info["rle"]["segmentation"][i]
    with pytest.raises(ValueError):
        p.post_process_instance_segmentation(_outputs(g), return_coco_annotation=True, return_binary_maps=True)


@pytest.mark.gpu
def test_hip_matches_oracle_at_model_size():
    """Q = 100, 256 x 256 logits, 1024 x 1024 targets (the reference's eval shape), random logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
    g = torch.Generator().manual_seed(5)
    B, Q, C = 2, 100, 3
    low = torch.randn(B, Q, 8, 8, generator=g) * 3.0 - 2.0
    masks = torch.nn.functional.interpolate(low, size=(256, 256), mode="bicubic", align_corners=False)
    cls = torch.randn(B, Q, C + 1, generator=g) * 3.0
    ts = [(1024, 1024)] * B
    ref = O.post_process_instance_segmentation(cls, masks, 0.5, ts)
    out = SimpleNamespace(class_queries_logits=cls.cuda(), masks_queries_logits=masks.cuda())
    res = Mask2FormerInstancePostProcessor().post_process_instance_segmentation(out, threshold=0.5, target_sizes=ts)
    for r, q in zip(res, ref):
        _same_info(r["segments_info"], q["segments_info"])
        _close_maps(r["segmentation"], q["segmentation"])
    assert sum(len(q["segments_info"]) for q in ref) > 5


def test_refuses_cpu_tensors():
    from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
    from weed_instance_segmentation_amd._lib import Wm2fError
    g, _ = _fixture()
    out = SimpleNamespace(class_queries_logits=T(g["class_logits"]), masks_queries_logits=T(g["mask_logits"]))
    with pytest.raises(Wm2fError):
        Mask2FormerInstancePostProcessor().post_process_instance_segmentation(out)
