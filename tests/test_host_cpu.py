"""CPU-side checks: C-ABI exports, state-dict naming, persistence round trip, product/oracle
separation, and that the product path REFUSES to run without a GPU (no fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "weed_instance_segmentation_amd")


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "wm2f.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wm2f_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from weed_instance_segmentation_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/wm2f.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    lib.wm2f_version.restype = ctypes.c_int
    assert lib.wm2f_version() == 100  # host-only call, no GPU needed


def test_state_dict_names_match_dependency():
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    ks = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    cfg = Mask2FormerConfig.from_dict(ks["resnet50_config"])
    with torch.device("meta"):
        m = Mask2FormerForUniversalSegmentation.__new__(Mask2FormerForUniversalSegmentation)
        torch.nn.Module.__init__(m)
    m = Mask2FormerForUniversalSegmentation(cfg)
    sd = m.state_dict()
    exp = ks["resnet50_q100_l3"]
    assert set(sd) == set(exp)
    assert all(list(sd[k].shape) == exp[k] for k in exp)


def test_save_and_from_pretrained_round_trip(tmp_path):
    from conftest import load_golden
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    g = load_golden("full_tiny.npz")
    cfg = Mask2FormerConfig.from_dict(json.loads(str(g["config_json"])))
    m = Mask2FormerForUniversalSegmentation(cfg)
    m.save_pretrained(str(tmp_path))
    assert os.path.exists(tmp_path / "config.json") and os.path.exists(tmp_path / "model.safetensors")
    m2 = Mask2FormerForUniversalSegmentation.from_pretrained(str(tmp_path))
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # the reference re-heads the class predictor (train.py:167-172)
    id2label = {0: "soil", 1: "crop", 2: "weed", 3: "partial_crop", 4: "partial_weed"}
    with pytest.raises(RuntimeError):
        Mask2FormerForUniversalSegmentation.from_pretrained(str(tmp_path), id2label=id2label,
                                                            label2id={v: k for k, v in id2label.items()})
    m3 = Mask2FormerForUniversalSegmentation.from_pretrained(str(tmp_path), id2label=id2label,
                                                             label2id={v: k for k, v in id2label.items()},
                                                             ignore_mismatched_sizes=True)
    assert m3.class_predictor.weight.shape[0] == 6 and m3.config.id2label[2] == "weed"
    with pytest.raises(FileNotFoundError):
        Mask2FormerForUniversalSegmentation.from_pretrained("facebook/mask2former-swin-large-coco-instance")


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|m2f_oracle|import transformers|from transformers", txt, re.M):
                    bad.append(f)
    assert not bad, bad


def test_product_refuses_cpu_tensors():
    """No silent fallback: on a host without a GPU the forward must raise, not compute."""
    from conftest import load_golden
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    from weed_instance_segmentation_amd._lib import Wm2fError
    g = load_golden("full_tiny.npz")
    m = Mask2FormerForUniversalSegmentation(Mask2FormerConfig.from_dict(json.loads(str(g["config_json"])))).eval()
    with pytest.raises(Wm2fError), torch.no_grad():
        m(pixel_values=torch.from_numpy(g["pixel_values"]))


def test_collate_fn_batch_contract():
    """The batch dict the boundary consumes (datasets/dataset_utils.py:32-53), restated."""
    items = [dict(pixel_values=torch.zeros(3, 8, 8), mask_labels=torch.zeros(i + 1, 8, 8), class_labels=torch.zeros(i + 1, dtype=torch.int64))
             for i in range(2)]
    batch = dict(pixel_values=torch.stack([it["pixel_values"] for it in items]),
                 mask_labels=[it["mask_labels"] for it in items], class_labels=[it["class_labels"] for it in items])
    assert batch["pixel_values"].shape == (2, 3, 8, 8) and isinstance(batch["mask_labels"], list)
