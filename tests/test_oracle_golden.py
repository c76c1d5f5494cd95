"""Pin the CPU oracle (oracle/m2f_oracle.py) against the golden vectors that
tests/golden/make_golden.py produced from transformers 5.15.0 (CPU).  CPU only."""
import json

import numpy as np
import pytest
import torch

from oracle import m2f_oracle as O
from conftest import load_golden

T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag", ["toy", "rect", "d8"])
def test_k1_core(tag):
    g = load_golden(f"k1_msdeform_{tag}.npz")
    out = O.msdeform_attn_core(T(g["value"]), g["level_hw"], T(g["loc"]), T(g["w"]))
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-5, atol=1e-6)
    out2 = O.msdeform_attn_core_explicit(T(g["value"]), g["level_hw"], T(g["loc"]), T(g["w"]))
    torch.testing.assert_close(out2, T(g["out"]), rtol=1e-4, atol=2e-6)


def test_a2_module():
    g = load_golden("a2_msdeform_module.npz")
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    out, attn = O.msdeform_attn_module(sd, "", T(g["hidden"]), T(g["pos"]), T(g["ref"]), g["level_hw"], int(g["n_heads"]))
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(attn, T(g["attn"]), rtol=1e-5, atol=1e-6)
    ref = O.reference_points(g["level_hw"], g["hidden"].shape[0])
    torch.testing.assert_close(ref.contiguous(), T(g["ref"]), rtol=0, atol=1e-7)


@pytest.mark.parametrize("tag", ["small", "q100"])
def test_k2_masked_xattn(tag):
    g = load_golden(f"k2_masked_xattn_{tag}.npz")
    H = int(g["n_heads"])
    sd = {"in_proj_weight": T(g["in_proj_weight"]), "in_proj_bias": T(g["in_proj_bias"]),
          "out_proj.weight": T(g["out_proj_weight"]), "out_proj.bias": T(g["out_proj_bias"])}
    out = O.masked_cross_attention(sd, "", T(g["query"]), T(g["key"]), T(g["value"]), T(g["mask"]), H)
    torch.testing.assert_close(out, T(g["out"]), rtol=1e-4, atol=2e-5)


def test_k3_mask_predictor():
    g = load_golden("k3_mask_predictor.npz")
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    emb = O.mlp3(sd, "mask_embedder.", T(g["outputs"]).transpose(0, 1))
    torch.testing.assert_close(emb, T(g["mask_embeddings"]), rtol=1e-5, atol=1e-5)
    logits = O.mask_einsum(emb, T(g["pix"]))
    torch.testing.assert_close(logits, T(g["logits"]), rtol=1e-4, atol=1e-4)
    for i in range(5):
        m = O.attention_mask_from_logits(T(g["logits"]), g[f"size_{i}"])
        assert torch.equal(m, T(g[f"attn_mask_{i}"]))


def test_k4_matcher_cost_and_indices():
    g = load_golden("k4_matcher.npz")
    wc, wm, wd = [float(x) for x in g["weights"]]
    for i in range(g["mask_logits"].shape[0]):
        cost = O.matcher_cost(T(g["mask_logits"][i]), T(g["class_logits"][i]), T(g[f"mask_labels_{i}"]).float(),
                              T(g[f"class_labels_{i}"]), T(g["points"][i:i + 1]), wc, wm, wd)
        torch.testing.assert_close(cost, T(g[f"cost_{i}"]), rtol=2e-5, atol=2e-5)
        r, c = O.hungarian(cost)
        assert np.array_equal(r.numpy(), g[f"row_{i}"]) and np.array_equal(c.numpy(), g[f"col_{i}"])


def _full_inputs(g):
    cfg = json.loads(str(g["config_json"]))
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    B = g["pixel_values"].shape[0]
    ml = [T(g[f"mask_labels_{i}"]).float() for i in range(B)]
    cl = [T(g[f"class_labels_{i}"]) for i in range(B)]
    draws = [T(g[f"draw_{i}"]) for i in range(int(g["n_draws"]))]
    return cfg, sd, ml, cl, draws[cfg["decoder_layers"] - 1:]  # drop the decoder's per-layer scalars (HF:1905)


def test_full_forward_and_loss():
    g = load_golden("full_tiny.npz")
    cfg, sd, ml, cl, draws = _full_inputs(g)
    res = O.forward(sd, cfg, T(g["pixel_values"]), ml, cl, O.RandSource(draws))
    for i, f in enumerate(res["backbone"]):
        torch.testing.assert_close(f, T(g[f"backbone_{i}"]), rtol=1e-4, atol=1e-5)
    for i, f in enumerate(res["multi_scale"]):
        torch.testing.assert_close(f, T(g[f"multi_scale_{i}"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res["mask_features"], T(g["mask_features"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res["masks_queries_logits"], T(g["masks_queries_logits"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(res["class_queries_logits"], T(g["class_queries_logits"]), rtol=1e-4, atol=1e-4)
    for i, m in enumerate(res["aux_masks"]):
        torch.testing.assert_close(m, T(g[f"aux_masks_{i}"]), rtol=1e-3, atol=1e-4)
    for i, (r, c) in enumerate(res["indices"]):
        assert np.array_equal(r.numpy(), g[f"row_{i}"]) and np.array_equal(c.numpy(), g[f"col_{i}"])
    for k, v in res["loss_dict"].items():
        torch.testing.assert_close(v, T(g["ld." + k]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(res["loss"], T(g["loss"]), rtol=1e-4, atol=1e-4)


def test_num_labels_reads_either_config_form():
    """config.json stores id2label; a config dict may also carry num_labels (it used to recurse forever)."""
    assert O.num_labels({"id2label": {"0": "a", "1": "b", "2": "c"}}) == 3
    assert O.num_labels({"num_labels": 5, "id2label": {"0": "a"}}) == 5
    assert O.num_labels({"num_labels": None, "id2label": {"0": "a", "1": "b"}}) == 2
