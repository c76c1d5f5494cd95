"""CPU-side checks: C-ABI exports, state-dict naming, persistence round trip, product/oracle
separation, and that the product path REFUSES to run without a GPU (no fallback)."""
import ctypes
import json
import os
import re

import pytest
import numpy as np
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


def test_production_library_has_no_profiling_surface():
    """include/wm2f.h promises: no global mutable state, no environment reads, valid outputs only.  The stamp buffer,
    wm2f_debug_stamps, the K1 timing ablations and the getenv knobs of K2 / K3 live in libwm2f_prof.so
    (include/wm2f_prof.h, -DWM2F_PROFILING), which only tools/ load."""
    import subprocess
    from weed_instance_segmentation_amd import _build, _lib
    lib = ctypes.CDLL(_build.build())
    for name in _lib.PROF_SIGNATURES:
        assert not hasattr(lib, name), f"production library exports {name}"
    nm = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert " getenv" not in nm and "g_stamps" not in nm, "production library reads the environment / holds the stamp buffer"
    prof_h = open(os.path.join(ROOT, "include", "wm2f_prof.h")).read()
    declared = set(re.findall(r"\b(wm2f_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", prof_h, flags=re.S)))
    assert declared == set(_lib.PROF_SIGNATURES)
    prof = ctypes.CDLL(_build.build(prof=True))
    for name in list(_lib.SIGNATURES) + list(_lib.PROF_SIGNATURES):
        assert hasattr(prof, name), name
    # nothing the product, the tests' parity checks or bench.py import ever switches libraries
    for path in [os.path.join(PKG, f) for f in os.listdir(PKG) if f.endswith(".py") and f != "_lib.py"] + [os.path.join(ROOT, "bench.py")]:
        assert "use_profiling_library" not in open(path).read(), path


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


def test_fused_pass_ops_refuse_cpu_tensors():
    """The fused inference passes have no CPU form either: each raises on host tensors instead of computing."""
    from weed_instance_segmentation_amd import ops
    from weed_instance_segmentation_amd._lib import Wm2fError
    x, b = torch.randn(1, 8, 8, 8), torch.randn(8)
    calls = [
        lambda: ops.bias_act_(x.clone(), b),
        lambda: ops.group_norm_act_(x.clone(), 2, b, b, 1e-5, up=torch.randn(1, 8, 4, 4)),
        lambda: ops.group_norm_tokens_(x, b, 2, b, b, 1e-5, torch.zeros(1, 64, 8), 0),
        lambda: ops.bias_relu_maxpool(x, b),
        lambda: ops.resize_bilinear(x, (4, 4)),
        lambda: ops.tokens_to_nchw(torch.randn(1, 64, 8), 0, 8, 8),
        lambda: ops.add_broadcast(torch.randn(2, 5, 8), torch.randn(1, 5, 8)),
        lambda: ops.select_top_points(torch.randn(2, 16), torch.rand(2, 16, 2), 4),
    ]
    for call in calls:
        with pytest.raises(Wm2fError):
            call()


def test_swin_backbone_matches_golden_and_names():
    """The Swin backbone is stock torch ops, so its parity runs on CPU: feature maps vs the fixture made
    from transformers' SwinBackbone (odd input sizes: every padding path), and the full Swin-T key set."""
    import numpy as np
    from conftest import load_golden
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    from weed_instance_segmentation_amd.backbone_swin import SwinBackbone
    g = load_golden("swin_tiny_backbone.npz")
    cfg = json.loads(str(g["config_json"]))
    m = SwinBackbone(cfg).eval()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)
    for tag in ("a", "b"):
        with torch.no_grad():
            fm = m(torch.from_numpy(g[f"x_{tag}"]))
        for i, f in enumerate(fm):
            torch.testing.assert_close(f, torch.from_numpy(g[f"fm_{tag}_{i}"]), rtol=1e-4, atol=1e-4)
    ks = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    full = Mask2FormerForUniversalSegmentation(Mask2FormerConfig.from_dict(ks["swin_tiny_config"]))
    own, exp = full.state_dict(), ks["swin_tiny_q100_l3"]
    assert set(own) == set(exp) and all(list(own[k].shape) == exp[k] for k in exp)


def test_legacy_swin_checkpoint_names_load(tmp_path):
    """A transformers-4.x style checkpoint (attention.self.query ... and no `swin.` prefix) loads into the same tensors."""
    from safetensors.torch import save_file
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    ks = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    cd = dict(ks["swin_tiny_config"])
    cd["backbone_config"] = dict(cd["backbone_config"], embed_dim=16, depths=[1, 1, 1, 1], num_heads=[1, 2, 4, 4], window_size=4)
    cd.update(feature_size=32, mask_feature_size=32, hidden_dim=32, num_attention_heads=2, encoder_layers=1,
              decoder_layers=2, num_queries=5, dim_feedforward=32, encoder_feedforward_dim=32)
    cfg = Mask2FormerConfig.from_dict(cd)
    torch.manual_seed(0)
    m = Mask2FormerForUniversalSegmentation(cfg)
    pre = "model.pixel_level_module.encoder."
    back = (("attention.q_proj", "attention.self.query"), ("attention.k_proj", "attention.self.key"),
            ("attention.v_proj", "attention.self.value"),
            ("attention.relative_position_bias.relative_position_bias_table", "attention.self.relative_position_bias_table"),
            ("attention.o_proj", "attention.output.dense"), ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense"))
    old = {}
    for k, v in m.state_dict().items():
        if k.startswith(pre + "swin.layernorm."):
            continue
        if k.startswith(pre + "swin."):
            k = pre + k[len(pre) + 5:]
            for a, b in back:
                k = k.replace(a, b)
        old[k] = v.detach().clone().contiguous()
    assert any("attention.self.query" in k for k in old) and not any(".swin." in k for k in old)
    cfg.save_pretrained(str(tmp_path))
    save_file(old, os.path.join(str(tmp_path), "model.safetensors"))
    m2 = Mask2FormerForUniversalSegmentation.from_pretrained(str(tmp_path))
    a, b = m.state_dict(), m2.state_dict()
    for k in a:
        if ".swin.layernorm." not in k:
            assert torch.equal(a[k], b[k]), k


def test_swin_large_200_query_checkpoint_family_names():
    """The reference trains from `facebook/mask2former-swin-large-coco-instance` (config.py:4: Swin-L, window 12, 200
    queries, 80 classes).  Same 760 tensor names and shapes as the dependency builds for that configuration (fixture from a
    local config on the meta device: tests/golden/make_golden.py keys)."""
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    ks = json.load(open(os.path.join(ROOT, "tests", "golden", "state_keys.json")))
    cfg = Mask2FormerConfig.from_dict(ks["swin_large_config"])
    assert cfg.num_queries == 200 and cfg.backbone_config["window_size"] == 12 and cfg.num_labels == 80
    with torch.device("meta"):
        m = Mask2FormerForUniversalSegmentation(cfg)
    own, exp = m.state_dict(), ks["swin_large_q200_l80"]
    assert set(own) == set(exp) and len(exp) == 760
    assert all(list(own[k].shape) == exp[k] for k in exp)


def test_dependency_class_loads_a_directory_written_here(tmp_path):
    """save_pretrained of this package -> the installed transformers class' from_pretrained(local_dir) (what
    models/mask2former/train.py:245 and models/model_utils.py:14 call): every tensor arrives, bit for bit, and the
    dependency's own forward runs on it.  Runs in the build container only (the dependency is imported by a TEST)."""
    tr = pytest.importorskip("transformers")
    from conftest import load_golden
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    g = load_golden("full_tiny.npz")
    cfg = Mask2FormerConfig.from_dict(json.loads(str(g["config_json"])))
    m = Mask2FormerForUniversalSegmentation(cfg)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)
    m.save_pretrained(str(tmp_path))
    hf = tr.Mask2FormerForUniversalSegmentation.from_pretrained(str(tmp_path)).eval()
    theirs = hf.state_dict()
    assert set(theirs) == set(sd)
    for k, v in sd.items():
        assert torch.equal(theirs[k], v), k
    assert {int(k): v for k, v in hf.config.id2label.items()} == cfg.id2label
    with torch.no_grad():
        out = hf(pixel_values=torch.from_numpy(g["pixel_values"]))
    ref = torch.from_numpy(g["masks_queries_logits"])
    assert (out.masks_queries_logits - ref).abs().max().item() / ref.abs().max().item() < 1e-5  # the golden, reproduced


def test_dependency_processor_consumes_the_output_object():
    """models/metrics.py:58-63 / inference.py:30 hand the model's output object to the dependency's
    post_process_instance_segmentation, which reads `.class_queries_logits` / `.masks_queries_logits`
    (image_processing_pil_mask2former.py:714-716): the output class of this package is accepted as is and gives the
    fixture's result."""
    pytest.importorskip("transformers")
    from transformers.models.mask2former.image_processing_pil_mask2former import Mask2FormerImageProcessorPil
    from conftest import load_golden
    from weed_instance_segmentation_amd.modeling import Mask2FormerForUniversalSegmentationOutput
    g = load_golden("postprocess_instances.npz")
    info = json.loads(str(g["info_json"]))
    out = Mask2FormerForUniversalSegmentationOutput(class_queries_logits=torch.from_numpy(g["class_logits"]),
                                                    masks_queries_logits=torch.from_numpy(g["mask_logits"]))
    ts = [tuple(t) for t in info["mixed"]["target_sizes"]]
    res = Mask2FormerImageProcessorPil().post_process_instance_segmentation(out, threshold=0.5, mask_threshold=0.5, target_sizes=ts)
    for i, r in enumerate(res):
        assert torch.equal(r["segmentation"].to(torch.int16), torch.from_numpy(g[f"seg_mixed_{i}"]))
        assert [s["label_id"] for s in r["segments_info"]] == [s["label_id"] for s in info["mixed"]["segments_info"][i]]


def _lsa_transcription(cost):
    """The steps of csrc/lsa.hip (= scipy's rectangular_lsap.cpp) in plain Python: shortest augmenting paths in float64, `remaining`
    filled in reverse and compacted by swap-with-last, the tie rule (a free column wins among equal shortest-path costs), output
    sorted by row, fewer columns than rows solved transposed."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    transpose = nc < nr
    if transpose:
        cost = cost.T.copy()
        nr, nc = nc, nr
    u, v, spc = np.zeros(nr), np.zeros(nc), np.empty(nc)
    path, col4row, row4col = np.full(nc, -1), np.full(nr, -1), np.full(nc, -1)
    SR, SC, remaining = np.zeros(nr, bool), np.zeros(nc, bool), np.empty(nc, int)
    for cur in range(nr):
        min_val, i, num_remaining, sink = 0.0, cur, nc, -1
        remaining[:] = nc - 1 - np.arange(nc)
        SR[:] = False
        SC[:] = False
        spc[:] = np.inf
        while sink == -1:
            index, lowest = -1, np.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + cost[i, j] - u[i] - v[j]
                if r < spc[j]:
                    path[j], spc[j] = i, r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest, index = spc[j], it
            min_val = lowest
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for i2 in range(nr):
            if SR[i2] and i2 != cur:
                u[i2] += min_val - spc[col4row[i2]]
        for j2 in range(nc):
            if SC[j2]:
                v[j2] -= min_val - spc[j2]
        j = sink
        while True:
            i2 = path[j]
            row4col[j] = i2
            col4row[i2], j = j, col4row[i2]
            if i2 == cur:
                break
    if transpose:
        order = np.argsort(col4row, kind="stable")
        return col4row[order], order
    return np.arange(nr), col4row


def test_lsa_transcription_equals_scipy():
    """The algorithm the device solver implements (csrc/lsa.hip) IS scipy's: its Python transcription returns scipy's indices on
    random matrices and on tie-heavy integer matrices (where the tie rule and the scan order decide), both orientations."""
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(0)
    for trial in range(1200):
        nr, nc = rng.integers(1, 20), rng.integers(1, 20)
        kind = trial % 4
        if kind == 0:
            c = rng.standard_normal((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 2:
            c = rng.integers(0, 2, (nr, nc)).astype(float) * 5
        else:
            c = rng.standard_normal((nr, nc)).astype(np.float32).astype(float).round(1)
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = _lsa_transcription(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (trial, c)
    for _ in range(20):
        c = rng.standard_normal((100, 16)).astype(np.float32)
        r0, c0 = linear_sum_assignment(c)
        r1, c1 = _lsa_transcription(c)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1)


def test_k1_rows_op_host_logic():
    """The training K1 op has no CPU form: the shape test says so for host tensors (the module then composes the op from stock
    autograd ops, which is what runs on a CPU), and calling it anyway raises instead of computing."""
    from weed_instance_segmentation_amd import ops
    shapes = [(2, 2), (4, 4), (8, 8)]
    S = sum(h * w for h, w in shapes)
    value, rows = torch.randn(1, S, 8, 32), torch.randn(1, S, 8 * 36)
    assert not ops.k1_rows_applies(value, rows, shapes, 8)
    with pytest.raises(ValueError):
        ops.ms_deform_attn_rows(value, shapes, rows, 8)
    assert ops.k1_lanes_applies(shapes, S, 32, 4, 1, 8) and not ops.k1_lanes_applies([(2, 2), (4, 4), (8, 9)], 4 + 16 + 72, 32, 4, 1, 8)


def test_uncertain_points_host_route_is_the_dependency_topk():
    """loss._uncertain_points on host tensors: HF:688-704's topk + gather, in the first n_unc slots of the (rows, P, 2) result."""
    from weed_instance_segmentation_amd.loss import Mask2FormerLoss
    g = torch.Generator().manual_seed(3)
    unc, pc = -torch.rand(3, 40, generator=g), torch.rand(3, 40, 2, generator=g)
    pts = Mask2FormerLoss._uncertain_points(unc, pc, 10, 16)
    assert pts.shape == (3, 16, 2)
    idx = torch.topk(unc, 10, dim=1)[1]
    assert torch.equal(pts[:, :10], torch.gather(pc, 1, idx[..., None].expand(-1, -1, 2)))
