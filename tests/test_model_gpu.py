"""Whole-module parity on the GPU: the drop-in module (HIP kernels) vs the golden fixture produced by
transformers 5.15.0 and vs the CPU oracle on the same weights.  Needs an MI355X (-m gpu)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m2f_oracle as O

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def tiny():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    g = load_golden("full_tiny.npz")
    cfg = Mask2FormerConfig.from_dict(json.loads(str(g["config_json"])))
    model = Mask2FormerForUniversalSegmentation(cfg)
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    return g, cfg, model.cuda().eval(), sd


def _labels(g, B, dev="cuda", dtype=torch.float32):
    ml = [T(g[f"mask_labels_{i}"]).to(dtype).to(dev) for i in range(B)]
    cl = [T(g[f"class_labels_{i}"]).to(dev) for i in range(B)]
    return ml, cl


def test_forward_matches_golden(tiny):
    g, cfg, model, _ = tiny
    with torch.no_grad():
        out = model(pixel_values=T(g["pixel_values"]).cuda(), output_hidden_states=True, output_auxiliary_logits=True)
    for i, f in enumerate(out.encoder_hidden_states):
        torch.testing.assert_close(f.cpu(), T(g[f"backbone_{i}"]), rtol=1e-4, atol=1e-4)
    for i, f in enumerate(out.pixel_decoder_hidden_states):
        torch.testing.assert_close(f.cpu(), T(g[f"multi_scale_{i}"]), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(out.pixel_decoder_last_hidden_state.cpu(), T(g["mask_features"]), rtol=1e-3, atol=2e-4)
    # headline bar: mask logits within 1e-3 relative (fp32)
    ref = T(g["masks_queries_logits"])
    err = (out.masks_queries_logits.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-3, err
    torch.testing.assert_close(out.class_queries_logits.cpu(), T(g["class_queries_logits"]), rtol=1e-3, atol=1e-3)
    for i, aux in enumerate(out.auxiliary_logits):
        r = T(g[f"aux_masks_{i}"])
        assert (aux["masks_queries_logits"].cpu() - r).abs().max().item() / r.abs().max().item() < 1e-3


@pytest.mark.parametrize("label_dtype", [torch.float32, torch.uint8])
@pytest.mark.parametrize("batched", [True, False])
def test_loss_and_indices_match_golden(tiny, label_dtype, batched):
    """Both loss paths: all levels in one pass (loss_masks_all_levels, the default) and one pass per level."""
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    g, cfg, model, _ = tiny
    model.criterion.batched_levels = batched
    B = g["pixel_values"].shape[0]
    ml, cl = _labels(g, B, dtype=label_dtype)
    n_layers = cfg.decoder_layers - 1
    draws = [T(g[f"draw_{i}"]) for i in range(int(g["n_draws"]))][n_layers:]
    prov = ReplayPointProvider(draws, cfg.decoder_layers, B, "cuda")
    with torch.no_grad():
        out = model(pixel_values=T(g["pixel_values"]).cuda(), mask_labels=ml, class_labels=cl, point_provider=prov)
    for i, (r, c) in enumerate(out.matched_indices):  # bit-exact assignment
        assert np.array_equal(r.numpy(), g[f"row_{i}"]) and np.array_equal(c.numpy(), g[f"col_{i}"])
    for k, v in out.loss_dict.items():
        torch.testing.assert_close(v.cpu(), T(g["ld." + k]), rtol=2e-3, atol=1e-4)
    torch.testing.assert_close(out.loss.cpu(), T(g["loss"]), rtol=1e-3, atol=1e-3)
    model.criterion.batched_levels = True


def test_forward_matches_oracle_other_input(tiny):
    """Different input than the fixture: the oracle is the checker (same weights)."""
    g, cfg, model, sd = tiny
    x = torch.randn(3, 3, 96, 64, generator=torch.Generator().manual_seed(5))
    res = O.forward(sd, json.loads(str(g["config_json"])), x)
    with torch.no_grad():
        out = model(pixel_values=x.cuda())
    ref = res["masks_queries_logits"]
    assert (out.masks_queries_logits.cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-3
    torch.testing.assert_close(out.class_queries_logits.cpu(), res["class_queries_logits"], rtol=1e-3, atol=1e-3)


def test_train_step_backward_runs(tiny):
    """Gradients flow through K1, K2 (HIP backward kernels), K3 and the point sampler."""
    g, cfg, model, _ = tiny
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    m = Mask2FormerForUniversalSegmentation(cfg).cuda().train()
    B = g["pixel_values"].shape[0]
    ml, cl = _labels(g, B)
    out = m(pixel_values=T(g["pixel_values"]).cuda(), mask_labels=ml, class_labels=cl)
    out.loss.backward()
    grads = [p.grad for p in m.parameters() if p.grad is not None]
    assert len(grads) > 100 and all(torch.isfinite(gr).all() for gr in grads)


def test_matched_row_mask_loss_equals_dense_route(tiny):
    """The mask losses read the matched queries' logits from ONE recomputed einsum (loss.matched_row_logits) instead of the
    dense per-level predictions: same loss dictionary and the same parameter gradients as the dense route (criterion.
    matched_row_masks = False) on the same weights, labels and recorded points -- fp32, and under bf16 autocast."""
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    g, cfg, _, sd = tiny
    B = g["pixel_values"].shape[0]
    n_layers = cfg.decoder_layers - 1
    draws = [T(g[f"draw_{i}"]) for i in range(int(g["n_draws"]))][n_layers:]
    mlg, clg = _labels(g, B)
    x = T(g["pixel_values"]).cuda()
    for amp in (False, True):
        res = {}
        for route in (True, False):
            m = Mask2FormerForUniversalSegmentation(cfg)
            m.load_state_dict(sd, strict=True)
            m = m.cuda().eval()
            m.criterion.matched_row_masks = route
            prov = ReplayPointProvider(draws, cfg.decoder_layers, B, "cuda")
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = m(pixel_values=x, mask_labels=mlg, class_labels=clg, point_provider=prov)
            out.loss.backward()
            res[route] = ({k: v.detach().float().cpu() for k, v in out.loss_dict.items()},
                          {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None})
        (la, ga), (lb, gb) = res[True], res[False]
        assert la.keys() == lb.keys() and ga.keys() == gb.keys()
        for k in la:
            torch.testing.assert_close(la[k], lb[k], rtol=(2e-2 if amp else 1e-5), atol=(2e-2 if amp else 1e-6))
        top = max(float(v.abs().max()) for v in gb.values())
        for n in ga:  # parameters whose gradient is mathematically zero (a bias in front of a normalisation) hold rounding noise only
            scale = max(float(gb[n].abs().max()), 1e-4 * top)
            assert float((ga[n] - gb[n]).abs().max()) <= (6e-2 if amp else 2e-4) * scale, n


def test_parameter_gradients_match_oracle(tiny):
    """d loss / d theta through every backward kernel (K1 atomics, K2 flash backward, K3 GEMMs, point
    sampler) against the oracle's CPU autograd on the same weights, labels and recorded points."""
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    g, cfg, _, sd = tiny
    cfgd = json.loads(str(g["config_json"]))
    B = g["pixel_values"].shape[0]
    n_layers = cfg.decoder_layers - 1
    draws = [T(g[f"draw_{i}"]) for i in range(int(g["n_draws"]))][n_layers:]
    names = ["class_predictor.weight", "model.transformer_module.decoder.layers.0.cross_attn.in_proj_weight",
             "model.transformer_module.decoder.layers.1.cross_attn.out_proj.weight",
             "model.transformer_module.decoder.mask_predictor.mask_embedder.0.0.weight",
             "model.pixel_level_module.decoder.encoder.layers.0.self_attn.sampling_offsets.weight",
             "model.pixel_level_module.decoder.encoder.layers.1.self_attn.value_proj.weight",
             "model.pixel_level_module.decoder.encoder.layers.0.self_attn.attention_weights.bias",
             "model.pixel_level_module.decoder.mask_projection.weight",
             "model.transformer_module.queries_features.weight"]
    # oracle side (eval-mode BatchNorm on both sides)
    sdo = {k: (v.clone().requires_grad_() if k in names else v) for k, v in sd.items()}
    ml, cl = _labels(g, B, dev="cpu")
    res = O.forward(sdo, cfgd, T(g["pixel_values"]), ml, cl, O.RandSource(draws), grad=True)
    res["loss"].backward()
    # product side
    m = Mask2FormerForUniversalSegmentation(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()  # eval: running-stat BatchNorm, no dropout (dropout is 0 anyway); grads still flow
    mlg, clg = _labels(g, B)
    prov = ReplayPointProvider(draws, cfg.decoder_layers, B, "cuda")
    out = m(pixel_values=T(g["pixel_values"]).cuda(), mask_labels=mlg, class_labels=clg, point_provider=prov)
    torch.testing.assert_close(out.loss.detach().cpu(), res["loss"].detach(), rtol=1e-3, atol=1e-3)
    out.loss.backward()
    params = dict(m.named_parameters())
    for n in names:
        a, b = params[n].grad.cpu(), sdo[n].grad
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / scale
        assert err < 2e-3, f"{n}: rel err {err:.3e}"


def test_forward_dim256_fused_inference_paths():
    """feature_size 256 / 8 heads (head_dim 32): the module takes the LDS-window K1 with the merged
    projection, the fused residual+LayerNorm, the folded-BatchNorm epilogues -- checked against the oracle
    on the same (random) weights, under no_grad (fused paths) AND with grad enabled (plain paths)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    bc = dict(model_type="resnet", num_channels=3, embedding_size=16, hidden_sizes=[32, 64, 128, 256], depths=[1, 2, 1, 1],
              layer_type="bottleneck", out_features=["stage1", "stage2", "stage3", "stage4"])
    cfg = Mask2FormerConfig(backbone_config=bc, num_labels=3, num_queries=20, encoder_layers=2, decoder_layers=4,
                            dim_feedforward=128, encoder_feedforward_dim=256)
    torch.manual_seed(3)
    m = Mask2FormerForUniversalSegmentation(cfg)
    with torch.no_grad():  # make every parameter and statistic count
        g = torch.Generator().manual_seed(4)
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.03)
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.add_(torch.randn(b.shape, generator=g) * 0.1)
            elif n.endswith("running_var"):
                b.mul_(1 + 0.3 * torch.rand(b.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(5))
    ref = O.forward(sd, cfg.to_dict(), x)["masks_queries_logits"]
    m = m.cuda().eval()
    with torch.no_grad():
        out_ng = m(pixel_values=x.cuda()).masks_queries_logits.cpu()
    out_g = m(pixel_values=x.cuda()).masks_queries_logits.detach().cpu()
    for out in (out_ng, out_g):
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-3, err


def test_low_resolution_attention_masks_equal_full_resolution_route(tiny):
    """Inference without auxiliary outputs builds the attention masks from mask features resized once per level
    (resize(einsum(E, P)) == einsum(E, resize(P))): same final predictions as the full-resolution route and as the oracle.
    128 x 128 input: level widths 4, 8, 16 (the route needs widths divisible by 4; the 64 x 96 golden input has 3, 6, 12)."""
    from weed_instance_segmentation_amd import ops as wops
    g, cfg, model, sd = tiny
    dec = model.model.transformer_module.decoder
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(31))
    ref = O.forward(sd, cfg.to_dict(), x)
    outs, launches = {}, {}
    try:
        for low in (True, False):
            dec.low_res_masks = low
            timer = wops.KernelTimer()
            wops.set_kernel_timer(timer)
            with torch.no_grad():
                o = model(pixel_values=x.cuda())
            torch.cuda.synchronize()
            wops.set_kernel_timer(None)
            launches[low] = {k: n for k, (n, _) in timer.summary().items() if k.startswith("mask_einsum")}
            assert o.auxiliary_logits is None
            outs[low] = (o.masks_queries_logits.cpu(), o.class_queries_logits.cpu())
    finally:
        dec.low_res_masks = True
        wops.set_kernel_timer(None)
    n_pred = len(dec.layers) + 1
    assert launches[False] == {"mask_einsum_fwd": n_pred}, launches  # every prediction at the mask-feature resolution
    assert launches[True]["mask_einsum_fwd"] == 1 and sum(launches[True].values()) == n_pred, launches  # one full, the rest per level
    assert all(k.startswith("mask_einsum_attn_mask_hw") for k in launches[True] if k != "mask_einsum_fwd"), launches  # fused epilogue
    rm = ref["masks_queries_logits"]
    for low in (True, False):
        assert (outs[low][0] - rm).abs().max().item() / rm.abs().max().item() < 1e-3
        torch.testing.assert_close(outs[low][1], ref["class_queries_logits"], rtol=1e-3, atol=1e-3)
    assert (outs[True][0] - outs[False][0]).abs().max().item() / rm.abs().max().item() < 1e-3
    # with auxiliary outputs requested every prediction is computed at full resolution again
    with torch.no_grad():
        o = model(pixel_values=x.cuda(), output_auxiliary_logits=True)
    assert all(a["masks_queries_logits"] is not None for a in o.auxiliary_logits)


def test_bf16_autocast_forward_and_train_step(tiny):
    """BASELINE configs 3-5 run under bf16 autocast.  Stock ops follow PyTorch's autocast policy; the wm2f kernels take
    the fp32 policy (inputs cast to fp32, as grid_sample / softmax get in the dependency).  Checked against the fp32
    run and against the CPU oracle executed under the same autocast (bf16 rounding differs per implementation: the
    bounds are a few bf16 ulps of the logit range)."""
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    g, cfg, model, sd = tiny
    x = T(g["pixel_values"]).cuda()
    with torch.no_grad():
        ref32 = model(pixel_values=x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pixel_values=x)
    scale = ref32.masks_queries_logits.abs().max().item()
    err = (out.masks_queries_logits.float() - ref32.masks_queries_logits).abs().max().item() / scale
    assert err < 6e-2, err
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        o = O.forward(sd, cfg.to_dict(), T(g["pixel_values"]))
    # the reference arithmetic under the same autocast is itself this far from fp32; ours must not be further
    err_o = (o["masks_queries_logits"].float() - ref32.masks_queries_logits.cpu()).abs().max().item() / scale
    assert err < 1.5 * err_o + 1e-2, (err, err_o)
    # train step under autocast: finite loss close to the fp32 loss, gradients reach the first backbone layer
    B = x.shape[0]
    ml, cl = _labels(g, B)
    n_layers = cfg.decoder_layers - 1
    draws = [T(g[f"draw_{i}"]) for i in range(int(g["n_draws"]))][n_layers:]
    model.train()
    try:
        losses = []
        for amp in (False, True):
            model.zero_grad(set_to_none=True)
            prov = ReplayPointProvider(draws, cfg.decoder_layers, B, "cuda")
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                res = model(pixel_values=x, mask_labels=ml, class_labels=cl, point_provider=prov)
            res.loss.backward()
            losses.append(res.loss.item())
            gsum = sum(p.grad.abs().sum().item() for p in model.parameters() if p.grad is not None)
            assert np.isfinite(gsum) and gsum > 0
        assert abs(losses[1] - losses[0]) / abs(losses[0]) < 5e-2, losses
    finally:
        model.eval()
        model.zero_grad(set_to_none=True)


@pytest.mark.parametrize("size", [(200, 333), (96, 160)])
def test_forward_odd_sizes_match_oracle(tiny, size):
    """Input sizes that are not multiples of 32 (BASELINE config 5 is 1333 x 800): the three levels are then not in
    the exact 1 : 2 : 4 ratio, so K1 falls back from the streaming kernel to the generic LDS-window / direct kernels;
    ragged tiles everywhere.  (96, 160) is the aligned control."""
    g, cfg, model, sd = tiny
    x = torch.randn(2, 3, *size, generator=torch.Generator().manual_seed(9))
    res = O.forward(sd, json.loads(str(g["config_json"])), x)
    with torch.no_grad():
        out = model(pixel_values=x.cuda())
    ref = res["masks_queries_logits"]
    assert out.masks_queries_logits.shape == ref.shape
    assert (out.masks_queries_logits.cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-3
    torch.testing.assert_close(out.class_queries_logits.cpu(), res["class_queries_logits"], rtol=1e-3, atol=1e-3)


def test_swin_backbone_model_matches_oracle():
    """BASELINE configs 4 / 5 use Swin backbones: a small Swin + the wm2f pixel decoder / transformer decoder on the GPU
    against the oracle fed with the same backbone features (the Swin backbone itself is pinned on CPU against the
    dependency's SwinBackbone fixture in test_host_cpu.py)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    g = load_golden("full_tiny.npz")
    cd = json.loads(str(g["config_json"]))
    cd["backbone_config"] = {"model_type": "swin", "embed_dim": 16, "depths": [1, 1, 2, 1], "num_heads": [1, 2, 4, 4],
                             "window_size": 4, "mlp_ratio": 2.0, "patch_size": 4, "num_channels": 3,
                             "out_features": ["stage1", "stage2", "stage3", "stage4"], "drop_path_rate": 0.0}
    cfg = Mask2FormerConfig.from_dict(cd)
    torch.manual_seed(3)
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    x = torch.randn(2, 3, 128, 160, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        feats = model.model.pixel_level_module.encoder(x)  # CPU, stock ops
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = O.forward(sd, cfg.to_dict(), x, backbone_feats=[f.clone() for f in feats])
    model = model.cuda()
    with torch.no_grad():
        out = model(pixel_values=x.cuda())
    ref = res["masks_queries_logits"]
    assert (out.masks_queries_logits.cpu() - ref).abs().max().item() / ref.abs().max().item() < 1e-3
    torch.testing.assert_close(out.class_queries_logits.cpu(), res["class_queries_logits"], rtol=1e-3, atol=1e-3)


def test_graphed_forward_equals_eager(tiny):
    """HIP-graph capture of the label-free forward: same numbers as the eager call, also after the input changes."""
    from weed_instance_segmentation_amd.graph import GraphedForward
    g, cfg, model, _ = tiny
    x1 = T(g["pixel_values"]).cuda()
    x2 = torch.randn_like(x1)
    fwd = GraphedForward(model, x1)
    for x in (x1, x2, x1):
        out = fwd(x)
        with torch.no_grad():
            ref = model(pixel_values=x)
        scale = ref.masks_queries_logits.abs().max().item()
        assert (out.masks_queries_logits - ref.masks_queries_logits).abs().max().item() / scale < 1e-5
        torch.testing.assert_close(out.class_queries_logits, ref.class_queries_logits, rtol=1e-4, atol=1e-4)


def test_many_queries_forward_and_loss_match_oracle():
    """BASELINE config 5 uses 200 queries (more than one 112-query pass of K2 / K3-bf16, 13 query tiles in K3): forward
    and loss of a small random-init model against the oracle with the same weights and the same recorded points."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    g = load_golden("full_tiny.npz")
    cd = json.loads(str(g["config_json"]))
    cd["num_queries"] = 200
    cfg = Mask2FormerConfig.from_dict(cd)
    torch.manual_seed(11)
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 96, 128, generator=gen)
    ml = [(torch.rand(3, 96, 128, generator=gen) < 0.3).float(), (torch.rand(5, 96, 128, generator=gen) < 0.2).float()]
    cl = [torch.randint(0, cfg.num_labels, (3,), generator=gen), torch.randint(0, cfg.num_labels, (5,), generator=gen)]
    class Recorder(O.RandSource):  # draws in the criterion's order: per level (final first) B matcher, 1 oversample, 1 random
        def __init__(self):
            super().__init__()
            self.rec, self.gen = [], torch.Generator().manual_seed(5)

        def rand(self, *shape):
            d = torch.rand(*shape, generator=self.gen)
            self.rec.append(d)
            return d

    rs = Recorder()
    res = O.forward(sd, cfg.to_dict(), x, ml, cl, rand_source=rs)
    model = model.cuda()
    prov = ReplayPointProvider(rs.rec, cfg.decoder_layers, 2, "cuda")
    with torch.no_grad():
        out = model(pixel_values=x.cuda(), mask_labels=[m.cuda() for m in ml], class_labels=[c.cuda() for c in cl],
                    point_provider=prov, output_auxiliary_logits=True)
    # With random-init weights many mask logits sit at the 0.5-probability threshold of the attention mask (HF:2053): a
    # 1e-6 difference can flip a mask bit and change THAT query from the next layer on.  So: the first prediction levels
    # must agree everywhere, the final one for (nearly) all queries, the loss to a tolerance that admits a flip.
    ref = res["masks_queries_logits"]
    scale = ref.abs().max().item()
    for i in range(2):
        a_ = out.auxiliary_logits[i]["masks_queries_logits"].cpu()
        assert (a_ - res["aux_masks"][i]).abs().max().item() / scale < 1e-4
    per_query = (out.masks_queries_logits.cpu() - ref).abs().amax(dim=(0, 2, 3)) / scale
    assert (per_query < 1e-4).float().mean().item() >= 0.97, per_query.max()
    torch.testing.assert_close(out.loss.cpu(), res["loss"], rtol=2e-2, atol=2e-2)
