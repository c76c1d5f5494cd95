#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the installed third-party
implementation of the hot path (transformers==5.15.0, torch CPU).

Runs ONLY in the build container (the GPU box never imports transformers'
mask2former module through this repo).  The reference repo pins no version of
`transformers`; the version used is recorded in every fixture (`hf_version`).

What is captured:
  k1_msdeform_*.npz      multi_scale_deformable_attention            HF:798-837
  a2_msdeform_module.npz MSDeformAttn module incl. projections       HF:954-1014
  k2_masked_xattn.npz    nn.MultiheadAttention masked cross-attn      HF:1618,1644-1650,1912-1914
  k3_mask_predictor.npz  mask predictor einsum + attention-mask build HF:2040-2056
  k4_matcher.npz         Hungarian matcher cost matrix + indices      HF:413-481
  full_tiny.npz          whole forward + loss of a reduced model      HF:2332-2530
  state_keys_r50.json    parameter names/shapes of the R50 / Swin-T configs

The global-RNG draws (`torch.rand`) made by the matcher, the loss and the
decoder are recorded in call order by wrapping `torch.rand` in this process
only; no installed file is edited.

Usage:  HF_HUB_OFFLINE=1 TRANSFORMERS_OFFLINE=1 python tests/golden/make_golden.py
"""
import json
import os
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")

import numpy as np
import torch
import transformers
from transformers import Mask2FormerConfig, Mask2FormerForUniversalSegmentation, ResNetConfig
from transformers.models.mask2former import modeling_mask2former as hf

HERE = os.path.dirname(os.path.abspath(__file__))
META = dict(hf_version=transformers.__version__, torch_version=torch.__version__)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["hf_version"] = np.asarray(META["hf_version"])
    out["torch_version"] = np.asarray(META["torch_version"])
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


class RandRecorder:
    """Wrap torch.rand so every draw is logged (shape + values) in call order."""

    def __init__(self):
        self.draws = []
        self._orig = torch.rand

    def __enter__(self):
        def rand(*a, **kw):
            r = self._orig(*a, **kw)
            self.draws.append(r.detach().clone().cpu())
            return r

        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


# --------------------------------------------------------------------------- K1
def gen_k1():
    g = torch.Generator().manual_seed(11)
    for tag, shapes, B, Q, H, D in [
        ("toy", [(2, 2), (4, 4), (8, 8)], 2, 84, 8, 32),
        ("rect", [(3, 5), (6, 10), (12, 20)], 2, 3 * 5 + 6 * 10 + 12 * 20, 8, 32),
        ("d8", [(2, 2), (4, 4), (8, 8)], 1, 20, 2, 8),
    ]:
        S = sum(h * w for h, w in shapes)
        L, P = len(shapes), 4
        value = torch.randn(B, S, H, D, generator=g)
        # locations: mostly inside [0,1], some well outside to pin zero padding,
        # plus exact border / pixel-centre values.
        loc = torch.rand(B, Q, H, L, P, 2, generator=g) * 1.3 - 0.15
        loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0])
        loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0])
        loc[0, 0, 0, 1, 0] = torch.tensor([-0.5, 0.5])
        loc[0, 0, 0, 1, 1] = torch.tensor([0.5, 1.5])
        loc[0, 1, 0, 2, 0] = torch.tensor([0.0625, 0.0625])  # pixel centre of an 8x8 map
        loc[0, 1, 0, 2, 1] = torch.tensor([5.0, -3.0])
        w = torch.softmax(torch.randn(B, Q, H, L * P, generator=g), -1).view(B, Q, H, L, P)
        out = hf.multi_scale_deformable_attention(value, shapes, loc, w)
        save(f"k1_msdeform_{tag}.npz", value=value, level_hw=np.asarray(shapes, np.int32), loc=loc, w=w, out=out)


# --------------------------------------------------------------------------- a2
def gen_a2():
    torch.manual_seed(12)
    g = torch.Generator().manual_seed(12)
    shapes = [(2, 3), (4, 6), (8, 12)]
    S = sum(h * w for h, w in shapes)
    B, dm, H = 2, 64, 2
    mod = hf.Mask2FormerPixelDecoderEncoderMultiscaleDeformableAttention(dm, H, 3, 4).eval()
    for p in mod.parameters():
        torch.nn.init.normal_(p, std=0.3, generator=g)
    hidden = torch.randn(B, S, dm, generator=g)
    pos = torch.randn(B, S, dm, generator=g)
    valid = torch.ones(B, 3, 2)
    ref = hf.Mask2FormerPixelDecoderEncoderOnly.get_reference_points(shapes, valid, "cpu")
    with torch.no_grad():
        out, attn = mod(
            hidden_states=hidden,
            attention_mask=torch.zeros(B, S, dtype=torch.bool),
            encoder_hidden_states=hidden,
            position_embeddings=pos,
            reference_points=ref,
            spatial_shapes_list=shapes,
        )
    sd = {"sd." + k: v for k, v in mod.state_dict().items()}
    save("a2_msdeform_module.npz", hidden=hidden, pos=pos, ref=ref, level_hw=np.asarray(shapes, np.int32),
         out=out, attn=attn, n_heads=np.asarray(H), **sd)


# --------------------------------------------------------------------------- K2
def gen_k2():
    g = torch.Generator().manual_seed(13)
    E, H = 256, 8
    for tag, Q, HW, B in [("small", 10, 48, 2), ("q100", 100, 320, 2)]:
        mha = torch.nn.MultiheadAttention(E, H, 0.0).eval()
        for p in mha.parameters():
            torch.nn.init.normal_(p, std=0.08, generator=g)
        query = torch.randn(Q, B, E, generator=g)  # hidden + query_pos
        key = torch.randn(HW, B, E, generator=g)  # feat + pos
        value = torch.randn(HW, B, E, generator=g)  # feat
        # un-replicated mask, True = blocked (HF:2053); rows 0 and Q-1 of batch 0 fully blocked
        mask = torch.rand(B, Q, HW, generator=g) < 0.6
        mask[0, 0, :] = True
        mask[0, Q - 1, :] = True
        mask[1, 2, :] = False
        mask[1, 3, :-1] = True  # a single open key
        attn_mask = mask[:, None].repeat(1, H, 1, 1).flatten(0, 1)  # (B*H, Q, HW)  HF:2052-2053
        # the decoder's fix-up for fully masked rows                  HF:1912-1914
        where = (attn_mask.sum(-1) != attn_mask.shape[-1]).to(attn_mask.dtype)
        attn_mask_fixed = attn_mask * where.unsqueeze(-1)
        with torch.no_grad():
            out, _ = mha(query=query, key=key, value=value, attn_mask=attn_mask_fixed, key_padding_mask=None)
            # intermediate (pre out_proj) context, for testing the bare kernel
            Wq, Wk, Wv = mha.in_proj_weight.chunk(3)
            bq, bk, bv = mha.in_proj_bias.chunk(3)
            q = torch.nn.functional.linear(query, Wq, bq)
            k = torch.nn.functional.linear(key, Wk, bk)
            v = torch.nn.functional.linear(value, Wv, bv)
        save(f"k2_masked_xattn_{tag}.npz", query=query, key=key, value=value, mask=mask, out=out,
             q_proj=q, k_proj=k, v_proj=v, n_heads=np.asarray(H),
             in_proj_weight=mha.in_proj_weight, in_proj_bias=mha.in_proj_bias,
             out_proj_weight=mha.out_proj.weight, out_proj_bias=mha.out_proj.bias)


# --------------------------------------------------------------------------- K3
def gen_k3():
    g = torch.Generator().manual_seed(14)
    B, Q, C, Hh, Ww = 2, 100, 256, 16, 24
    mp = hf.Mask2FormerMaskPredictor(hidden_size=C, num_heads=8, mask_feature_size=C).eval()
    for p in mp.parameters():
        torch.nn.init.normal_(p, std=0.08, generator=g)
    outputs = torch.randn(Q, B, C, generator=g)  # (Q, B, C) layer-normed decoder state
    pix = torch.randn(B, C, Hh, Ww, generator=g)
    arrays = dict(outputs=outputs, pix=pix)
    with torch.no_grad():
        emb = mp.mask_embedder(outputs.transpose(0, 1))
        arrays["mask_embeddings"] = emb
        for i, size in enumerate([(4, 6), (8, 12), (2, 3), (16, 24), (5, 7)]):
            logits, attn = mp(outputs, pix, size)
            arrays[f"size_{i}"] = np.asarray(size, np.int32)
            # un-replicate heads: every head holds the same mask (HF:2052)
            attn = attn.view(B, 8, Q, -1)
            assert bool((attn == attn[:, :1]).all())
            arrays[f"attn_mask_{i}"] = attn[:, 0]
        arrays["logits"] = logits
    sd = {"sd." + k: v for k, v in mp.state_dict().items()}
    save("k3_mask_predictor.npz", **arrays, **sd)


# --------------------------------------------------------------------------- K4
def gen_k4():
    g = torch.Generator().manual_seed(15)
    B, Q, Hm, Wm, Ht, Wt, P, NL = 3, 100, 32, 32, 128, 128, 12544, 3
    matcher = hf.Mask2FormerHungarianMatcher(cost_class=2.0, cost_mask=5.0, cost_dice=5.0, num_points=P)
    masks = torch.randn(B, Q, Hm, Wm, generator=g) * 3
    cls = torch.randn(B, Q, NL + 1, generator=g)
    Ts = [16, 1, 5]
    mask_labels, class_labels = [], []
    for T in Ts:
        m = torch.zeros(T, Ht, Wt)
        for t in range(T):
            y0, x0 = [int(v) for v in torch.randint(0, Ht - 40, (2,), generator=g)]
            hh, ww = [int(v) for v in torch.randint(8, 40, (2,), generator=g)]
            m[t, y0:y0 + hh, x0:x0 + ww] = 1.0
        mask_labels.append(m)
        class_labels.append(torch.randint(0, NL, (T,), generator=g))
    torch.manual_seed(123)
    with RandRecorder() as rr:
        idx = matcher(masks, cls, mask_labels, class_labels)
    points = torch.stack([d[0] for d in rr.draws])  # (B, P, 2)
    arrays = dict(mask_logits=masks, class_logits=cls, points=points,
                  weights=np.asarray([2.0, 5.0, 5.0], np.float32))  # class, mask, dice
    # rebuild the cost matrix with the same points (HF:444-472)
    for i in range(B):
        pc = points[i:i + 1]
        pred_probs = cls[i].softmax(-1)
        cost_class = -pred_probs[:, class_labels[i]]
        tm = hf.sample_point(mask_labels[i][:, None], pc.repeat(Ts[i], 1, 1), align_corners=False).squeeze(1)
        pm = hf.sample_point(masks[i][:, None], pc.repeat(Q, 1, 1), align_corners=False).squeeze(1)
        cm = hf.pair_wise_sigmoid_cross_entropy_loss(pm, tm)
        cd = hf.pair_wise_dice_loss(pm, tm)
        cost = 5.0 * cm + 2.0 * cost_class + 5.0 * cd
        cost = torch.minimum(cost, torch.tensor(1e10))
        cost = torch.maximum(cost, torch.tensor(-1e10))
        cost = torch.nan_to_num(cost, 0)
        from scipy.optimize import linear_sum_assignment
        r, c = linear_sum_assignment(cost)
        assert np.array_equal(r, idx[i][0].numpy()) and np.array_equal(c, idx[i][1].numpy())
        arrays[f"cost_{i}"] = cost
        arrays[f"row_{i}"] = idx[i][0]
        arrays[f"col_{i}"] = idx[i][1]
        arrays[f"class_labels_{i}"] = class_labels[i]
        # targets are rectangles: store as uint8 to keep the fixture small
        arrays[f"mask_labels_{i}"] = mask_labels[i].to(torch.uint8)
    save("k4_matcher.npz", **arrays)


# --------------------------------------------------------------------------- full forward
def tiny_config():
    bc = ResNetConfig(embedding_size=16, hidden_sizes=[16, 32, 64, 128], depths=[1, 2, 1, 1],
                      layer_type="bottleneck", out_features=["stage1", "stage2", "stage3", "stage4"])
    return Mask2FormerConfig(backbone_config=bc, num_labels=3, num_queries=10, feature_size=64,
                             mask_feature_size=64, hidden_dim=64, encoder_feedforward_dim=128,
                             dim_feedforward=96, encoder_layers=2, decoder_layers=4,
                             num_attention_heads=2, train_num_points=160)


def randomise(model, seed):
    """Random-init leaves several tensors constant (zeros / ones); perturb everything so each
    parameter and buffer takes part in the result."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.add_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.mul_(1 + 0.3 * torch.rand(b.shape, generator=g))


def gen_full():
    torch.manual_seed(16)
    cfg = tiny_config()
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    randomise(model, 16)
    g = torch.Generator().manual_seed(17)
    B, Himg, Wimg = 2, 64, 96
    pixel_values = torch.randn(B, 3, Himg, Wimg, generator=g)
    Ts = [3, 1]
    mask_labels, class_labels = [], []
    for T in Ts:
        m = torch.zeros(T, Himg, Wimg)
        for t in range(T):
            y0, x0 = [int(v) for v in torch.randint(0, Himg - 24, (2,), generator=g)]
            hh, ww = [int(v) for v in torch.randint(6, 24, (2,), generator=g)]
            m[t, y0:y0 + hh, x0:x0 + ww] = 1.0
        mask_labels.append(m)
        class_labels.append(torch.randint(0, 3, (T,), generator=g))

    arrays = dict(pixel_values=pixel_values, config_json=np.asarray(json.dumps(cfg.to_dict(), default=str)))
    for i in range(B):
        arrays[f"mask_labels_{i}"] = mask_labels[i].to(torch.uint8)
        arrays[f"class_labels_{i}"] = class_labels[i]

    torch.manual_seed(321)
    with RandRecorder() as rr, torch.no_grad():
        out = model(pixel_values=pixel_values, mask_labels=mask_labels, class_labels=class_labels,
                    output_auxiliary_logits=True, output_hidden_states=True)
        # matched indices per level, replaying the same draws
    draws = rr.draws
    arrays["n_draws"] = np.asarray(len(draws))
    for i, d in enumerate(draws):
        arrays[f"draw_{i}"] = d
    arrays["loss"] = out.loss
    arrays["masks_queries_logits"] = out.masks_queries_logits
    arrays["class_queries_logits"] = out.class_queries_logits
    for i, aux in enumerate(out.auxiliary_logits):
        arrays[f"aux_masks_{i}"] = aux["masks_queries_logits"]
        arrays[f"aux_class_{i}"] = aux["class_queries_logits"]
    arrays["mask_features"] = out.pixel_decoder_last_hidden_state
    for i, f in enumerate(out.encoder_hidden_states):
        arrays[f"backbone_{i}"] = f
    for i, f in enumerate(out.pixel_decoder_hidden_states):
        arrays[f"multi_scale_{i}"] = f

    # loss dict + indices of the final level: rerun the criterion with the recorded draws replayed
    class Replay:
        def __init__(self, draws):
            self.draws, self.i, self._orig = draws, 0, torch.rand

        def __enter__(self):
            def rand(*a, **kw):
                r = self.draws[self.i]
                self.i += 1
                return r.clone()

            torch.rand = rand
            return self

        def __exit__(self, *exc):
            torch.rand = self._orig

    n_layers = cfg.decoder_layers - 1
    crit_draws = draws[n_layers:]  # the decoder draws one scalar per layer first (HF:1905)
    with Replay(crit_draws), torch.no_grad():
        idx = model.criterion.matcher(out.masks_queries_logits, out.class_queries_logits, mask_labels, class_labels)
    for i in range(B):
        arrays[f"row_{i}"] = idx[i][0]
        arrays[f"col_{i}"] = idx[i][1]
    with Replay(crit_draws), torch.no_grad():
        ld = model.get_loss_dict(out.masks_queries_logits, out.class_queries_logits, mask_labels, class_labels,
                                 out.auxiliary_logits)
    assert abs(float(sum(ld.values())) - float(out.loss)) < 1e-4 * abs(float(out.loss))
    for k, v in ld.items():
        arrays["ld." + k] = v

    # no-label call (eval path, metrics.py:56)
    with torch.no_grad():
        out2 = model(pixel_values=pixel_values)
    assert torch.equal(out2.masks_queries_logits, out.masks_queries_logits)

    for k, v in model.state_dict().items():
        arrays["sd." + k] = v
    save("full_tiny.npz", **arrays)


def gen_swin():
    """Swin backbone (transformers/models/swin/modeling_swin.py:1070-1150): feature maps of a reduced config,
    on an input whose size is divisible by neither the patch nor the window (exercises every padding path)."""
    from transformers import SwinConfig
    from transformers.models.swin.modeling_swin import SwinBackbone
    torch.manual_seed(21)
    cfg = SwinConfig(embed_dim=16, depths=[1, 2, 2, 1], num_heads=[1, 2, 4, 4], window_size=4, mlp_ratio=2.0,
                     out_features=["stage1", "stage2", "stage3", "stage4"])
    m = SwinBackbone(cfg).eval()
    randomise(m, 21)
    g = torch.Generator().manual_seed(22)
    arrays = {}
    for tag, shape in (("a", (2, 3, 70, 98)), ("b", (1, 3, 64, 64))):
        x = torch.randn(*shape, generator=g)
        with torch.no_grad():
            fm = m(x).feature_maps
        arrays[f"x_{tag}"] = x
        for i, f in enumerate(fm):
            arrays[f"fm_{tag}_{i}"] = f
    for k, v in m.state_dict().items():
        arrays["sd." + k] = v
    arrays["config_json"] = np.asarray(json.dumps(cfg.to_dict(), default=str))
    save("swin_tiny_backbone.npz", **arrays)


def gen_postprocess():
    """Outputs of the dependency's own post_process_instance_segmentation (CPU) on small synthetic logits:
    three images, three target-size regimes (smaller than the 384 grid, larger, none) + the binary-map variant."""
    from types import SimpleNamespace
    # torchvision is absent here, so the torchvision-backed Mask2FormerImageProcessor does not import; the PIL-backed
    # class of the same package carries the same post_process_instance_segmentation (:665-785 of its file)
    from transformers.models.mask2former.image_processing_pil_mask2former import Mask2FormerImageProcessorPil
    proc = Mask2FormerImageProcessorPil()
    g = torch.Generator().manual_seed(31)
    B, Q, C, h, w = 3, 12, 3, 32, 40
    low = torch.randn(B, Q, 5, 6, generator=g) * 4.0 - 1.0
    masks = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False) + 0.3 * torch.randn(B, Q, h, w, generator=g)
    masks[0, 3] = -5.0  # an empty mask
    cls = torch.randn(B, Q, C + 1, generator=g) * 4.0
    cls[1, :, :] = -3.0
    cls[1, :, -1] = 3.0  # image 1: nothing above threshold
    cls[1, 2, 1] = 9.0   # ... except one query
    out = SimpleNamespace(class_queries_logits=cls, masks_queries_logits=masks)
    arrays = {"class_logits": cls, "mask_logits": masks}
    info = {}
    for tag, ts in (("none", None), ("mixed", [(50, 70), (400, 500), (384, 384)]), ("small", [(33, 47)] * 3)):
        res = proc.post_process_instance_segmentation(out, threshold=0.5, mask_threshold=0.5, target_sizes=ts)
        for i, r in enumerate(res):
            arrays[f"seg_{tag}_{i}"] = r["segmentation"].to(torch.int16)
        info[tag] = {"target_sizes": ts, "segments_info": [r["segments_info"] for r in res]}
    res = proc.post_process_instance_segmentation(out, threshold=0.5, target_sizes=[(50, 70)] * 3, return_binary_maps=True)
    for i, r in enumerate(res):
        arrays[f"maps_{i}"] = r["segmentation"].to(torch.int16)
    info["maps"] = {"target_sizes": [(50, 70)] * 3, "segments_info": [r["segments_info"] for r in res]}
    res = proc.post_process_instance_segmentation(out, threshold=0.5, target_sizes=[(20, 24)] * 3, return_coco_annotation=True)
    info["rle"] = {"target_sizes": [(20, 24)] * 3, "segments_info": [r["segments_info"] for r in res],
                   "segmentation": [r["segmentation"] for r in res]}
    arrays["info_json"] = np.asarray(json.dumps(info, default=lambda o: int(o)))
    save("postprocess_instances.npz", **arrays)
    print("kept per image:", [len(x) for x in info["mixed"]["segments_info"]])


def gen_labelmap():
    """convert_segmentation_map_to_binary_masks of the dependency on a synthetic instance map in the reference's
    format (datasets/pheno_bench/dataset.py:85-116: int32 map, 255 = background/ignore, ids 1..N skipping 255)."""
    from transformers.models.mask2former.image_processing_pil_mask2former import convert_segmentation_map_to_binary_masks
    rng = np.random.default_rng(7)
    H, W = 48, 64
    inst = np.full((H, W), 255, dtype=np.int32)
    id2sem = {}
    for iid in [1, 2, 3, 5, 8, 254, 256, 300]:
        hh, ww = rng.integers(4, 16, 2)
        y0, x0 = rng.integers(0, H - hh), rng.integers(0, W - ww)
        inst[y0:y0 + hh, x0:x0 + ww] = iid
        id2sem[iid] = int(rng.integers(1, 5))
    masks, labels = convert_segmentation_map_to_binary_masks(inst, id2sem, ignore_index=255)
    save("labelmap_masks.npz", instance_map=inst, masks=masks.astype(np.uint8), labels=labels,
         id2sem_json=np.asarray(json.dumps({str(k): v for k, v in id2sem.items()})))


def gen_state_keys():
    from transformers import SwinConfig
    res = {}
    with torch.device("meta"):
        bc = ResNetConfig(out_features=["stage1", "stage2", "stage3", "stage4"])
        m = Mask2FormerForUniversalSegmentation(Mask2FormerConfig(backbone_config=bc, num_labels=3, num_queries=100))
        res["resnet50_q100_l3"] = {k: list(v.shape) for k, v in m.state_dict().items()}
        res["resnet50_config"] = json.loads(json.dumps(m.config.to_dict(), default=str))
        sc = SwinConfig(out_features=["stage1", "stage2", "stage3", "stage4"])
        m = Mask2FormerForUniversalSegmentation(Mask2FormerConfig(backbone_config=sc, num_labels=3, num_queries=100))
        res["swin_tiny_q100_l3"] = {k: list(v.shape) for k, v in m.state_dict().items()}
        res["swin_tiny_config"] = json.loads(json.dumps(m.config.to_dict(), default=str))
        # the checkpoint family the reference itself trains from (config.py:4 `facebook/mask2former-swin-large-coco-
        # instance`): Swin-L (embed 192, depths 2-2-18-2, heads 6-12-24-48, window 12, 384 px pretraining), 200 queries,
        # 80 COCO classes -- built from a LOCAL config on the meta device (no download, no weights)
        sl = SwinConfig(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=12, image_size=384,
                        drop_path_rate=0.3, out_features=["stage1", "stage2", "stage3", "stage4"])
        m = Mask2FormerForUniversalSegmentation(Mask2FormerConfig(backbone_config=sl, num_labels=80, num_queries=200))
        res["swin_large_q200_l80"] = {k: list(v.shape) for k, v in m.state_dict().items()}
        res["swin_large_config"] = json.loads(json.dumps(m.config.to_dict(), default=str))
    res["hf_version"] = META["hf_version"]
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(res, f)
    print("wrote state_keys.json", os.path.getsize(os.path.join(HERE, "state_keys.json")) // 1024, "KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["k1", "a2", "k2", "k3", "k4", "full", "keys", "swin", "post", "labelmap"]
    torch.set_num_threads(8)
    for w in which:
        {"k1": gen_k1, "a2": gen_a2, "k2": gen_k2, "k3": gen_k3, "k4": gen_k4, "full": gen_full,
         "keys": gen_state_keys, "swin": gen_swin, "post": gen_postprocess, "labelmap": gen_labelmap}[w]()
