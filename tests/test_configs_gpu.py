"""One test per BASELINE.json configuration, at that configuration's workload, on the GPU (-m gpu).

configs[0]  PhenoBench crop/weed 256 x 256, ResNet-50 Mask2Former, batch 2: sample files -> collate_fn -> model -> loss ->
            instance post-processing, every stage against the CPU oracle.
configs[1]  synthetic 1024 x 1024, R50, 100 queries, fp32 forward, batch 8: tests/test_fullsize_gpu.py (kernels at size) and
            bench.py's cpu_baseline leg (whole forward against the oracle); here the batch-8 forward's size-independent
            properties.
configs[2]  the same model, bf16 autocast, full train step, batch 16.
configs[3]  Swin-T backbone, 1024 x 1024, bf16, batch 8 per GPU (64 over 8 GPUs): forward + train step of one rank's share
            (the exchange between ranks: tests/test_parallel_gpu.py).
configs[4]  Swin-B backbone, 1333 x 800 (COCO-style: padded to 1344 x 800 by the processor, and unpadded), 200 queries, bf16.

How parity is made exact where random weights make it chaotic.  The attention mask of layer i+1 is `sigmoid(resized
logits of layer i) < 0.5` (HF:2048-2054).  With random-init weights some of those logits sit within float rounding of
zero; two correct implementations then differ in a bit, and that query differs from there on.  So the comparisons below
teacher-force the oracle with the PRODUCT's mask bits (`forced_masks`): every level's logits, the class logits, the
matching and the loss are then compared exactly (fp32 round-off), and the bits themselves are compared with the oracle's
own decisions wherever the deciding logit is farther than 1e-4 of its range from the threshold.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import m2f_oracle as O

pytestmark = pytest.mark.gpu

PHENOBENCH_ID2LABEL = {0: "background", 1: "crop", 2: "weed", 3: "partial-crop", 4: "partial-weed"}  # datasets/pheno_bench/definitions.py:20-26


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


class Recorder(O.RandSource):
    """Draws what the criterion asks for (from a private generator) and keeps the draws, in the dependency's call order."""

    def __init__(self, seed=5):
        super().__init__()
        self.rec, self.gen = [], torch.Generator().manual_seed(seed)

    def rand(self, *shape):
        d = torch.rand(*shape, generator=self.gen)
        self.rec.append(d)
        return d


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def product_vs_forced_oracle(model, sd, cfgd, x, ml=None, cl=None, backbone_feats=None, draws=None, logit_tol=1e-3,
                             tie_tol=1e-4, max_ambiguous=2e-3):
    """Runs the product on the GPU (recording its attention-mask bytes), then the oracle with those bytes forced, and
    compares everything.  Returns (product output, oracle result)."""
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    dec = model.model.transformer_module.decoder
    dec.record_attention_masks = []
    try:
        kw = {}
        if ml is not None:
            kw = dict(mask_labels=[m.cuda() for m in ml], class_labels=[c.cuda() for c in cl],
                      point_provider=ReplayPointProvider(draws, cfgd["decoder_layers"], x.shape[0], "cuda"))
        with torch.no_grad():
            out = model(pixel_values=x.cuda(), output_auxiliary_logits=True, **kw)
        masks = [m.cpu() for m in dec.record_attention_masks]
    finally:
        dec.record_attention_masks = None
    assert len(masks) == cfgd["decoder_layers"] - 1
    res = O.forward(sd, cfgd, x, ml, cl, O.RandSource(draws) if draws is not None else None, backbone_feats=backbone_feats,
                    forced_masks=masks)
    # ---- the mask bits against the oracle's own decisions (same history on both sides)
    n_amb = n_all = 0
    for i, (m, d) in enumerate(zip(masks, res["mask_decisions"])):
        clear = d.abs() > tie_tol * d.abs().max()
        want = d < 0  # sigmoid(d) < 0.5
        bad = (m.bool().reshape(want.shape) != want) & clear
        assert not bad.any(), f"layer {i}: {int(bad.sum())} attention-mask bits differ where the oracle's logit is clear of the threshold"
        n_amb += int((~clear).sum())
        n_all += clear.numel()
    assert n_amb / n_all < max_ambiguous, (n_amb, n_all)
    # ---- every prediction level, the class logits
    levels = [a["masks_queries_logits"] for a in out.auxiliary_logits] + [out.masks_queries_logits]
    for i, (a, b) in enumerate(zip(levels, res["aux_masks"] + [res["masks_queries_logits"]])):
        assert a.shape == b.shape
        assert _rel(a.cpu(), b) < logit_tol, f"mask logits of level {i}: {_rel(a.cpu(), b):.3e}"
    torch.testing.assert_close(out.class_queries_logits.cpu(), res["class_queries_logits"], rtol=1e-3, atol=1e-3)
    if ml is not None:
        for i, ((r, c), (ro, co)) in enumerate(zip(out.matched_indices, res["indices"])):  # final level, per image
            assert torch.equal(r, ro) and torch.equal(c, co), f"Hungarian assignment of image {i} differs"
        torch.testing.assert_close(out.loss.cpu(), res["loss"], rtol=1e-3, atol=1e-3)
        for k, v in out.loss_dict.items():
            torch.testing.assert_close(v.cpu(), res["loss_dict"][k], rtol=2e-3, atol=1e-4)
    return out, res


def _phenobench_like_item(idx, size=256):
    """What PhenoBenchDataset.__getitem__ returns (datasets/pheno_bench/dataset.py:85-135) for a synthetic 256 x 256 tile:
    blobs of classes 1..4, one instance id per blob, 255 = background; mask stack / class ids as the processor's
    convert_segmentation_map_to_binary_masks gives them (oracle restatement, pinned by labelmap_masks.npz)."""
    rng = np.random.default_rng(40 + idx)
    inst = np.full((size, size), 255, dtype=np.int32)
    id2sem = {}
    for iid in range(1, 6 + idx):
        hh, ww = rng.integers(16, 80, 2)
        y0, x0 = rng.integers(0, size - hh), rng.integers(0, size - ww)
        yy, xx = np.ogrid[:hh, :ww]
        blob = ((yy - hh / 2) / (hh / 2)) ** 2 + ((xx - ww / 2) / (ww / 2)) ** 2 <= 1.0
        inst[y0:y0 + hh, x0:x0 + ww][blob] = iid
        id2sem[iid] = int(rng.integers(1, 5))
    id2sem = {k: v for k, v in id2sem.items() if (inst == k).any()}
    masks, labels = O.convert_segmentation_map_to_binary_masks(torch.from_numpy(inst), id2sem, ignore_index=255)
    pix = torch.randn(3, size, size, generator=torch.Generator().manual_seed(70 + idx))
    return {"pixel_values": pix, "mask_labels": masks, "class_labels": labels, "target_size": (size, size), "original_map": inst,
            "id_to_semantic": id2sem, "file_name": f"tile_{idx:03d}.png"}


def test_config0_phenobench_256_resnet50_batch2_sample_to_postprocess(tmp_path):
    """BASELINE.json configs[0]: PhenoBench crop/weed 256 x 256, ResNet-50 Mask2Former (full depth, 100 queries, the 5
    PhenoBench classes), batch 2, the reference's plumbing: .pt sample files in the reference's layout (torch.save of
    the item dict, numpy original_map: datasets/dataset_utils.py:56-70) -> PreprocessedDataset -> collate_fn (:32-53) ->
    model(pixel_values, mask_labels, class_labels).loss (train.py:196) and model(pixel_values) ->
    post_process_instance_segmentation(threshold=0.5, mask_threshold=0.5, target_sizes) (metrics.py:56-63)."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation, data
    from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
    for i in range(2):  # the reference's writer is torch.save(item, <basename>.pt)
        item = _phenobench_like_item(i)
        torch.save(item, str(tmp_path / (os.path.splitext(item["file_name"])[0] + ".pt")))
    ds = data.PreprocessedDataset(str(tmp_path))
    batch = data.collate_fn([ds[0], ds[1]])
    assert batch["pixel_values"].shape == (2, 3, 256, 256) and isinstance(batch["original_maps"][0], np.ndarray)
    cfg = Mask2FormerConfig(id2label=PHENOBENCH_ID2LABEL, num_queries=100)  # ResNet-50, 6 + 9 layers: the full model
    torch.manual_seed(0)
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfgd = cfg.to_dict()
    model = model.cuda()
    x, ml, cl = batch["pixel_values"], batch["mask_labels"], batch["class_labels"]
    rs = Recorder()
    O.forward(sd, cfgd, x, ml, cl, rand_source=rs)  # free-running pass: only to learn the draws the criterion makes
    out, res = product_vs_forced_oracle(model, sd, cfgd, x, ml, cl, draws=rs.rec)
    assert out.masks_queries_logits.shape == (2, 100, 64, 64) and out.class_queries_logits.shape == (2, 100, 6)
    # the device batch path (expand_labels) gives the same loss as handing the tensors over by hand
    from weed_instance_segmentation_amd.loss import ReplayPointProvider
    dev_batch = data.expand_labels(batch, "cuda")
    with torch.no_grad():
        out2 = model(pixel_values=dev_batch["pixel_values"], mask_labels=dev_batch["mask_labels"],
                     class_labels=dev_batch["class_labels"], point_provider=ReplayPointProvider(rs.rec, cfgd["decoder_layers"], 2, "cuda"))
    torch.testing.assert_close(out2.loss, out.loss, rtol=1e-6, atol=1e-6)
    # ---- evaluation leg: forward without labels, then post-processing at the reference's thresholds (and a lower one,
    # so that random-init scores keep some instances), against the oracle's restatement fed the ORACLE's logits
    with torch.no_grad():
        ev = model(pixel_values=dev_batch["pixel_values"])
    assert _rel(ev.masks_queries_logits.cpu(), out.masks_queries_logits.cpu()) < 1e-3  # level-resolution mask route == full route
    proc = Mask2FormerInstancePostProcessor()
    kept = 0
    for thr in (0.5, 0.02):
        mine = proc.post_process_instance_segmentation(out, threshold=thr, mask_threshold=0.5, target_sizes=batch["target_sizes"])
        ref = O.post_process_instance_segmentation(res["class_queries_logits"], res["masks_queries_logits"], thr, batch["target_sizes"])
        for a, b in zip(mine, ref):
            assert [s["label_id"] for s in a["segments_info"]] == [s["label_id"] for s in b["segments_info"]]
            for sa, sb in zip(a["segments_info"], b["segments_info"]):
                assert abs(sa["score"] - sb["score"]) < 1e-4
            assert (a["segmentation"].cpu() != b["segmentation"]).float().mean().item() < 1e-3
            kept += len(b["segments_info"])
    assert kept > 0


def _synthetic_labels(B, H, W, T=16, seed=0):
    """T random axis-aligned rectangles per image (side 32..256) as uint8 masks, classes in {0,1,2}."""
    rng = np.random.default_rng(seed)
    ml, cl = [], []
    for _ in range(B):
        m = torch.zeros(T, H, W, dtype=torch.uint8)
        for t in range(T):
            hh, ww = rng.integers(32, 257, 2)
            y0, x0 = rng.integers(0, H - hh + 1), rng.integers(0, W - ww + 1)
            m[t, y0:y0 + hh, x0:x0 + ww] = 1
        ml.append(m.cuda())
        cl.append(torch.as_tensor(rng.integers(0, 3, T), dtype=torch.int64).cuda())
    return ml, cl


def _train_step_losses(model, x, ml, cl, seed=3):
    """The fp32 loss (forward only: on a fresh box every new convolution shape costs a MIOpen kernel build, and an fp32
    backward nobody asked for would double them) and ONE full bf16-autocast train step -- forward, loss, backward -- on
    the same batch with the same sampled points.  Returns (fp32 loss, bf16 loss); checks finite gradients."""
    from weed_instance_segmentation_amd.loss import DevicePointProvider
    prov = lambda: DevicePointProvider("cuda", torch.Generator(device="cuda").manual_seed(seed))
    model.train()
    try:
        with torch.no_grad():
            l32 = float(model(pixel_values=x, mask_labels=ml, class_labels=cl, point_provider=prov()).loss)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pixel_values=x, mask_labels=ml, class_labels=cl, point_provider=prov())
        out.loss.backward()
        assert torch.isfinite(out.loss)
        n = 0
        for p in model.parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all()
                n += 1
        assert n > 100
        return l32, float(out.loss)
    finally:
        model.eval()
        model.zero_grad(set_to_none=True)


def test_config1_resnet50_1024_fp32_batch8_forward_properties():
    """BASELINE.json configs[1]: synthetic 1024 x 1024 3-class, ResNet-50, 100 queries, fp32 forward-only, batch 8 -- the
    bench workload.  (Against the oracle at this size: bench.py's cpu_baseline leg, 2 images, every run.)
    Run to run: STRICT.  Everything behind the backbone -- pixel decoder (K1, token GEMMs, fused passes, library GEMMs /
    convolutions) and transformer decoder (K2, K3, mask bits) -- is bit-identical between two calls on identical backbone
    features, and so are two whole forwards once the libraries have settled on their kernels (the FIRST call of a process
    may run other convolution solvers than the later ones: reported below, not asserted).
    Size-independent properties that legitimately change the rounding (other GEMM shapes / another summation order): image i
    alone against image i of the batch, the level-resolution mask route against the full-resolution one -- compared per
    query, because a logit within rounding of the threshold may flip a mask bit (module docstring)."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    torch.manual_seed(0)
    model = Mask2FormerForUniversalSegmentation(Mask2FormerConfig(num_labels=3, num_queries=100)).cuda().eval()
    x = torch.randn(8, 3, 1024, 1024, generator=torch.Generator().manual_seed(1)).cuda()
    plm = model.model.pixel_level_module
    feats = []
    hook = plm.encoder.register_forward_hook(lambda m, i, o: feats.append([t.clone() for t in o]))
    with torch.no_grad():
        first = model(pixel_values=x)
        a = model(pixel_values=x)
        b = model(pixel_values=x)
    hook.remove()
    assert a.masks_queries_logits.shape == (8, 100, 256, 256) and torch.isfinite(a.masks_queries_logits).all()
    same = lambda u, v: all(torch.equal(p, q) for p, q in zip(u, v))
    first_differs = not same(feats[0], feats[1])
    backbone_stable = same(feats[1], feats[2])
    print(f"config1: backbone features call 1 vs call 2 {'DIFFER' if first_differs else 'equal'}; call 2 vs call 3 "
          f"{'equal' if backbone_stable else 'DIFFER'}")
    # (1) identical backbone features -> bit-identical pixel decoder + transformer decoder, twice
    real_encoder = plm.encoder

    class _Cached(torch.nn.Module):
        channels = getattr(real_encoder, "channels", None)

        def forward(self, _x):
            return [t.clone() for t in feats[1]]

    plm.encoder = _Cached()
    recs, order = [], []
    leafish = (torch.nn.Conv2d, torch.nn.Linear, torch.nn.GroupNorm, torch.nn.LayerNorm)
    flat = lambda o: [o] if isinstance(o, torch.Tensor) else [t for e in o for t in flat(e)] if isinstance(o, (list, tuple)) else []
    hooks = []

    def mk(name):
        def fn(_m, inp, out):
            recs[-1][name] = ([t.detach().clone() for t in flat(inp)], [t.detach().clone() for t in flat(out)])
            if name not in order:
                order.append(name)
        return fn

    for name, mod in model.named_modules():
        if name and (isinstance(mod, leafish) or type(mod).__name__ in ("MSDeformAttn", "MaskedCrossAttention", "SelfAttention")):
            hooks.append(mod.register_forward_hook(mk(name)))
    try:
        with torch.no_grad():
            recs.append({})
            c = model(pixel_values=x)
            recs.append({})
            d = model(pixel_values=x)
    finally:
        plm.encoder = real_encoder
        for h in hooks:
            h.remove()
    strict = True
    if not (torch.equal(c.masks_queries_logits, d.masks_queries_logits) and torch.equal(c.class_queries_logits, d.class_queries_logits)):
        strict = False
        # say WHERE two runs on identical features part: the first module (execution order) whose output differs, and whether its
        # inputs were still identical (then the op itself is not reproducible) -- this package's kernels hold bit-identity on their
        # own (tests/test_fullsize_gpu.py), the library GEMM / convolution kernels around them are not promised to
        eq = lambda u, v: len(u) == len(v) and all(torch.equal(p, q) for p, q in zip(u, v))
        first = next((n for n in order if n in recs[0] and n in recs[1] and not eq(recs[0][n][1], recs[1][n][1])), None)
        where, ours = "outside the hooked modules", False
        if first is not None:
            mod = dict(model.named_modules())[first]
            same_in = eq(recs[0][first][0], recs[1][first][0])
            dm = max(float((p.float() - q.float()).abs().max()) for p, q in zip(recs[0][first][1], recs[1][first][1]))
            where = (f"{first} ({type(mod).__name__}), inputs {'identical' if same_in else 'already different'}, "
                     f"max abs difference {dm:.3e}")
            ours = same_in and not isinstance(mod, leafish)  # MSDeformAttn / attention modules: this package's kernels inside
        # One of this package's kernels giving two answers on identical inputs is a failure.  A stock module (library GEMM /
        # convolution) doing so is the libraries' business (seen once in ~10 runs of this test, never in 150 forwards of
        # tools/probes/determinism_probe.py): reported, and the two results held to round-off per query instead.
        assert not ours, "two forwards on identical backbone features differ inside this package's kernels: " + where
        print("config1: two forwards on identical backbone features differ; first differing module: " + where)
        scale_cd = c.masks_queries_logits.abs().max().item()
        per_q = (c.masks_queries_logits - d.masks_queries_logits).abs().flatten(-2).amax(-1) / scale_cd
        assert (per_q < 1e-4).float().mean().item() > 0.97, per_q.max()
    del recs
    # (2) whole forwards: the cached features ARE call a's; second against third call strict whenever the backbone's own outputs
    # repeated (and no library noise was seen above)
    if strict:
        assert torch.equal(c.masks_queries_logits, a.masks_queries_logits)
    if backbone_stable and strict:
        assert torch.equal(a.masks_queries_logits, b.masks_queries_logits) and torch.equal(a.class_queries_logits, b.class_queries_logits)
    scale = a.masks_queries_logits.abs().max().item()
    with torch.no_grad():
        one = model(pixel_values=x[5:6])
        full = model(pixel_values=x, output_auxiliary_logits=True)  # every prediction at the mask-feature resolution
    pairs = [(one.masks_queries_logits[0], a.masks_queries_logits[5]), (full.masks_queries_logits, a.masks_queries_logits),
             (first.masks_queries_logits, a.masks_queries_logits)]
    if not (backbone_stable and strict):
        pairs.append((b.masks_queries_logits, a.masks_queries_logits))
    for other, sl in pairs:
        per_q = (other - sl).abs().flatten(-2).amax(-1) / scale
        assert (per_q < 1e-4).float().mean().item() > 0.97, per_q.max()
    torch.testing.assert_close(a.class_queries_logits, first.class_queries_logits, rtol=1e-3, atol=1e-3)


def _grad_step(model, x, ml, cl, amp, seed=3):
    """One full train-mode step (forward, loss, backward) with fixed sampled points; returns (loss, {name: grad})."""
    from weed_instance_segmentation_amd.loss import DevicePointProvider
    prov = DevicePointProvider("cuda", torch.Generator(device="cuda").manual_seed(seed))
    model.train()
    try:
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = model(pixel_values=x, mask_labels=ml, class_labels=cl, point_provider=prov)
        out.loss.backward()
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        return float(out.loss.detach()), grads
    finally:
        model.eval()
        model.zero_grad(set_to_none=True)


def test_config2_resnet50_1024_bf16_train_step_batch16():
    """BASELINE.json configs[2]: the configs[1] model, bf16 autocast, full train step (Hungarian matching + mask / dice /
    class loss over 10 levels + backward through K1 / K2 / K3), batch 16 on one GPU, against the SAME step in fp32 on the
    same batch and the same sampled points: finite loss and gradients, the bf16 loss within 5 % of the fp32 loss, and the
    parameter gradients tensor by tensor -- direction (cosine) and size (norm ratio) -- not only the scalar.  bf16 rounds every
    stock-op activation to 8 bits and may flip a Hungarian assignment or a mask bit, so single tensors may disagree; the
    bounds are on the distribution over the ~650 parameter tensors, weighted towards the ones that carry the gradient."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    torch.manual_seed(0)
    model = Mask2FormerForUniversalSegmentation(Mask2FormerConfig(num_labels=3, num_queries=100)).cuda()
    x = torch.randn(16, 3, 1024, 1024, generator=torch.Generator().manual_seed(1)).cuda()
    ml, cl = _synthetic_labels(16, 1024, 1024)
    l32, g32 = _grad_step(model, x, ml, cl, amp=False)
    l16, g16 = _grad_step(model, x, ml, cl, amp=True)
    assert np.isfinite(l16) and abs(l16 - l32) / abs(l32) < 5e-2, (l32, l16)
    assert g32.keys() == g16.keys() and len(g16) > 100
    names, cos, ratio, weight = [], [], [], []
    for n in g32:
        a, b = g32[n].flatten().double(), g16[n].flatten().double()
        assert torch.isfinite(b).all(), n
        na, nb = float(a.norm()), float(b.norm())
        if na == 0.0:
            continue
        names.append(n)
        cos.append(float(a @ b) / max(na * nb, 1e-300))
        ratio.append(nb / na)
        weight.append(na)
    cos, ratio, weight = np.array(cos), np.array(ratio), np.array(weight)
    backbone = np.array([n.startswith("model.pixel_level_module.encoder.") for n in names])

    def stats(sel, label):
        e = weight[sel] ** 2 / (weight[sel] ** 2).sum()  # share of this group's squared gradient norm
        c, r = cos[sel], ratio[sel]
        worst = np.argsort(c)[:5]
        nm = [n for n, s_ in zip(names, sel) if s_]
        print(f"config2 bf16 vs fp32 gradients, {label} ({int(sel.sum())} tensors): cosine median {np.median(c):.4f}, "
              f"energy-weighted mean {float((e * c).sum()):.4f}, min {c.min():.4f}; norm ratio median {np.median(r):.3f}, energy-weighted "
              f"{float((e * r).sum()):.3f}; energy in tensors with cosine < 0.5: {float(e[c < 0.5].sum()):.2e}; worst: "
              + ", ".join(f"{nm[i]} cos {c[i]:.2f} energy {e[i]:.1e}" for i in worst))
        return float((e * c).sum()), float(np.median(c)), float(e[c < 0.5].sum()), float((e * r).sum()), float(np.median(r))

    # Everything behind the backbone -- pixel decoder (K1 and its token Linears), transformer decoder (K2, K3), heads: the
    # gradient there has passed this package's backward kernels and a few stock Linears only.  Strict.
    ew, med, lost, rw, rmed = stats(~backbone, "pixel decoder + transformer decoder + heads")
    assert ew > 0.99 and med > 0.995 and lost < 1e-3, (ew, med, lost)
    assert 0.95 < rw < 1.05 and 0.95 < rmed < 1.05, (rw, rmed)
    # The ResNet-50 backbone (stock convolutions and train-mode BatchNorm, all bf16 under autocast): REPORTED, and held to
    # finiteness and size only.  Measured here: the gradient that ARRIVES at the backbone is right (the input projections and
    # FPN adapters that consume its features are in the strict group above, cosine 0.9999), its size is right (norm ratio
    # 0.995 - 1.0), but its DIRECTION inside the backbone largely is not (energy-weighted cosine 0.11, median 0.31): 53
    # train-mode BatchNorm backwards in bf16 -- each subtracts two batch means from a gradient rounded to 8 bits -- on a
    # random-init network at batch 16.  These are stock ops run exactly as autocast runs them for the dependency's own ResNet;
    # nothing of this package is on that path.
    ew_b, med_b, lost_b, rw_b, rmed_b = stats(backbone, "ResNet-50 backbone (stock ops)")
    assert 0.8 < rw_b < 1.25 and 0.8 < rmed_b < 1.25, (rw_b, rmed_b)


def _swin_config(embed_dim, depths, heads, window, num_queries, num_labels=3, **over):
    from weed_instance_segmentation_amd import Mask2FormerConfig
    bc = {"model_type": "swin", "embed_dim": embed_dim, "depths": depths, "num_heads": heads, "window_size": window,
          "mlp_ratio": 4.0, "patch_size": 4, "num_channels": 3, "out_features": ["stage1", "stage2", "stage3", "stage4"],
          "drop_path_rate": 0.0}
    return Mask2FormerConfig(backbone_config=bc, num_labels=num_labels, num_queries=num_queries, **over)


def test_config3_swin_tiny_1024_bf16_batch8_forward_and_train_step():
    """BASELINE.json configs[3]: Swin-T backbone (embed 96, depths 2-2-6-2, window 7), 1024 x 1024, bf16, batch 64 sharded
    over 8 GPUs = 8 per GPU: one rank's share.  bf16 forward against the fp32 forward of the same weights (a few bf16 ulps
    of the logit range), and a full train step under autocast against the fp32 step."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    torch.manual_seed(0)
    model = Mask2FormerForUniversalSegmentation(_swin_config(96, [2, 2, 6, 2], [3, 6, 12, 24], 7, 100)).cuda().eval()
    x = torch.randn(8, 3, 1024, 1024, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        ref = model(pixel_values=x, output_auxiliary_logits=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pixel_values=x, output_auxiliary_logits=True)
    assert out.masks_queries_logits.shape == (8, 100, 256, 256) and torch.isfinite(out.masks_queries_logits.float()).all()
    # the first prediction precedes every thresholded mask: a clean bf16-vs-fp32 comparison
    first32, first16 = ref.auxiliary_logits[0]["masks_queries_logits"], out.auxiliary_logits[0]["masks_queries_logits"].float()
    assert _rel(first16, first32) < 4e-2, _rel(first16, first32)
    ml, cl = _synthetic_labels(8, 1024, 1024)
    l32, l16 = _train_step_losses(model, x, ml, cl)
    assert abs(l16 - l32) / abs(l32) < 5e-2, (l32, l16)


@pytest.mark.parametrize("hw", [(800, 1344), (800, 1333)])
def test_config4_200_queries_coco_size_against_oracle(hw):
    """BASELINE.json configs[4]: Swin backbone, 1333 x 800 COCO-style, 200 queries.  (800, 1344) is what the dependency's
    processor hands the model (padded to a multiple of 32: levels 25x42 / 50x84 / 100x168, exact 1 : 2 : 4 but not a
    multiple of the K1 tile); (800, 1333) unpadded gives 25x42 / 50x84 / 100x167 -- NOT 1 : 2 : 4.  Reduced-width Swin
    (the backbone is stock ops, pinned on CPU elsewhere), full-width pixel decoder (256, 8 heads: the production K1 / K2 /
    K3 kernels) with 2 + 4 layers, 200 queries, one image, against the oracle fed the same backbone features."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    cfg = _swin_config(24, [1, 1, 1, 1], [1, 2, 4, 8], 7, 200, encoder_layers=2, decoder_layers=4)
    torch.manual_seed(7)
    model = Mask2FormerForUniversalSegmentation(cfg).eval()
    with torch.no_grad():  # random-init offsets are a fixed pattern: add a data-dependent part so K1 sees irregular points
        g = torch.Generator().manual_seed(8)
        for n, p in model.named_parameters():
            if "sampling_offsets.weight" in n or "attention_weights.weight" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
    x = torch.randn(1, 3, *hw, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        feats = model.model.pixel_level_module.encoder(x)
    assert [tuple(f.shape[-2:]) for f in feats][1:] == [(100, (hw[1] + 7) // 8), (50, (hw[1] + 15) // 16), (25, (hw[1] + 31) // 32)]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out, res = product_vs_forced_oracle(model.cuda(), sd, cfg.to_dict(), x, backbone_feats=[f.clone() for f in feats])
    assert out.masks_queries_logits.shape == (1, 200, 200, (hw[1] + 3) // 4)


def test_config4_swin_base_1333x800_200_queries_bf16_properties():
    """BASELINE.json configs[4] at full width: Swin-B (embed 128, depths 2-2-18-2, window 12 as the COCO checkpoints),
    200 queries, 800 x 1344, bf16 autocast, batch 2 (one rank's share is data-parallel in the batch): finite outputs of
    the right shape, deterministic, and image i alone equals image i of the batch for (nearly) every query."""
    _need_gpu()
    from weed_instance_segmentation_amd import Mask2FormerForUniversalSegmentation
    torch.manual_seed(0)
    model = Mask2FormerForUniversalSegmentation(_swin_config(128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 200, num_labels=80)).cuda().eval()
    x = torch.randn(2, 3, 800, 1344, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = model(pixel_values=x)
        b = model(pixel_values=x)
        one = model(pixel_values=x[1:2])
    assert a.masks_queries_logits.shape == (2, 200, 200, 336) and a.class_queries_logits.shape == (2, 200, 81)
    assert torch.isfinite(a.masks_queries_logits.float()).all() and torch.isfinite(a.class_queries_logits.float()).all()
    assert torch.equal(a.masks_queries_logits, b.masks_queries_logits)
    scale = a.masks_queries_logits.float().abs().max().item()
    per_q = (one.masks_queries_logits[0].float() - a.masks_queries_logits[1].float()).abs().flatten(-2).amax(-1) / scale
    assert (per_q < 2e-2).float().mean().item() > 0.9, per_q.max()  # bf16 GEMMs pick batch-size dependent kernels


def test_a2_msdeform_module_golden_on_gpu():
    """the MSDeformAttn MODULE (value / offset / weight / output projections around K1,
    HF:954-1014) against the dependency's own output (a2_msdeform_module.npz), through both product paths: grad-enabled
    (separate projections + ops.ms_deform_attn) and inference (merged projection + the fused packed kernel)."""
    _need_gpu()
    from weed_instance_segmentation_amd.modeling import MSDeformAttn
    g = load_golden("a2_msdeform_module.npz")
    T = torch.from_numpy
    level_hw = [tuple(int(v) for v in r) for r in g["level_hw"]]
    d_model, heads = int(g["hidden"].shape[-1]), int(g["n_heads"])
    m = MSDeformAttn(d_model, heads, len(level_hw), 4)
    m.load_state_dict({k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    m = m.cuda()
    hidden, pos, ref = T(g["hidden"]).cuda(), T(g["pos"]).cuda(), T(g["ref"]).cuda()
    want = T(g["out"])
    # the product module shares reference points over the batch (valid ratios are 1: HF:1343-1345) and takes
    # `hidden + pos` ready-made; the fixture's position embedding is a random per-image tensor
    assert torch.equal(ref[0], ref[-1])
    hp = hidden + pos
    out_grad = m(hidden, None, ref[0].contiguous(), level_hw, hp=hp)
    with torch.no_grad():
        out_inf = m(hidden, None, ref[0].contiguous(), level_hw, hp=hp)
    assert out_grad.requires_grad and not out_inf.requires_grad  # the two product paths did run
    for o in (out_grad.detach(), out_inf):
        torch.testing.assert_close(o.cpu(), want, rtol=1e-4, atol=1e-5)
