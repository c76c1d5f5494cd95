#!/usr/bin/env python3
"""The dependency's own path on the same GPU: transformers' Mask2FormerForUniversalSegmentation (stock PyTorch ops, eager) at the
bench workload (BASELINE.json configs[1]: 1024x1024, ResNet-50, 100 queries, fp32 forward, B = 8), timed exactly like bench.py, next to
this package's model holding THE SAME weights -- plus the difference of their outputs.  Not part of the product or of bench.py.
Usage: python tools/hf_gpu_baseline.py [--B 8] [--size 1024] [--steps 5] [--warmup 2] [--amp off|bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--amp", default="off", choices=["off", "bf16"])
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"])
    a = ap.parse_args()
    from transformers import Mask2FormerConfig, Mask2FormerForUniversalSegmentation, ResNetConfig
    import transformers
    import weed_instance_segmentation_amd as W

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bc = ResNetConfig(out_features=["stage1", "stage2", "stage3", "stage4"])
    cfg = Mask2FormerConfig(backbone_config=bc, num_labels=3, num_queries=100)
    hf = Mask2FormerForUniversalSegmentation(cfg).eval().to(dev)
    mine = W.Mask2FormerForUniversalSegmentation(W.Mask2FormerConfig(num_labels=3, num_queries=100)).eval().to(dev)
    missing = mine.load_state_dict(hf.state_dict(), strict=False)
    x = torch.randn(a.B, 3, a.size, a.size, device=dev)
    ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if a.amp == "bf16" else (lambda: torch.autocast("cuda", enabled=False))
    if a.mode == "train":  # full train step (loss with Hungarian matching, backward, AdamW) on bench.py's synthetic labels
        from bench import synthetic_labels
        ml, cl = synthetic_labels(a.B, a.size, a.size, seed=0, device=dev)
        res = {"transformers": transformers.__version__, "torch": torch.__version__, "B": a.B, "size": a.size, "amp": a.amp, "mode": "train"}
        for name, model in (("hf_eager", hf), ("wm2f", mine)):
            model.train()
            opt = torch.optim.AdamW(model.parameters(), lr=5e-5)
            last = {}

            def step():
                with ctx():
                    out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
                out.loss.backward()
                opt.step()
                opt.zero_grad(set_to_none=True)
                last["loss"] = out.loss.detach()
            res[f"{name}_ms_per_step"] = round(timed(step, a.steps, a.warmup), 3)
            res[f"{name}_last_loss"] = round(float(last["loss"]), 4)
            res[f"{name}_images_per_s"] = round(a.B / res[f"{name}_ms_per_step"] * 1e3, 2)
            del opt
            model.eval()
            torch.cuda.empty_cache()
        res["speedup"] = round(res["hf_eager_ms_per_step"] / res["wm2f_ms_per_step"], 3)
        print(json.dumps(res))
        return
    with torch.no_grad(), ctx():
        o_hf = hf(pixel_values=x)
        o_me = mine(pixel_values=x)
        mh, mm = o_hf.masks_queries_logits.float(), o_me.masks_queries_logits.float()
        ch, cm = o_hf.class_queries_logits.float(), o_me.class_queries_logits.float()
        res = {
            "transformers": transformers.__version__, "torch": torch.__version__, "B": a.B, "size": a.size, "amp": a.amp,
            "missing_keys": len(missing.missing_keys), "unexpected_keys": len(missing.unexpected_keys),
            "mask_logit_rel_err": float((mh - mm).abs().max() / mh.abs().max()),
            "class_logit_max_abs_err": float((ch - cm).abs().max()),
        }
        res["hf_eager_ms_per_step"] = round(timed(lambda: hf(pixel_values=x), a.steps, a.warmup), 3)
        res["wm2f_ms_per_step"] = round(timed(lambda: mine(pixel_values=x), a.steps, a.warmup), 3)
    res["hf_eager_images_per_s"] = round(a.B / res["hf_eager_ms_per_step"] * 1e3, 2)
    res["wm2f_images_per_s"] = round(a.B / res["wm2f_ms_per_step"] * 1e3, 2)
    res["speedup"] = round(res["hf_eager_ms_per_step"] / res["wm2f_ms_per_step"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
