#!/usr/bin/env python3
"""ResNet-50 backbone forward at the bench shape (B = 8, 1024^2, fp32, inference) in three forms:
   fused   -- what runs: NCHW, conv without bias + wm2f bias(+residual)+ReLU pass
   stock   -- NCHW, conv with bias, torch add / relu
   nhwc    -- channels_last input and weights (PYTORCH_MIOPEN_SUGGEST_NHWC=1), conv with bias, torch add / relu
Usage: python tools/probes/backbone_layout_probe.py"""
import json
import os
import sys
import time

os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import weed_instance_segmentation_amd as W  # noqa: E402
from weed_instance_segmentation_amd import backbone_resnet as R  # noqa: E402


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e3, 3)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = W.Mask2FormerForUniversalSegmentation(W.Mask2FormerConfig(num_labels=3, num_queries=100)).eval().to(dev)
    bb = [m for m in model.modules() if isinstance(m, R.ResNetBackbone)][0]
    x = torch.randn(8, 3, 1024, 1024, device=dev)
    res = {}
    with torch.no_grad():
        ref = bb(x)
        res["fused_ms"] = timed(lambda: bb(x))
        fused_ok = R._fused_ok
        R._fused_ok = lambda t: False
        out = bb(x)
        res["stock_vs_fused_max_rel"] = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(ref, out))
        res["stock_ms"] = timed(lambda: bb(x))
        # channels_last: folded weights are cached .contiguous(); convert the cache entries after the first call
        xcl = x.contiguous(memory_format=torch.channels_last)
        for m in bb.modules():
            if hasattr(m, "_fold") and "w" in m._fold:
                m._fold["w"] = m._fold["w"].contiguous(memory_format=torch.channels_last)
        out = bb(xcl)
        res["nhwc_out_is_channels_last"] = bool(out[-1].is_contiguous(memory_format=torch.channels_last))
        res["nhwc_vs_fused_max_rel"] = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(ref, out))
        res["nhwc_ms"] = timed(lambda: bb(xcl))
        R._fused_ok = fused_ok
    print(json.dumps(res))


if __name__ == "__main__":
    main()
