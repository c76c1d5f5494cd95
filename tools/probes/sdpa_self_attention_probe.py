"""The decoder's 100 x 100 self-attention (HF:1451-1584) as matmul + softmax + matmul (shipped) against the stock library's fused
scaled_dot_product_attention, inside the configs[1] forward.  Usage: python tools/probes/sdpa_self_attention_probe.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from weed_instance_segmentation_amd import modeling

dev = torch.device("cuda:0")
model = bench.build_model(0).to(dev).eval()
x = torch.randn(8, 3, 1024, 1024, device=dev)
ref = None
with torch.no_grad():
    for mode in (False, True, False, True):
        modeling.SDPA_SELF_ATTENTION = mode
        for _ in range(5):
            out = model(pixel_values=x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = model(pixel_values=x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        if ref is None:
            ref = out.masks_queries_logits.clone()
        d = float((out.masks_queries_logits - ref).abs().max() / ref.abs().max())
        print(json.dumps({"sdpa": mode, "ms_per_step": round(ms, 3), "rel_diff_vs_first": d}), flush=True)
