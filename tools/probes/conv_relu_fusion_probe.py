"""Does MIOpen's fused convolution + bias + ReLU (aten::miopen_convolution_relu / _add_relu) beat convolution + this package's
bias (+ residual) + ReLU pass (ops.bias_act_, 2.6 ms of the 43.5 ms forward)?  ResNet-50 shapes at configs[1] (B 8, 1024 x 1024 input).
Usage: python tools/probes/conv_relu_fusion_probe.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops

dev = torch.device("cuda:0")
B = 8


def t(f, n=10):
    for _ in range(3):
        r = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, r


# (cin, cout, k, stride, H_in): stage 1 / 2 / 3 bottleneck convolutions
for cin, cout, k, s, H in [(64, 64, 1, 1, 256), (64, 64, 3, 1, 256), (64, 256, 1, 1, 256), (256, 64, 1, 1, 256), (128, 128, 3, 1, 128),
                           (128, 512, 1, 1, 128), (256, 256, 3, 1, 64), (256, 1024, 1, 1, 64), (1024, 256, 1, 1, 64)]:
    x = torch.randn(B, cin, H, H, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    z = torch.randn(B, cout, Ho, Ho, device=dev)
    with torch.no_grad():
        us_conv, _ = t(lambda: torch.nn.functional.conv2d(x, w, None, s, pad))
        us_ours, r0 = t(lambda: ops.bias_act_(torch.nn.functional.conv2d(x, w, None, s, pad), b, None, True))
        us_fused, r1 = t(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [pad, pad], [1, 1], 1))
        us_ours_res, r2 = t(lambda: ops.bias_act_(torch.nn.functional.conv2d(x, w, None, s, pad), b, z, True))
        us_fused_res, r3 = t(lambda: torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, b, [s, s], [pad, pad], [1, 1], 1))
    print(json.dumps({"conv": [cin, cout, k, s, H], "conv_us": round(us_conv, 1), "conv+bias_act_us": round(us_ours, 1),
                      "miopen_convolution_relu_us": round(us_fused, 1), "conv+bias_act(residual)_us": round(us_ours_res, 1),
                      "miopen_convolution_add_relu_us": round(us_fused_res, 1),
                      "max_diff": [float((r0 - r1).abs().max()), float((r2 - r3).abs().max())]}), flush=True)
