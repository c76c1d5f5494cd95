"""configs[1] forward (fp32, B 8): the ResNet-50 backbone with NCHW activations (as shipped) against channels_last, stock ops only
(the fused bias + residual + ReLU pass switched off in both, so only the layout differs).  The shipped inference trace
(profiles/r03_bench_step_breakdown.txt) shows NHWC implicit-GEMM kernels behind 24 batched transposes next to NCHW Winograd
kernels: does handing MIOpen NHWC tensors pay in fp32?  Usage: python tools/probes/infer_channels_last.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from weed_instance_segmentation_amd import backbone_resnet

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = bench.build_model(0).to(dev).eval()
bb = model.model.pixel_level_module.encoder
x0 = torch.randn(B, 3, 1024, 1024, device=dev)
fused_ok = backbone_resnet._fused_ok


def run(tag, x, fused):
    backbone_resnet._fused_ok = fused_ok if fused else (lambda t: False)
    with torch.no_grad():
        for _ in range(3):
            feats = bb(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            feats = bb(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(json.dumps({"backbone": tag, "B": B, "ms": round(ms, 3), "checksum": round(float(sum(f.double().abs().mean() for f in feats)), 6),
                      "out_strides": [list(f.stride()) for f in feats][-1]}), flush=True)


for _ in range(2):
    run("nchw, fused epilogue (shipped)", x0, True)
    run("nchw, stock ops", x0, False)
    run("channels_last input, stock ops", x0.contiguous(memory_format=torch.channels_last), False)
# folded weights in channels_last as well (the fold cache is keyed on parameter versions: patched in place)
for m in bb.modules():
    f = getattr(m, "_fold", None)
    if f and "w" in f:
        f["w"] = f["w"].contiguous(memory_format=torch.channels_last)
for _ in range(2):
    run("channels_last input + weights, stock ops", x0.contiguous(memory_format=torch.channels_last), False)
