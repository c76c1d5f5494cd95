"""K1 inside the model (configs[1] forward, B 8) with the profiling build's in-model switches: WM2F_K1_MODE = 200 (tiles of an
image walked in 2-wide vertical strips), 500 (Z-order), 300 (round-1 loader schedule), unset (raster, the shipped kernel).  kbench / probe
launches run with the operands in cache; only in the model is the launch fed from HBM, which is where a tile order that
re-uses window halos in L2 could matter.  Usage: python tools/k1_inmodel_modes.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weed_instance_segmentation_amd import _lib, ops
_lib.use_profiling_library()
import bench

dev = torch.device("cuda:0")
model = bench.build_model().to(dev).eval()
x = torch.randn(8, 3, 1024, 1024, device=dev)
for mode in ("", "600", "700", "500", "", "600", "700", "500"):
    if mode:
        os.environ["WM2F_K1_MODE"] = mode
    else:
        os.environ.pop("WM2F_K1_MODE", None)
    with torch.no_grad():
        for _ in range(3):
            model(pixel_values=x)
        torch.cuda.synchronize()
        t = ops.KernelTimer()
        ops.set_kernel_timer(t)
        for _ in range(10):
            model(pixel_values=x)
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
    print(json.dumps({"WM2F_K1_MODE": mode or "default", "k1_in_model_us": round(t.summary()["msdeform_fused_fwd"][1], 2)}), flush=True)
