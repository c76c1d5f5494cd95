"""K1 inside the model (configs[1] forward, B 8): tile ORDER x operand-row layout x cache hints, profiling build.

  rows   tm = token-major lane rows (B, S, heads * 36) from the library GEMM;  hm = head-major (heads, B, S, 36) from
         wm2f_token_linear_fwd(out_group = 36);  hv = hm + the value tensor head-major too, (heads, B, S, 32)
  mode   0 = what the product runs for these rows (tm: heads innermost, 4 tiles x 8 heads resident per XCD; hm / hv: slab order);
         800 = SLAB order (heads outermost: the XCD's 32 workgroups walk one (image, head) slab together); 801 / 802 / 803 = slab
         order + non-temporal operand loads / output stores / both; 814 / 815 / 816 = timing ablations of the slab-order kernel (no LDS reads / no window DMA / no operand loads and stores):
         OUTPUTS NOT VALID, the model's later layers see garbage -- only the K1 launch time means anything

Only in the model is the launch fed from HBM (kbench launches find their operands in L2 / Infinity Cache).
Usage: python tools/k1_slab_inmodel.py [--cases tm:0,hm:0,hm:800,...] [--iters 10] [--rounds 2]
Under rocprofv3 --pmc: one case, --rounds 1 --iters 3.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weed_instance_segmentation_amd import _lib, modeling, ops  # noqa: E402

_lib.use_profiling_library()
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="tm:0,hm:0,hm:802")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()

dev = torch.device("cuda:0")


def kernel_check(modes):
    """Every profiling mode of the slab-order kernel must give the bits of the product's kernel on the same operands."""
    B, H, S = 3, 8, 21 * 16 * 16
    shapes = [(16, 16), (32, 32), (64, 64)]
    g = torch.Generator(device="cpu").manual_seed(5)
    value = torch.randn(B, S, H, 32, generator=g).to(dev)
    res = {}
    rows = (torch.randn(H, B, S, 36, generator=torch.Generator().manual_seed(6)) * 2.0).to(dev)
    os.environ.pop("WM2F_K1_MODE", None)
    ref = ops.ms_deform_attn_fused_lanes(value, shapes, rows, H, head_major=True, slab_order=True)
    for m in modes:
        os.environ["WM2F_K1_MODE"] = str(m)
        out = ops.ms_deform_attn_fused_lanes(value, shapes, rows, H, head_major=True, slab_order=True)
        res[f"mode {m}"] = bool(torch.equal(out, ref))
    os.environ.pop("WM2F_K1_MODE", None)
    print(json.dumps({"kernel_check_bit_equal": res}), flush=True)
    assert all(res.values()), res


kernel_check(sorted(m for m in {int(c.split(":")[1]) for c in args.cases.split(",")} - {0} if m < 810 or 816 < m < 820))  # 814-816: timing ablations, outputs not valid
model = bench.build_model().to(dev).eval()
x = torch.randn(args.batch, 3, 1024, 1024, device=dev)
ref_out = None
for rnd in range(args.rounds):
    for case in args.cases.split(","):
        rows, mode = case.split(":")
        modeling.HEAD_MAJOR_ROWS = rows in ("hm", "hv")
        modeling.HEAD_MAJOR_VALUE = rows == "hv"
        if mode != "0":
            os.environ["WM2F_K1_MODE"] = mode
        else:
            os.environ.pop("WM2F_K1_MODE", None)
        with torch.no_grad():
            for _ in range(3):
                out = model(pixel_values=x)
            torch.cuda.synchronize()
            t = ops.KernelTimer()
            ops.set_kernel_timer(t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                out = model(pixel_values=x)
            e1.record()
            torch.cuda.synchronize()
            ops.set_kernel_timer(None)
        logits = out.masks_queries_logits.float()
        if ref_out is None:
            ref_out = logits.clone()
        err = float((logits - ref_out).abs().max() / ref_out.abs().max())
        print(json.dumps({"rows": rows, "mode": int(mode), "k1_in_model_us": round(t.summary()["msdeform_fused_fwd"][1], 2),
                          "step_ms": round(e0.elapsed_time(e1) / args.iters, 3), "rel_diff_vs_first_case": err}), flush=True)
