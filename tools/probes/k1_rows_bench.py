"""K1 training op on the projection rows (ops.ms_deform_attn_rows) at configs[2] size: forward and backward kernel times
(torch events around the op; the backward includes the zero fill of grad_value and, for bf16, its cast).
Usage: python tools/probes/k1_rows_bench.py [B] [bf16|f32]   (profiling build, WM2F_K1_LW_THREADS = 512 / 768: 8 / 12 waves per workgroup in the grad-rows kernel; 1 / 2: without staging / staging only; 3: that kernel alone)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import _lib, ops
if os.environ.get("WM2F_K1_LW_THREADS"):
    _lib.use_profiling_library()  # the A/B variants exist only in libwm2f_prof.so

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
dev = torch.device("cuda:0")
shapes = [(32, 32), (64, 64), (128, 128)]
S, H = sum(h * w for h, w in shapes), 8
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, S, H, 32, device=dev, generator=g).to(dt).requires_grad_()
off = torch.randn(B, S, H * 24, device=dev, generator=g) * 2.0
rows = torch.cat([off, torch.randn(B, S, H * 12, device=dev, generator=g)], -1).to(dt).requires_grad_()
go = torch.randn(B, S, 256, device=dev, generator=g).to(dt)


def ev():
    return torch.cuda.Event(enable_timing=True)


tf, tb = [], []
for i in range(8):
    value.grad = rows.grad = None
    a, b, c = ev(), ev(), ev()
    a.record()
    out = ops.ms_deform_attn_rows(value, shapes, rows, H)
    b.record()
    out.backward(go)
    c.record()
    torch.cuda.synchronize()
    if i >= 3:
        tf.append(a.elapsed_time(b) * 1e3)
        tb.append(b.elapsed_time(c) * 1e3)
print(json.dumps({"B": B, "dtype": str(dt), "lw_threads": os.environ.get("WM2F_K1_LW_THREADS", "1024"), "fwd_us": round(sum(tf) / len(tf), 1),
                  "bwd_us": round(sum(tb) / len(tb), 1), "checksum": [float(out.float().abs().mean()), float(rows.grad.float().abs().mean()),
                                                                       float(value.grad.float().abs().mean())]}))
