"""Which stock elementwise / copy ops run in the configs[1] forward, with shapes (torch.profiler, record_shapes): the rocprofv3 trace
names kernels, not call sites.  Usage: python tools/probes/fwd_elementwise_trace.py"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

dev = torch.device("cuda:0")
model = bench.build_model(0).to(dev).eval()
x = torch.randn(8, 3, 1024, 1024, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(pixel_values=x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        model(pixel_values=x)
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0 and not any(s in e.key for s in ("mm", "conv", "linear", "wm2f")):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for t, n, k, s in rows[:40]:
    print(f"{t:10.1f} us  n={n:3d}  {k:40s} {s}")
