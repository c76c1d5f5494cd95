#!/usr/bin/env python3
"""K2 forward: full-tile kernel (WM2F_K2_FULL=1) against the general one (=0) on the same inputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from weed_instance_segmentation_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (B, Q, N, H, D, dens) in [(1, 16, 16, 1, 32, 0.0), (1, 16, 64, 1, 32, 0.0), (1, 16, 64, 1, 32, 0.5), (2, 20, 64, 8, 32, 0.5),
                              (2, 100, 1024, 8, 32, 0.5), (2, 100, 4096, 8, 32, 0.9), (1, 40, 256, 4, 16, 0.5), (1, 40, 256, 4, 64, 0.5)]:
    g = torch.Generator().manual_seed(N + Q)
    E = H * D
    q = (torch.randn(B, Q, E, generator=g) * 0.3).to(dev)
    k = torch.randn(B, N, E, generator=g).to(dev)
    v = torch.randn(B, N, E, generator=g).to(dev)
    mask = (torch.rand(B, Q, N, generator=g) < dens).to(torch.uint8).to(dev)
    ro = (mask == 0).any(-1).to(torch.int32)
    outs = []
    for full in ("0", "1"):
        os.environ["WM2F_K2_FULL"] = full
        o = ops.masked_xattn(q, k, v, mask, ro, H)
        torch.cuda.synchronize()
        outs.append(o.clone())
    a, b = outs
    nan = torch.isnan(b)
    d = (a - b).abs()
    print(f"B{B} Q{Q} N{N} H{H} D{D} dens{dens}: nan {int(nan.sum())}/{b.numel()}  max|diff| {float(d[~nan].max()) if (~nan).any() else -1:.3e}  ref nan {int(torch.isnan(a).sum())}")
    if nan.any():
        idx = nan.nonzero()
        print("   first nan idx", idx[:4].tolist(), " nan rows (b,q):", sorted({(int(i[0]), int(i[1])) for i in idx})[:10])
