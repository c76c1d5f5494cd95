#!/usr/bin/env python3
"""What the library (hipBLASLt through torch) makes of the token GEMMs of the pixel decoder / transformer decoder at config 2:
out (M, N) = x (M, K) @ W (N, K)^T + bias, fp32.  TFLOP/s per shape against the 157.3 TFLOP/s fp32 matrix peak."""
import json
import sys
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2]


shapes = [("value_proj / output_proj", 172032, 256, 256), ("offsets+logits merged", 172032, 256, 288), ("fc1 (+ReLU)", 172032, 256, 1024),
          ("fc2", 172032, 1024, 256), ("decoder K/V fine", 131072, 256, 256), ("decoder K/V mid", 32768, 256, 256),
          ("decoder K/V coarse", 8192, 256, 256)]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    us = timeit(lambda: F.linear(x, w, b))
    flop = 2.0 * M * K * N
    print(json.dumps({"gemm": name, "M": M, "K": K, "N": N, "us": round(us, 1), "TFLOPs": round(flop / us / 1e6, 1),
                      "frac_of_157.3": round(flop / us / 1e6 / 157.3, 3)}))

# does the library do better on the same GEMM cut into row chunks? (172 032 rows = 131 072 + 40 960, = 2 x 86 016, = 3 x 57 344)
print("--- row chunking of the 172 032-token GEMMs")
for name, K, N in (("value/output_proj", 256, 256), ("offsets+logits", 256, 288)):
    M = 172032
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for cuts in ([M], [131072, 40960], [86016, 86016], [57344] * 3, [65536, 65536, 40960], [43008] * 4):
        def run():
            o = 0
            for c in cuts:
                torch.addmm(b, x[o:o + c], w.t(), out=out[o:o + c])
                o += c
        us = timeit(run)
        print(json.dumps({"gemm": name, "cuts": cuts, "us": round(us, 1), "TFLOPs": round(2.0 * M * K * N / us / 1e6, 1)}))
