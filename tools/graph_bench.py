"""Eager vs HIP-graph forward (weed_instance_segmentation_amd.graph.GraphedForward) at two shapes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from weed_instance_segmentation_amd.graph import GraphedForward


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    model = bench.build_model().cuda().eval()
    res = {}
    for size, B, n in ((256, 2, 20), (1024, 8, 8)):
        x = torch.randn(B, 3, size, size, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                model(pixel_values=x)
            eager = timeit(lambda: model(pixel_values=x), n)
        fwd = GraphedForward(model, x)
        graphed = timeit(lambda: fwd(x), n)
        res[f"{size}x{size}_B{B}"] = {"eager_ms": round(eager, 3), "graph_ms": round(graphed, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
