"""How do the library convolutions of the ResNet-50 backbone (inference, fp32, batch 8, 1024x1024 -- configs[1]) respond to
MIOpen's find mode (torch.backends.cudnn.benchmark) and to channels_last activations?  Times the backbone alone with HIP events.
Usage: python tools/probes/backbone_probe.py [--iters 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from weed_instance_segmentation_amd.configuration import Mask2FormerConfig
    from weed_instance_segmentation_amd.backbone_resnet import build_backbone
    cfg = Mask2FormerConfig.tiny_resnet() if False else None
    bcfg = {"model_type": "resnet", "num_channels": 3, "embedding_size": 64, "hidden_sizes": [256, 512, 1024, 2048], "depths": [3, 4, 6, 3],
            "layer_type": "bottleneck", "out_features": ["stage1", "stage2", "stage3", "stage4"], "downsample_in_first_stage": False,
            "downsample_in_bottleneck": False}
    torch.manual_seed(0)
    net = build_backbone(bcfg).cuda().eval()
    x = torch.randn(8, 3, 1024, 1024, device="cuda")

    def timeit(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    with torch.no_grad():
        for bench in (False, True):
            torch.backends.cudnn.benchmark = bench
            t0 = time.time()
            ms = timeit(lambda: net(x), a.iters)
            print(json.dumps({"backbone": "resnet50 fp32 b8 1024", "cudnn_benchmark": bench, "memory_format": "contiguous", "ms": round(ms, 3), "setup_s": round(time.time() - t0, 1)}), flush=True)
        net_cl = net.to(memory_format=torch.channels_last)
        x_cl = x.contiguous(memory_format=torch.channels_last)
        for bench in (False, True):
            torch.backends.cudnn.benchmark = bench
            t0 = time.time()
            try:
                ms = timeit(lambda: net_cl(x_cl), a.iters)
                print(json.dumps({"backbone": "resnet50 fp32 b8 1024", "cudnn_benchmark": bench, "memory_format": "channels_last", "ms": round(ms, 3), "setup_s": round(time.time() - t0, 1)}), flush=True)
            except Exception as e:  # the fused bias pass is NCHW-only
                print(json.dumps({"memory_format": "channels_last", "cudnn_benchmark": bench, "error": str(e)[:200]}), flush=True)


if __name__ == "__main__":
    main()
