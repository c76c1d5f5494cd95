"""configs[2] train step (bf16 autocast, B 16) with MIOpen's immediate mode (torch.backends.cudnn.benchmark = False, as shipped) against
its find mode (benchmark = True: every convolution configuration is searched at its first call).
Usage: python tools/probes/train_miopen_find_probe.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from weed_instance_segmentation_amd.parallel import DataParallelEngine

dev = torch.device("cuda:0")
B = 16
for find in (False, True):
    torch.backends.cudnn.benchmark = find
    model = bench.build_model(0).to(dev).train()
    x = torch.randn(B, 3, 1024, 1024, device=dev)
    ml, cl = bench.synthetic_labels(B, 1024, 1024, seed=0, device=dev)
    eng = DataParallelEngine(model, lr=5e-5)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
        eng.backward_and_step(out.loss)
        return out.loss.detach()

    t0 = time.perf_counter()
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    print(json.dumps({"cudnn_benchmark": find, "ms_per_step": round((time.perf_counter() - t0) / 4 * 1e3, 2), "first_two_steps_s": round(setup, 1),
                      "loss": round(float(loss), 4)}), flush=True)
    del model, eng
    torch.cuda.empty_cache()
