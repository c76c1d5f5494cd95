"""configs[2] train step (bf16 autocast, B 16) with the backbone in NCHW (as shipped) against channels_last: under bf16 MIOpen's
implicit-GEMM convolutions are NHWC kernels and every NCHW tensor costs a batched transpose in and out
(profiles/r03_train_bf16_b16_step_breakdown_start_of_round.txt: 192 launches, 9.8 ms per step).  Usage: python tools/probes/train_channels_last.py"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from weed_instance_segmentation_amd.parallel import DataParallelEngine

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for mode in ("nchw", "input_channels_last", "channels_last", "nchw", "input_channels_last", "channels_last"):
    model = bench.build_model(0).to(dev).train()
    x = torch.randn(B, 3, 1024, 1024, device=dev)
    if mode == "channels_last":
        model.model.pixel_level_module.encoder.to(memory_format=torch.channels_last)
    if mode != "nchw":  # "input_channels_last": only the activations change format, the parameters (and the optimiser state) stay
        x = x.contiguous(memory_format=torch.channels_last)
    ml, cl = bench.synthetic_labels(B, 1024, 1024, seed=0, device=dev)
    eng = DataParallelEngine(model, lr=5e-5)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
        eng.backward_and_step(out.loss)
        return out.loss.detach()

    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    print(json.dumps({"backbone": mode, "B": B, "ms_per_step": round((time.perf_counter() - t0) / 3 * 1e3, 2), "loss": round(float(loss), 4)}), flush=True)
    del model, eng
    torch.cuda.empty_cache()
