#!/usr/bin/env python3
"""Headline benchmark: images/sec of the Mask2Former hot path at 1024x1024, batch 8 per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: synthetic 1024x1024 3-class inputs, ResNet-50 Mask2Former,
100 queries, fp32 forward-only, batch 8 per GPU, random-init weights (seed 0).  A step is one pass
of the whole forward over one batch already resident in HBM.  N > 1: the image batch is sharded,
one process per GPU with its own batch of 8 (weak scaling, no data-path collective in the
forward-only workload; `--mode train` adds the RCCL gradient all-reduce).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant hand-written kernel (K1, MSDeformAttn): algorithmic bytes per launch /
                mean launch duration measured with HIP events inside the timed region.
  roofline_k3   same for the mask einsum against the fp32 MFMA peak.
  cpu_baseline  the CPU oracle (oracle/m2f_oracle.py, kind "port") timed on this host on a
                bounded sample of the same workload (N = 1, rank 0 only).
  train_step    BASELINE.json configs[2] beside the headline line (N = 1, default flags only): a short bf16-autocast
                full train step leg at batch 16 (Hungarian matching + losses + backward + AdamW), run AFTER the timed
                forward region and the CPU sample -- ms per step, images/sec and the K1 / K2 / K3 launch times.
                `value` / `metric` stay configs[1].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (spec)
# HBM bytes per K1 launch from the PMC counters of a separate rocprofv3 run over this same command (tools/pmc_bench.sh;
# counters cannot be collected inside the driver's own run): the newest committed measurement
K1_TRAFFIC_PROFILE = "r03_pmc_k1_traffic.json"


def build_model(seed=0, num_labels=3, num_queries=100):
    from weed_instance_segmentation_amd import Mask2FormerConfig, Mask2FormerForUniversalSegmentation
    torch.manual_seed(seed)
    cfg = Mask2FormerConfig(num_labels=num_labels, num_queries=num_queries)  # ResNet-50 backbone by default
    return Mask2FormerForUniversalSegmentation(cfg)


def synthetic_labels(B, H, W, T=16, seed=0, device="cpu"):
    """T axis-aligned random rectangles per image (side 32..256), classes uniform in {0,1,2}."""
    import numpy as np
    rng = np.random.default_rng(seed)
    ml, cl = [], []
    for _ in range(B):
        m = torch.zeros(T, H, W)
        for t in range(T):
            hh, ww = rng.integers(32, 257, 2)
            y0, x0 = rng.integers(0, H - hh + 1), rng.integers(0, W - ww + 1)
            m[t, y0:y0 + hh, x0:x0 + ww] = 1.0
        ml.append(m.to(device))
        cl.append(torch.as_tensor(rng.integers(0, 3, T), dtype=torch.int64, device=device))
    return ml, cl


def cpu_baseline(model, size, n_images, seed=0):
    """Time the CPU oracle (same weights, same input distribution) on `n_images` images."""
    from oracle import m2f_oracle as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = model.config.to_dict()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_images, 3, size, size, generator=g)
    # the GPU box exposes every host core but a 1-GPU job owns a share of 16 (gpurun rules); more
    # threads than that only oversubscribe (measured: 256 threads -> 16x slower)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    t0 = time.perf_counter()
    res = O.forward(sd, cfg, x)
    dt = time.perf_counter() - t0
    return dict(value=n_images / dt, unit="images/sec", cores=torch.get_num_threads(), kind="port",
                sample=f"{n_images} image(s) {size}x{size}, one batched oracle forward, fp32, {dt:.2f} s",
                seconds=dt), res, x


def train_leg(dev, batch=16, size=1024, warmup=2, steps=3, amp=True, seed=0):
    """BASELINE.json configs[2]: full train step under bf16 autocast at `batch` images of `size`^2, synthetic labels
    (16 rectangles per image), AdamW lr 5e-5 (train.py:174).  A fresh model; HIP-event launch times of the wm2f kernels."""
    from weed_instance_segmentation_amd import ops
    from weed_instance_segmentation_amd.parallel import DataParallelEngine
    model = build_model(seed).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(2000)
    x = torch.randn(batch, 3, size, size, generator=g).to(dev)
    ml, cl = synthetic_labels(batch, size, size, seed=0, device=dev)
    engine = DataParallelEngine(model, lr=5e-5)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
        engine.backward_and_step(out.loss)
        return out.loss.detach()

    for _ in range(warmup):
        loss = step()
    timer = ops.KernelTimer()
    torch.cuda.synchronize()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    ks = timer.summary()
    return {"workload": f"BASELINE.json configs[2]: synthetic {size}x{size} 3-class, ResNet-50 Mask2Former, 100 queries, "
                        f"{'bf16 autocast' if amp else 'fp32'} full train step (Hungarian matching + mask/dice/class loss + backward + AdamW), bs={batch}",
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "images_per_sec": round(batch * steps / dt, 2),
            "loss": round(float(loss), 4), "finite": bool(torch.isfinite(loss)),
            "kernels_us": {k: {"launches_per_step": n // steps, "avg_us": round(us, 1)} for k, (n, us) in sorted(ks.items())}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--mode", choices=["fwd", "train"], default="fwd")
    ap.add_argument("--cpu-images", type=int, default=2, help="images in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--train-leg", type=int, default=1, help="1: after the default run also time a short configs[2] train leg (0 = skip)")
    ap.add_argument("--amp", choices=["off", "bf16"], default="off",
                    help="bf16: run the step under torch.autocast(bfloat16) (BASELINE configs 2-4)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm

    from weed_instance_segmentation_amd import ops
    model = build_model(0).to(dev)
    B, S = a.batch, a.size
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    x = torch.randn(B, 3, S, S, generator=g).to(dev)

    if a.mode == "fwd":
        model.eval()

        def step():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.amp == "bf16"):
                return model(pixel_values=x)
    else:
        from weed_instance_segmentation_amd.parallel import DataParallelEngine
        model.train()
        ml, cl = synthetic_labels(B, S, S, seed=rank, device=dev)
        engine = DataParallelEngine(model, lr=5e-5)

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.amp == "bf16"):
                out = model(pixel_values=x, mask_labels=ml, class_labels=cl)
            engine.backward_and_step(out.loss)
            return out.loss.detach()

    for _ in range(a.warmup):
        step()
    timer = ops.KernelTimer()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ksum = timer.summary()
        value = world * B * a.steps / dt
        prec = "fp32" if a.amp == "off" else "bf16 autocast"
        what = "forward-only" if a.mode == "fwd" else "full train step (Hungarian matching + mask/dice/class loss + backward + AdamW)"
        # which BASELINE.json configuration this run is (the default flags are configs[1], the headline metric)
        if (a.mode, a.amp, B, S) == ("fwd", "off", 8, 1024):
            which = "BASELINE.json configs[1]"
        elif (a.mode, a.amp, B, S) == ("train", "bf16", 16, 1024):
            which = "BASELINE.json configs[2]"
        else:
            which = "variant of BASELINE.json configs[1] (not a BASELINE configuration)"
        line = {
            "metric": f"images/sec at {S}x{S} bs={B} per GPU (Mask2Former R50, 100 queries, {prec} {what.split(' (')[0]})",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.amp == "off" else "bf16 autocast (stock ops, K2, K3 and the token weight gradients on bf16 operands; K1, matcher, point sampler compute fp32)", "data": "synthetic (randn pixels, random-init weights seed 0)",
            "config": {"workload": f"{which}: synthetic {S}x{S} 3-class, ResNet-50 Mask2Former, 100 queries, {prec} {what}, "
                                   f"bs={B} per GPU", "global_batch": world * B, "image_size": S,
                       "parallelism": f"dp{world}", "mode": a.mode, "amp": a.amp},
        }
        # which route built the attention masks: named by the launches that actually ran
        line["config"]["attention_masks"] = ("einsum at level resolution for the 9 intermediate predictions "
                                             "(mask bits from the MFMA epilogue, no logits written), full resolution for the returned one"
                                             if any(k.startswith("mask_einsum_attn_mask") for k in ksum)
                                             else "every prediction at full resolution")
        # ---- roofline of the dominant hand-written kernel (K1) and of K3
        L, P, H, D, Q = 3, 4, 8, 32, 100
        Stok = sum((S // s) ** 2 for s in (32, 16, 8))
        k1_name = "msdeform_fused_fwd" if "msdeform_fused_fwd" in ksum else "msdeform_fwd"
        if k1_name in ksum:
            n, us = ksum[k1_name]
            nbytes = 4 * (2 * B * Stok * H * D + B * Stok * H * L * P * 3)
            ach = nbytes / us / 1e3
            traffic = None  # HBM bytes per launch from PMC counters of a separate rocprofv3 run (profiles/)
            tpath = os.path.join(ROOT, "profiles", K1_TRAFFIC_PROFILE)
            if B == 8 and S == 1024 and os.path.exists(tpath):
                traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
            line["roofline"] = {"kernel": k1_name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                                "algorithmic_bytes_per_launch": nbytes, "launches": n, "avg_us": round(us, 2)}
        if "mask_einsum_fwd" in ksum:
            n, us = ksum["mask_einsum_fwd"]
            flop = 2 * B * Q * 256 * (S // 4) ** 2
            ach = flop / us / 1e6
            line["roofline_k3"] = {"kernel": "mask_einsum_fwd", "bound": "mfma", "achieved": round(ach, 2),
                                   "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                                   "algorithmic_flop_per_launch": flop, "launches": n, "avg_us": round(us, 2)}
        line["kernels_us"] = {k: {"launches": n, "avg_us": round(us, 2)} for k, (n, us) in sorted(ksum.items())}
        if world == 1 and a.cpu_images > 0:
            cb, res, xc = cpu_baseline(model, S, a.cpu_images)
            # the same sample through the HIP path, as a last parity check beside the timing
            model.eval()
            with torch.no_grad():
                o = model(pixel_values=xc.to(dev))
            ref = res["masks_queries_logits"]
            cb["mask_logit_max_abs_err"] = float((o.masks_queries_logits.cpu() - ref).abs().max())
            cb["mask_logit_rel_err"] = cb["mask_logit_max_abs_err"] / float(ref.abs().max())
            line["cpu_baseline"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in cb.items()}
        if world == 1 and a.train_leg and which == "BASELINE.json configs[1]":
            torch.cuda.empty_cache()  # 288 GB of HBM: the forward model simply stays resident beside the train leg's
            line["train_step"] = train_leg(dev)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
