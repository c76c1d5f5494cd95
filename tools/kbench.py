#!/usr/bin/env python3
"""Per-kernel timing at the config-2 shapes (1024x1024, B=8, R50, Q=100, fp32) with HIP events.
Usage: python tools/kbench.py [--iters 20] [--only k1,k3,k2,mask,k4]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weed_instance_segmentation_amd import ops  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)  # us
    return dict(min_us=ts[0], med_us=ts[len(ts) // 2], mean_us=sum(ts) / len(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="k1,k1f,k3,mask,k2,k4")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--smooth", action="store_true", help="K1: slowly varying offsets instead of independent uniform ones")
    ap.add_argument("--init", action="store_true", help="K1: the module's initial offset pattern (point i of head h sits (i + 1) px along direction h, HF:2154-2166) -- what bench.py's random-init model feeds the kernel")
    ap.add_argument("--prof", action="store_true", help="load libwm2f_prof.so: K1 timing ablations (variant 44 ...), WM2F_K2_* / WM2F_K3_DBG environment knobs")
    ap.add_argument("--fine-only", action="store_true", help="K2: the finest level (N = 16384) only, so that a counter pass averages one shape")
    ap.add_argument("--lib", default=None, help="load this build of libwm2f.so instead (A/B of two builds of the library on one box, e.g. the previous commit's)")
    a = ap.parse_args()
    if a.lib:
        from weed_instance_segmentation_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    if a.prof:
        from weed_instance_segmentation_amd import _lib
        _lib.use_profiling_library()
    only = set(a.only.split(","))
    dev = torch.device("cuda:0")
    B, H, D, L, P, Q = a.B, 8, 32, 3, 4, 100
    shapes = [(32, 32), (64, 64), (128, 128)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cpu").manual_seed(0)
    res = {}
    if any(k in only for k in ("k1", "k1f", "k1v", "k1t", "k1q", "k1s", "k1b", "k1l", "k1o", "k1h")):
        value = torch.randn(B, S, H, D, device=dev)
        # reference points + the module's initial offset pattern (|offset| <= 4 px) + noise
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                         for h, w in shapes]).to(dev)  # (S,2) x,y
        # the module's initial pattern: |offset| <= 4 px (HF:2154-2166); uniform in [-4, 4]
        off = (torch.rand(B, S, H, L, P, 2, device=dev) * 8.0 - 4.0)
        if a.smooth:  # a slowly varying offset field (what a Linear of neighbouring tokens' features gives) + small noise
            pos = ref[None, :, None, None, None, :] * 6.283
            ph = torch.rand(1, 1, H, L, P, 2, device=dev) * 6.283
            off = 3.5 * torch.sin(pos * torch.tensor([1.0, 1.7], device=dev) + ph) + 0.1 * torch.randn(B, S, H, L, P, 2, device=dev)
        if a.init:
            th = torch.arange(H, device=dev, dtype=torch.float32) * (2.0 * 3.141592653589793 / H)
            grid = torch.stack([th.cos(), th.sin()], -1)
            grid = grid / grid.abs().max(-1, keepdim=True)[0]
            off = (grid[None, None, :, None, None, :] * torch.arange(1, P + 1, device=dev, dtype=torch.float32)[None, None, None, None, :, None]
                   ).expand(B, S, H, L, P, 2).contiguous()
        norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32)
        loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
        logits = torch.randn(B, S, H, L * P, device=dev)
        aw = torch.softmax(logits, -1).view(B, S, H, L, P).contiguous()
        nbytes = 4 * (value.numel() + loc.numel() + aw.numel() + value.numel())
        if "k1" in only:
            r = timeit(lambda: ops.ms_deform_attn(value, shapes, loc, aw), a.iters)
            r.update(bytes=nbytes, GBps=nbytes / r["med_us"] / 1e3)
            res["k1_msdeform_fwd"] = r
        refl = ref[:, None, :].expand(S, L, 2).contiguous()
        if "k1v" in only:
            for variant, margin in ((1, 4), (2, 4), (3, 4), (4, 4)) + (((44, 4),) if a.prof else ()):
                r = timeit(lambda: ops.ms_deform_attn_variant(value, shapes, loc, aw, variant=variant, margin=margin), a.iters)
                r.update(bytes=nbytes, GBps=nbytes / r["med_us"] / 1e3)
                res[f"k1_unfused_variant{variant}_margin{margin}"] = r
                r = timeit(lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=variant, margin=margin), a.iters)
                r.update(bytes=nbytes, GBps=nbytes / r["med_us"] / 1e3)
                res[f"k1_fused_variant{variant}_margin{margin}"] = r
        if "k1l" in only:  # fused forms: [offsets | logits] rows vs lane-major rows vs the round-1 loader schedule, interleaved rounds
            packed = torch.cat([off.reshape(B, S, -1), logits.reshape(B, S, -1)], -1).contiguous()
            lanes = ops.k1_lane_rows(off, logits)
            fns = {"packed_rows": lambda: ops.ms_deform_attn_fused_packed(value, shapes, packed, refl, H, L, P),
                   "lane_major_rows": lambda: ops.ms_deform_attn_fused_lanes(value, shapes, lanes, H),
                   "two_arrays": lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=4),
                   "two_arrays_half_head": lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=8),
                   "two_arrays_sched0": lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=7),
                   "two_arrays_strips": lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=6)}
            rounds = {k: [] for k in fns}
            for _ in range(5):
                for k_, fn in fns.items():
                    rounds[k_].append(timeit(fn, a.iters)["med_us"])
            for k_ in fns:
                ts = sorted(rounds[k_])
                res[f"k1_fused_{k_}"] = dict(min_us=ts[0], med_us=ts[len(ts) // 2], GBps=nbytes / ts[len(ts) // 2] / 1e3)
        if "k1o" in only:  # tile work order of the streaming kernel: 2-wide strips (4) against raster (6), interleaved rounds
            fns = {v: (lambda v=v: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=v, margin=4)) for v in (4, 6)}  # 4 raster, 6 strips
            rounds = {v: [] for v in fns}
            for _ in range(5):
                for v, fn in fns.items():
                    rounds[v].append(timeit(fn, a.iters)["med_us"])
            for v in fns:
                ts = sorted(rounds[v])
                res[f"k1_fused_variant{v}_rounds"] = dict(min_us=ts[0], med_us=ts[len(ts) // 2], GBps=nbytes / ts[len(ts) // 2] / 1e3)
        if "k1b" in only:  # backward
            vv, ll, ww_ = value.clone().requires_grad_(), loc.clone().requires_grad_(), aw.clone().requires_grad_()
            go = torch.randn(B, S, H * D, device=dev)
            def fb():
                o = ops.ms_deform_attn(vv, shapes, ll, ww_)
                o.backward(go)
                vv.grad = ll.grad = ww_.grad = None
            res["k1_fwd_plus_bwd"] = timeit(fb, a.iters)
            res["k1_bwd_float_atomics"] = timeit(lambda: ops.ms_deform_attn_bwd(value, shapes, loc, aw, go, deterministic=False), a.iters)
            res["k1_bwd_deterministic"] = timeit(lambda: ops.ms_deform_attn_bwd(value, shapes, loc, aw, go, deterministic=True), a.iters)
        if any(k in only for k in ("k1t", "k1q", "k1s", "k1h")):  # one variant only, for PMC runs (LDS-window / phased quad / streaming half-head / full-head)
            vv_ = 2 if "k1t" in only else (3 if "k1q" in only else (8 if "k1h" in only else 4))  # k1h: half-head form
            r = timeit(lambda: ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=vv_, margin=4), a.iters)
            r.update(bytes=nbytes, GBps=nbytes / r["med_us"] / 1e3)
            res[f"k1_fused_variant{vv_}_margin4"] = r
        if "k1f" in only:
            r = timeit(lambda: ops.ms_deform_attn_fused(value, shapes, off, logits, refl), a.iters)
            r.update(bytes=nbytes, GBps=nbytes / r["med_us"] / 1e3)
            res["k1_msdeform_fused_fwd"] = r
    if "pp" in only:  # instance post-processing at the reference's eval shape (Q = 100, 256^2 logits -> 1024^2 targets)
        import time
        from types import SimpleNamespace
        from weed_instance_segmentation_amd.postprocess import Mask2FormerInstancePostProcessor
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import m2f_oracle as O
        gg = torch.Generator().manual_seed(5)
        low = torch.randn(B, Q, 8, 8, generator=gg) * 3.0 - 2.0
        masks = torch.nn.functional.interpolate(low, size=(256, 256), mode="bicubic", align_corners=False)
        cls = torch.randn(B, Q, 4, generator=gg) * 3.0
        ts = [(1024, 1024)] * B
        out = SimpleNamespace(class_queries_logits=cls.to(dev), masks_queries_logits=masks.to(dev))
        proc = Mask2FormerInstancePostProcessor()
        proc.post_process_instance_segmentation(out, threshold=0.5, target_sizes=ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = proc.post_process_instance_segmentation(out, threshold=0.5, target_sizes=ts)
        torch.cuda.synchronize()
        t_hip = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        ref = O.post_process_instance_segmentation(cls, masks, 0.5, ts)
        t_cpu = time.perf_counter() - t0
        res["postprocess_instances"] = {"hip_ms_per_batch": t_hip * 1e3, "oracle_cpu_ms_per_batch": t_cpu * 1e3, "B": B,
                                        "kept": sum(len(x["segments_info"]) for x in r), "cpu_threads": torch.get_num_threads()}
    if "k3" in only or "mask" in only:
        emb = torch.randn(B, Q, 256, device=dev)
        pix = torch.randn(B, 256, 256, 256, device=dev)
        flop = 2 * B * Q * 256 * 65536
        if "k3" in only:
            r = timeit(lambda: ops.mask_einsum(emb, pix), a.iters)
            r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6, bytes=4 * (emb.numel() + pix.numel() + B * Q * 65536))
            res["k3_mask_einsum_fwd"] = r
            r2 = timeit(lambda: torch.einsum("bqc,bchw->bqhw", emb, pix), a.iters)
            r2.update(TFLOPs=flop / r2["med_us"] / 1e6)
            res["k3_torch_einsum_ref"] = r2
        if "mask" in only:
            logits = ops.mask_einsum(emb, pix)
            for hw in shapes:
                res[f"attn_mask_build_{hw[0]}"] = timeit(lambda: ops.attn_mask_build(logits, hw), a.iters)
    if "k1c5" in only:  # K1 at BASELINE config 5's level sizes (1333 x 800 unpadded: NOT 1:2:4): streaming kernel against its fallbacks
        sh5 = [(25, 42), (50, 84), (100, 167)]
        S5 = sum(h * w for h, w in sh5)
        value = torch.randn(B, S5, H, D, device=dev)
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                         for h, w in sh5]).to(dev)
        refl = ref[:, None, :].expand(S5, L, 2).contiguous()
        off = torch.rand(B, S5, H, L, P, 2, device=dev) * 8.0 - 4.0
        logits = torch.randn(B, S5, H, L * P, device=dev)
        nb5 = 4 * (2 * value.numel() + off.numel() + logits.numel())
        for variant, name in ((4, "streaming_general"), (2, "lds_window_kernel"), (1, "direct_gather")):
            r = timeit(lambda: ops.ms_deform_attn_variant(value, sh5, off, logits, refl, fused=True, variant=variant), a.iters)
            r.update(bytes=nb5, GBps=nb5 / r["med_us"] / 1e3)
            res[f"k1_config5_levels_{name}"] = r
    if "tg" in only:  # token GEMMs of the encoder layers: wm2f_token_linear_fwd against the library (F.linear) + the separate LayerNorm pass
        import torch.nn.functional as F
        Mtok = B * S
        for name, K, N, ln in (("value_proj", 256, 256, False), ("offsets_logits", 256, 288, False), ("output_proj+ln", 256, 256, True),
                               ("fc2+ln+pos", 1024, 256, True)):
            x = torch.randn(Mtok, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            bb = torch.randn(N, device=dev)
            res_ = torch.randn(Mtok, N, device=dev) if ln else None
            gm, bt = torch.randn(N, device=dev), torch.randn(N, device=dev)
            pe = torch.randn(S, N, device=dev) if "pos" in name else None
            flop = 2.0 * Mtok * K * N
            r = timeit(lambda: ops.token_linear(x, w, bb, residual=res_, ln=(gm, bt, 1e-5) if ln else None, pos=pe), a.iters)
            r.update(TFLOPs=flop / r["med_us"] / 1e6)
            res[f"token_linear_{name}"] = r
            if ln:
                r = timeit(lambda: ops.add_layernorm(F.linear(x, w, bb), res_, gm, bt, 1e-5, pos=pe), a.iters)
            else:
                r = timeit(lambda: F.linear(x, w, bb), a.iters)
            r.update(TFLOPs=flop / r["med_us"] / 1e6)
            res[f"library_{name}"] = r
    if "k3m" in only:  # K3 with the fused attention-mask epilogue at the three level resolutions (the inference route)
        emb = torch.randn(B, Q, 256, device=dev)
        for hw in shapes:
            pl = torch.randn(B, 256, hw[0], hw[1], device=dev)
            flop = 2 * B * Q * 256 * hw[0] * hw[1]
            r = timeit(lambda: ops.mask_einsum_attn_mask(emb, pl), a.iters)
            r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6)
            res[f"k3_fused_attn_mask_hw{hw[0] * hw[1]}"] = r
            r = timeit(lambda: ops.attn_mask_build(ops.mask_einsum(emb, pl), hw), a.iters)
            r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6)
            res[f"k3_two_launch_route_hw{hw[0] * hw[1]}"] = r
        pix = torch.randn(B, 256, 256, 256, device=dev)
        r = timeit(lambda: ops.resize_pyramid(pix), a.iters)
        r.update(bytes=4 * pix.numel() * 85 // 64, GBps=4 * pix.numel() * 85 / 64 / r["med_us"] / 1e3)
        res["resize_pyramid_all_three"] = r
        for hw in shapes:
            r = timeit(lambda: ops.resize_bilinear(pix, hw), a.iters)
            r.update(bytes=4 * (pix.numel() + B * 256 * hw[0] * hw[1]))
            res[f"resize_mask_features_to_{hw[0]}"] = r
    if "k3g" in only:  # K3 backward: hand-written pair against the two batched library GEMMs autograd derives
        emb = torch.randn(B, Q, 256, device=dev)
        pix = torch.randn(B, 256, 256, 256, device=dev)
        go = torch.randn(B, Q, 256, 256, device=dev)
        flop = 2 * B * Q * 256 * 65536
        for name, we, wp in (("both", True, True), ("g_emb", True, False), ("g_pix", False, True)):
            r = timeit(lambda: ops.mask_einsum_bwd(emb, pix, go, we, wp), a.iters)
            r.update(flop=flop * (we + wp), TFLOPs=flop * (we + wp) / r["med_us"] / 1e6)
            res[f"k3_backward_{name}"] = r
        gof, pf = go.reshape(B, Q, -1), pix.reshape(B, 256, -1)
        r = timeit(lambda: torch.bmm(gof, pf.transpose(1, 2)), a.iters)
        r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6)
        res["library_bmm_g_emb"] = r
        r = timeit(lambda: torch.bmm(emb.transpose(1, 2), gof), a.iters)
        r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6)
        res["library_bmm_g_pix"] = r
    if "k3gb" in only:  # K3 backward under bf16 autocast: hand-written pair against .to(bf16) + two batched library GEMMs
        emb = torch.randn(B, Q, 256, device=dev).to(torch.bfloat16)
        pix = torch.randn(B, 256, 256, 256, device=dev).to(torch.bfloat16)
        pix_t = ops.nchw_to_pixel_major_bf16(pix)
        go = torch.randn(B, Q, 256, 256, device=dev)
        for name, we, wp in (("both", True, True), ("g_emb", True, False), ("g_pix", False, True)):
            r = timeit(lambda: ops.mask_einsum_bf16_bwd(emb, pix, go, we, wp), a.iters)
            nb = go.numel() * 4 * (we + wp) + (pix.numel() * 2 if we else 0) + (pix.numel() * 2 if wp else 0)
            r.update(bytes=nb, GBps=nb / r["med_us"] / 1e3)
            res[f"k3_bf16_backward_{name}"] = r

        def lib():
            gb = go.reshape(B, Q, -1).to(torch.bfloat16)
            return torch.bmm(gb, pix_t), torch.bmm(emb.transpose(1, 2), gb)
        res["library_bf16_cast_and_two_bmm"] = timeit(lib, a.iters)
    if "k3b" in only:  # K3 on the bf16 matrix cores (bf16 autocast path): HBM-bound
        emb = torch.randn(B, Q, 256, device=dev).to(torch.bfloat16)
        pix = torch.randn(B, 256, 256, 256, device=dev).to(torch.bfloat16)
        r = timeit(lambda: ops.nchw_to_pixel_major_bf16(pix), a.iters)
        r.update(bytes=2 * pix.numel() * 2, GBps=2 * pix.numel() * 2 / r["med_us"] / 1e3)
        res["k3_bf16_pixel_major_transpose"] = r
        pix_t = ops.nchw_to_pixel_major_bf16(pix)
        r = timeit(lambda: ops.mask_einsum_bf16(emb, pix, pix_t), a.iters)
        nb = pix.numel() * 2 + emb.numel() * 2 + B * Q * 65536 * 4
        r.update(bytes=nb, GBps=nb / r["med_us"] / 1e3, flop=2 * B * Q * 256 * 65536)
        res["k3_bf16_mask_einsum"] = r
    if "k2" in only:
        E = H * D
        q = torch.randn(B, Q, E, device=dev) * 0.3
        for hw in (shapes[-1:] if a.fine_only else shapes):
            N = hw[0] * hw[1]
            k = torch.randn(B, N, E, device=dev)
            v = torch.randn(B, N, E, device=dev)
            mask = (torch.rand(B, Q, N, device=dev) < 0.5).to(torch.uint8)
            ro = torch.ones(B, Q, device=dev, dtype=torch.int32)
            r = timeit(lambda: ops.masked_xattn(q, k, v, mask, ro, H), a.iters)
            flop = 4 * B * H * Q * N * D
            r.update(flop=flop, TFLOPs=flop / r["med_us"] / 1e6, kv_bytes=2 * B * N * E * 4)
            res[f"k2_masked_xattn_N{N}"] = r
    if "k2b" in only:  # K2 forward + backward through autograd (the backward alone = difference to the k2 line); --prof: WM2F_K2_BWD_FULL=0
        E = H * D
        for hw in (shapes[-1:] if a.fine_only else shapes):
            N = hw[0] * hw[1]
            q = (torch.randn(B, Q, E, device=dev) * 0.3).requires_grad_()
            k = torch.randn(B, N, E, device=dev).requires_grad_()
            v = torch.randn(B, N, E, device=dev).requires_grad_()
            mask = (torch.rand(B, Q, N, device=dev) < 0.5).to(torch.uint8)
            ro = torch.ones(B, Q, device=dev, dtype=torch.int32)
            go = torch.randn(B, Q, E, device=dev)

            def fb():
                o = ops.masked_xattn(q, k, v, mask, ro, H)
                o.backward(go)
                q.grad = k.grad = v.grad = None
            res[f"k2_fwd_plus_bwd_N{N}"] = timeit(fb, a.iters)
    if "k4" in only:
        NL, Tn, Pn = 10, 16, 12544
        ml = torch.randn(NL, B, Q, 256, 256, device=dev)
        cl = torch.randn(NL, B, Q, 4, device=dev)
        tgt = (torch.rand(B * Tn, 1024, 1024, device=dev) < 0.1).float()
        cls = torch.randint(0, 3, (B * Tn,), device=dev)
        pts = torch.rand(NL, B, Pn, 2, device=dev)
        res["k4_matcher_cost_all_levels_f32tgt"] = timeit(lambda: ops.matcher_cost(ml, cl, tgt, [Tn] * B, cls, pts, 2., 5., 5.), max(3, a.iters // 4))
        tgt8 = tgt.to(torch.uint8)
        res["k4_matcher_cost_all_levels_u8tgt"] = timeit(lambda: ops.matcher_cost(ml, cl, tgt8, [Tn] * B, cls, pts, 2., 5., 5.), max(3, a.iters // 4))
    for k, v in res.items():
        print(json.dumps({"kernel": k, **{kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()}}))


if __name__ == "__main__":
    main()
