"""In-kernel timeline of the phased quad K1 kernel (variant 73 + wm2f_debug_stamps).

Prints, per stamp interval, the median / p10 / p90 over workgroups in s_memtime ticks (100 MHz on gfx950:
1 tick = 10 ns) and the workgroup's whole span."""
import argparse, ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weed_instance_segmentation_amd import _lib, ops
from weed_instance_segmentation_amd._lib import load, check
_lib.use_profiling_library()  # stamped kernels exist only in libwm2f_prof.so (include/wm2f_prof.h)

NAMES = ["setup+operand loads issue", "DMA issue", "wait coarse(+operands)", "softmax/coords", "barrier0", "phase0 gather",
         "wait mid", "barrier1", "phase1 gather", "wait fine", "barrier2", "phase2 gather", "slow+stores"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=73)
    ap.add_argument("--lanes", action="store_true", help="variant 74 on the lane-major operand rows (the model's inference path): sets WM2F_K1_STAMP=1 for the profiling library")
    ap.add_argument("--slab", type=int, default=0, help="with --lanes: WM2F_K1_MODE of a stamped slab-order build (807) on head-major rows")
    ap.add_argument("--init", action="store_true", help="the module's initial offset pattern instead of uniform offsets in [-4, 4]")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, D, L, P = 8, 8, 32, 3, 4
    shapes = [(32, 32), (64, 64), (128, 128)]
    S = sum(h * w for h, w in shapes)
    torch.manual_seed(0)
    value = torch.randn(B, S, H, D, device=dev)
    off = torch.rand(B, S, H, L, P, 2, device=dev) * 8.0 - 4.0
    if a.init:
        th = torch.arange(H, device=dev, dtype=torch.float32) * (2.0 * 3.141592653589793 / H)
        grid = torch.stack([th.cos(), th.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        off = (grid[None, None, :, None, None, :] * torch.arange(1, P + 1, device=dev, dtype=torch.float32)[None, None, None, None, :, None]
               ).expand(B, S, H, L, P, 2).contiguous()
    logits = torch.randn(B, S, H, L * P, device=dev)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")[::-1], -1).reshape(-1, 2)
                     for h, w in shapes]).to(dev)
    refl = ref[:, None, :].expand(S, L, 2).contiguous()
    if a.lanes:
        os.environ["WM2F_K1_STAMP"] = "1"
        lanes = ops.k1_lane_rows(off, logits)
        if a.slab:
            os.environ["WM2F_K1_MODE"] = str(a.slab)
            lanes = lanes.view(B, S, H, 36).permute(2, 0, 1, 3).contiguous()
    for _ in range(3):
        if a.lanes:
            ops.ms_deform_attn_fused_lanes(value, shapes, lanes, H, head_major=bool(a.slab))
        else:
            ops.ms_deform_attn_variant(value, shapes, off, logits, refl, fused=True, variant=a.variant)
    torch.cuda.synchronize()
    n_wg = B * H * 64 if a.variant == 73 else torch.cuda.get_device_properties(0).multi_processor_count * (2 if a.variant == 84 else 1)
    buf = np.zeros((min(n_wg, 8192), 160), dtype=np.int64)
    check(load().wm2f_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes), "wm2f_debug_stamps")
    if a.variant in (74, 84):  # streaming kernel: second tile of every workgroup, every wave: [wg][wave][slot]
        g_names = ["coords/softmax", "Bc wait", "gather coarse", "Bm wait", "gather mid", "Bf wait", "fetch next operands",
                   "gather fine", "slow+stores"]
        l_names = ["Bc wait", "issue fine A", "wait mid + Bm wait", "issue fine B + coarse(next) + wait fine", "Bf wait"]
        st = buf.reshape(buf.shape[0], 10, 16).astype(np.float64)
        pct = lambda x: [float(np.percentile(x, q)) for q in (10, 50, 90)]
        out = {"ticks": "shader cycles (s_memtime)", "workgroups": int(st.shape[0]), "per_wave": {}}
        ngw = 7 if a.variant == 84 else 8  # half-head form: 7 gather waves + 1 loader; full-head: 8 + 2
        loaders = [7] if a.variant == 84 else [8, 9]
        for w in range(ngw):
            dg = np.diff(st[:, w, :10], axis=1)
            out["per_wave"][f"gather{w}"] = {n: pct(dg[:, i])[1] for i, n in enumerate(g_names)}
            out["per_wave"][f"gather{w}"]["tile span"] = pct(st[:, w, 9] - st[:, w, 0])[1]
        for w in loaders:
            dl = np.diff(st[:, w, 10:16], axis=1)
            out["per_wave"][f"loader{w - ngw}"] = {n: pct(dl[:, i])[1] for i, n in enumerate(l_names)}
        # who reaches each barrier last (wave index -> share of workgroups), and how long the first arrival waited
        arrive = {"Bc": (1, 10), "Bm": (3, 12), "Bf": (5, 14)}  # (gather slot, loader slot) stamped just before the barrier
        for name, (gs, ls) in arrive.items():
            t = np.concatenate([st[:, :ngw, gs], st[:, loaders, ls]], axis=1)  # [wg][waves]
            last = t.argmax(1)
            out[f"{name}: last arrival by wave"] = {int(w): round(float((last == w).mean()), 3) for w in range(t.shape[1]) if (last == w).any()}
            out[f"{name}: first-to-last arrival spread"] = pct(t.max(1) - t.min(1))
        print(json.dumps(out, indent=1))
        return
    st = buf[:, :14].astype(np.float64)
    d = np.diff(st, axis=1)
    out = {"ticks": "s_memtime", "workgroups": int(st.shape[0])}
    for i, n in enumerate(NAMES):
        out[n] = [float(np.percentile(d[:, i], q)) for q in (10, 50, 90)]
    span = st[:, 13] - st[:, 0]
    out["span"] = [float(np.percentile(span, q)) for q in (10, 50, 90)]
    out["kernel_ticks"] = float(st[:, 13].max() - st[:, 0].min())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
