"""Same-box A/B of wm2f_msdeform_fused_packed_fwd from two builds of the library (e.g. a previous round's) on the operands the
random-init model feeds it: offsets = the module's initial grid (point i of head h at (i + 1) px along direction h, HF:2154-2166),
logits = 0, for every token.  Usage: python tools/probes/k1_two_libs.py libA.so libB.so [--iters 30]"""
import ctypes, json, math, sys
import torch

def main():
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    iters = 30
    dev = torch.device("cuda:0")
    B, H, D, L, P = 8, 8, 32, 3, 4
    shapes = [(32, 32), (64, 64), (128, 128)]
    S = sum(h * w for h, w in shapes)
    torch.manual_seed(0)
    value = torch.randn(B, S, H, D, device=dev)
    th = torch.arange(H, dtype=torch.float32) * (2.0 * math.pi / H)
    g = torch.stack([th.cos(), th.sin()], -1)
    g = g / g.abs().max(-1, keepdim=True)[0]
    off = (g.view(H, 1, 1, 2) * torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)).expand(H, L, P, 2)
    packed = torch.cat([off.reshape(-1), torch.zeros(H * L * P)]).to(dev).expand(B, S, -1).contiguous()
    out = torch.empty(B, S, H * D, device=dev)
    hw = (ctypes.c_int32 * (2 * L))(*[x for s in shapes for x in s])
    fns = []
    for p in libs:
        lib = ctypes.CDLL(p)
        f = lib.wm2f_msdeform_fused_packed_fwd
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 9 + [ctypes.c_void_p]
        fns.append((p, f))
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for rep in range(7):
        for name, f in fns:
            call = lambda: f(value.data_ptr(), packed.data_ptr(), out.data_ptr(), hw, B, S, S, H, D, L, P, 0, 4, st)
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            res.setdefault(name, []).append(round(ts[len(ts) // 2], 2))
    for k, v in res.items():
        print(json.dumps({"lib": k, "k1_fused_packed_init_operands_med_us": v}))
    if "--lanes" in sys.argv:  # the LAST library's lane-major entry point (what the model runs) on the same operands
        lib = ctypes.CDLL(libs[-1])
        f = lib.wm2f_msdeform_fused_lanes_fwd
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
        row = torch.zeros(H, P, 9)
        for l in range(L):
            row[:, :, 2 * l:2 * l + 2] = off[:, l]  # (H, P, 2)
        rows = row.reshape(-1).to(dev).expand(B, S, -1).contiguous()
        call = lambda: f(value.data_ptr(), rows.data_ptr(), out.data_ptr(), hw, B, S, S, H, D, L, P, 0, st)
        meds = []
        for rep in range(7):
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            meds.append(round(ts[len(ts) // 2], 2))
        print(json.dumps({"lib": libs[-1], "k1_fused_lanes_init_operands_med_us": meds}))
        import os
        if os.environ.get("WM2F_K1_HEAD_MAJOR") == "1":  # profiling build: the same rows stored (heads, B, S, 36); random values in both
            torch.manual_seed(1)
            tm = (torch.randn(B, S, H, 36, device=dev) * 2).contiguous()
            hm = tm.permute(2, 0, 1, 3).contiguous()
            o2 = torch.empty_like(out)
            os.environ["WM2F_K1_HEAD_MAJOR"] = "0"
            assert f(value.data_ptr(), tm.data_ptr(), out.data_ptr(), hw, B, S, S, H, D, L, P, 0, st) == 0
            os.environ["WM2F_K1_HEAD_MAJOR"] = "1"
            assert f(value.data_ptr(), hm.data_ptr(), o2.data_ptr(), hw, B, S, S, H, D, L, P, 0, st) == 0
            torch.cuda.synchronize()
            print(json.dumps({"head_major_equals_token_major": bool(torch.equal(out, o2))}))
            res2 = {"token_major": [], "head_major": []}
            for rep in range(7):
                for name, arr, flag in (("token_major", rows, "0"), ("head_major", rows.view(B, S, H, 36).permute(2, 0, 1, 3).contiguous(), "1")):
                    os.environ["WM2F_K1_HEAD_MAJOR"] = flag
                    call = lambda: f(value.data_ptr(), arr.data_ptr(), out.data_ptr(), hw, B, S, S, H, D, L, P, 0, st)
                    for _ in range(5):
                        assert call() == 0
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(iters):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(); call(); b.record(); torch.cuda.synchronize()
                        ts.append(a.elapsed_time(b) * 1e3)
                    ts.sort()
                    res2[name].append(round(ts[len(ts) // 2], 2))
            print(json.dumps(res2))
    if "--variants" in sys.argv:  # the LAST library's streaming-kernel variants (two-array fused form) on the same operands
        lib = ctypes.CDLL(libs[-1])
        f = lib.wm2f_msdeform_fwd_v
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 11 + [ctypes.c_void_p]
        offs = off.reshape(-1).to(dev).expand(B, S, -1).contiguous()
        logits = torch.zeros(B, S, H * L * P, device=dev)
        ys = [torch.linspace(0.5, hh - 0.5, hh) / hh for hh, ww in shapes]
        xs = [torch.linspace(0.5, ww - 0.5, ww) / ww for hh, ww in shapes]
        ref = torch.cat([torch.stack(torch.meshgrid(y, x, indexing="ij")[::-1], -1).reshape(-1, 2) for y, x in zip(ys, xs)]).to(dev).contiguous()
        for variant in (4, 7, 6, 4, 7, 6, 4, 7, 6):
            call = lambda: f(value.data_ptr(), offs.data_ptr(), logits.data_ptr(), ref.data_ptr(), out.data_ptr(), hw, B, S, S, H, D, L, P, 0, 1, variant, 4, st)
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); call(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            print(json.dumps({"lib": libs[-1], "variant": variant, "two_array_fused_init_operands_med_us": round(ts[len(ts) // 2], 2)}))

if __name__ == "__main__":
    main()
