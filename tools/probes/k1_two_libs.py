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
