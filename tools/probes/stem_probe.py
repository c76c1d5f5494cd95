"""Which convolutions are the two 2.45 ms igemm launches of the forward?  Times the ResNet stem (7x7 / 2, 3 -> 64 at 1024^2, B 8) and
the pixel decoder's 3x3 convolutions alone, fp32."""
import json, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return round(ts[len(ts) // 2], 3)
x = torch.randn(8, 3, 1024, 1024, device=dev); w = torch.randn(64, 3, 7, 7, device=dev)
print(json.dumps({"conv": "stem 7x7/2 3->64 @1024", "ms": t(lambda: F.conv2d(x, w, None, 2, 3)), "gflop": 2 * 8 * 512 * 512 * 64 * 147 / 1e9}))
for name, cin, cout, hw in (("fpn 3x3 256->256 @256", 256, 256, 256), ("fpn 3x3 256->256 @128", 256, 256, 128), ("layer1 3x3 64->64 @256", 64, 64, 256), ("layer2 3x3 128->128 @128", 128, 128, 128), ("layer3 3x3 256->256 @64", 256, 256, 64), ("layer4 3x3 512->512 @32", 512, 512, 32)):
    xx = torch.randn(8, cin, hw, hw, device=dev); ww = torch.randn(cout, cin, 3, 3, device=dev)
    ms = t(lambda: F.conv2d(xx, ww, None, 1, 1))
    gf = 2 * 8 * hw * hw * cout * cin * 9 / 1e9
    print(json.dumps({"conv": name, "ms": ms, "gflop": gf, "tflops": round(gf / ms, 1)}))
