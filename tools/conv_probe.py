import torch, time
x = torch.randn(8, 256, 256, 256, device="cuda")
w = torch.randn(64, 256, 1, 1, device="cuda") * 0.05
b = torch.randn(64, device="cuda")
w3 = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
x3 = torch.randn(8, 64, 256, 256, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
with torch.no_grad():
    print("conv1x1+bias+relu  ", t(lambda: torch.relu(torch.nn.functional.conv2d(x, w, b))))
    print("conv1x1 nobias     ", t(lambda: torch.nn.functional.conv2d(x, w)))
    try:
        y = torch.ops.aten.miopen_convolution_relu(x, w, b, [1,1], [0,0], [1,1], 1)
        ref = torch.relu(torch.nn.functional.conv2d(x, w, b))
        print("fused ok, max err", (y-ref).abs().max().item())
        print("miopen_conv_relu 1x1", t(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1,1], [0,0], [1,1], 1)))
    except Exception as e:
        print("miopen_convolution_relu failed:", repr(e)[:300])
    print("conv3x3+bias+relu  ", t(lambda: torch.relu(torch.nn.functional.conv2d(x3, w3, b, 1, 1))))
    try:
        print("miopen_conv_relu 3x3", t(lambda: torch.ops.aten.miopen_convolution_relu(x3, w3, b, [1,1], [1,1], [1,1], 1)))
        z = torch.randn(8, 64, 256, 256, device="cuda")
        y = torch.ops.aten.miopen_convolution_add_relu(x3, w3, z, 1.0, b, [1,1], [1,1], [1,1], 1)
        ref = torch.relu(torch.nn.functional.conv2d(x3, w3, b, 1, 1) + z)
        print("add_relu ok, err", (y-ref).abs().max().item())
        print("miopen_conv_add_relu 3x3", t(lambda: torch.ops.aten.miopen_convolution_add_relu(x3, w3, z, 1.0, b, [1,1], [1,1], [1,1], 1)))
    except Exception as e:
        print("3x3 fused failed:", repr(e)[:300])
    # channels_last
    xc = x.to(memory_format=torch.channels_last); wc = w.to(memory_format=torch.channels_last)
    print("conv1x1 channels_last+bias+relu", t(lambda: torch.relu(torch.nn.functional.conv2d(xc, wc, b))))
    # GEMM N padding probe
    h = torch.randn(172032, 256, device="cuda")
    for N in (192, 288, 320, 384, 512):
        W = torch.randn(N, 256, device="cuda"); bb = torch.randn(N, device="cuda")
        ms = t(lambda: torch.nn.functional.linear(h, W, bb))
        print(f"linear N={N}: {ms*1e3:.0f} us  {2*172032*256*N/ms/1e9:.1f} TF")
    W = torch.randn(1024, 256, device="cuda"); bb = torch.randn(1024, device="cuda")
    print("fc1 linear+relu", t(lambda: torch.relu(torch.nn.functional.linear(h, W, bb)))*1e3, "us")
    print("fc1 addmm_act  ", t(lambda: torch._addmm_activation(bb, h, W.t(), use_gelu=False))*1e3, "us")
