import torch, time
x = torch.randn(344064, 256, device="cuda", dtype=torch.bfloat16)
w = torch.randn(256, 256, device="cuda", dtype=torch.bfloat16)
b = torch.randn(256, device="cuda", dtype=torch.float32)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
try:
    us, r = t(lambda: torch.mm(x, w.t(), out_dtype=torch.float32)); print("mm out_dtype f32", us, r.dtype)
except Exception as e: print("mm out_dtype:", repr(e)[:200])
try:
    us, r = t(lambda: torch.addmm(b, x, w.t(), out_dtype=torch.float32)); print("addmm f32 bias out_dtype f32", us, r.dtype)
except Exception as e: print("addmm out_dtype (f32 bias):", repr(e)[:200])
try:
    us, r = t(lambda: torch.addmm(b.bfloat16(), x, w.t(), out_dtype=torch.float32)); print("addmm bf16 bias out_dtype f32", us, r.dtype)
except Exception as e: print("addmm out_dtype (bf16 bias):", repr(e)[:200])
us, r = t(lambda: torch.nn.functional.linear(x, w, b.bfloat16())); print("linear bf16", us)
us, r = t(lambda: torch.nn.functional.linear(x, w, b.bfloat16()).float()); print("linear bf16 + .float()", us)
