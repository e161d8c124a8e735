"""Streaming GEMV: bf16 weights vs weight-only fp8 (e4m3fn codes, 8-byte loads per lane), eager back-to-back launches on fresh weights."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, dual, f8, iters=200):
    x = torch.randn(m, k, device="cuda")
    byts = n * k * (1 if f8 else 2) * (2 if dual else 1)
    nb = max(2, int(400e6 // byts))
    mk = (lambda: torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda")) if f8 else (lambda: (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16())
    ws = [(mk(), mk() if dual else None) for _ in range(nb)]
    sc = torch.ones(n, device="cuda"); out = torch.zeros(m, n, device="cuda"); nw = torch.ones(k, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m, a.n, a.k, a.out, a.ldo = x.data_ptr(), k, m, n, k, out.data_ptr(), n
    a.wdt = L.VV_FP8 if f8 else L.VV_BF16
    a.wscale = a.w2scale = sc.data_ptr()
    if dual: a.act, a.pro, a.norm_w, a.eps = 2, 1, nw.data_ptr(), 1e-6
    s = torch.cuda.current_stream().cuda_stream
    def run(i):
        a.w = ws[i % nb][0].data_ptr()
        if dual: a.w2 = ws[i % nb][1].data_ptr()
        L.check(lib.vv_linear(C.byref(a), s), "lin")
    for i in range(10): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"m={m} n={n:5d} k={k:5d} dual={int(dual)} {'fp8 ' if f8 else 'bf16'}: {us:7.2f} us  {byts/us/1e3:7.1f} GB/s", flush=True)
for shp in ((2, 4608, 1536, True), (2, 1536, 4608, False), (2, 8960, 1536, True), (2, 1536, 8960, False), (2, 2048, 1536, False), (1, 8192, 2048, False), (1, 2048, 8192, False)):
    bench(*shp, False); bench(*shp, True)
