"""Long-K down-projection GEMV (m=2 n=1536 k=8960, residual epilogue): waves that split K x persistent grid size."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, iters=300):
    x = torch.randn(m, k, device="cuda"); res = torch.randn(m, n, device="cuda")
    byts = n * k * 2
    nb = max(2, int(600e6 // byts))
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() for _ in range(nb)]
    out = torch.zeros(m, n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m, a.n, a.k, a.wdt, a.out, a.ldo = x.data_ptr(), k, m, n, k, L.VV_BF16, out.data_ptr(), n
    a.res, a.ldres = res.data_ptr(), n
    s = torch.cuda.current_stream().cuda_stream
    def run(i):
        a.w = ws[i % nb].data_ptr()
        L.check(lib.vv_linear(C.byref(a), s), "lin")
    for i in range(10): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, byts / us / 1e3
for ku in (5, 3, 2):
    for cap in (192, 256, 384, 512, 768):
        lib.vv_tune(b"gemv_long_ku", ku); lib.vv_tune(b"gemv_long_cap", cap)
        us, gbs = bench(2, 1536, 8960)
        us2, gbs2 = bench(1, 2048, 8192)
        print(f"ku<={ku} cap={cap:4d}: n=1536 k=8960 {us:6.2f} us {gbs:7.1f} GB/s | m=1 n=2048 k=8192 {us2:6.2f} us {gbs2:7.1f} GB/s", flush=True)
