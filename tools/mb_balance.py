"""Row-group balance of the streaming GEMV: persistent grid sizes that give every wave the same number of row groups."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, dual, blocks, iters=300):
    x = torch.randn(m, k, device="cuda")
    byts = n * k * 2 * (2 if dual else 1)
    nb = max(2, int(600e6 // byts))
    ws = [((torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(), (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() if dual else None) for _ in range(nb)]
    out = torch.zeros(m, n, device="cuda"); nw = torch.ones(k, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m, a.n, a.k, a.wdt, a.out, a.ldo = x.data_ptr(), k, m, n, k, L.VV_BF16, out.data_ptr(), n
    a.pro, a.norm_w, a.eps = 1, nw.data_ptr(), 1e-5
    if dual: a.act = 2
    lib.vv_tune(b"gemv_blocks", blocks)
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
    print(f"m={m} n={n} k={k} dual={dual} blocks={blocks:4d}: {us:7.2f} us  {byts/us/1e3:7.1f} GB/s", flush=True)
for blocks in (0, 384, 512, 576, 640, 768, 1152):
    bench(2, 4608, 1536, True, blocks)
for blocks in (0, 448, 512, 560, 640, 747, 1120):
    bench(2, 8960, 1536, True, blocks)
for blocks in (0, 128, 192, 256, 384):
    bench(2, 2048, 1536, False, blocks)
    bench(2, 1536, 1536, False, blocks)
lib.vv_tune(b"gemv_blocks", 0)
