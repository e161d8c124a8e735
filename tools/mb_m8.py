import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, pro=0, act=0, bias=False, iters=200, blocks=0):
    x = torch.randn(m, k, device="cuda")
    byts = n * k * 2
    nb = max(2, int(400e6 // byts))
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() for _ in range(nb)]
    out = torch.zeros(m, n, device="cuda"); nw = torch.ones(k, device="cuda"); b = torch.zeros(n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m = x.data_ptr(), k, m
    a.n, a.k, a.wdt = n, k, L.VV_BF16
    a.out, a.ldo = out.data_ptr(), n
    a.pro, a.act, a.eps = pro, act, 1e-5
    if pro == 1: a.norm_w = nw.data_ptr()
    if bias: a.bias = b.data_ptr()
    lib.vv_tune(b"gemv_blocks", blocks)
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
    print(f"m={m} n={n} k={k} pro={pro} act={act} bias={bias} blocks={blocks}: {us:7.2f} us  {byts/us/1e3:7.1f} GB/s", flush=True)
for m in (1, 4, 8):
    bench(m, 4096, 1024)
    bench(m, 4096, 1024, pro=1, act=1, bias=True)
    bench(m, 1024, 4096)
    bench(m, 1024, 4096, bias=True)
for blocks in (128, 256, 512, 1024):
    bench(8, 4096, 1024, pro=1, act=1, bias=True, blocks=blocks)
    bench(8, 1024, 4096, bias=True, blocks=blocks)
