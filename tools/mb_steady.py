"""Steady-state vs per-call streaming rate of the GEMV kernels: the same kernel on a 16x taller matrix shows what the memory
pipe sustains once launch ramp and tail are amortised (the gap is what a persistent chained kernel could recover)."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()

def bench(m, n, k, dual, blocks, iters=200, copies=None):
    x = torch.randn(m, k, device="cuda")
    byts = n * k * 2 * (2 if dual else 1)
    nb = copies or max(2, int(800e6 // byts))
    ws = [((torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(), (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() if dual else None) for _ in range(nb)]
    out = torch.zeros(m, n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m = x.data_ptr(), k, m
    a.n, a.k, a.wdt = n, k, L.VV_BF16
    a.out, a.ldo = out.data_ptr(), n
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
    print(f"m={m} n={n:7d} k={k} dual={dual} copies={nb} blocks={blocks:5d}: {us:8.2f} us  {byts/us/1e3:7.1f} GB/s", flush=True)

for rw in (1, 2):
    lib.vv_tune(b"gemv_dual_rw", rw)
    print("dual rw", rw)
    for blocks in (256, 384, 512):
        bench(2, 4608, 1536, True, blocks)
        bench(2, 4608 * 16, 1536, True, blocks, iters=40)
        bench(2, 8960, 1536, True, blocks)
        bench(2, 8960 * 8, 1536, True, blocks, iters=40)
for blocks in (256, 512, 768):
    bench(2, 1536, 4608, False, blocks)
    bench(2, 1536 * 16, 4608, False, blocks, iters=40)
    bench(2, 1536, 8960, False, blocks)
    bench(2, 1536 * 8, 8960, False, blocks, iters=40)
    bench(2, 2048, 1536, False, blocks)
    bench(2, 2048 * 32, 1536, False, blocks, iters=40)
# plain device copy for scale
src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
for _ in range(3): dst.copy_(src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): dst.copy_(src)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
print(f"copy 1 GiB: {us:.1f} us  read+write {2*(1<<30)/us/1e3:.0f} GB/s", flush=True)
t = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
for _ in range(3): t.sum()
torch.cuda.synchronize()
e0.record()
for _ in range(10): t.sum()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
print(f"sum 1 GiB: {us:.1f} us  read {(1<<30)/us/1e3:.0f} GB/s", flush=True)
