"""The hoisted adaLN modulation GEMM of the diffusion head (40 rows x [4608, 1536] bf16): fp32 x staged through LDS vs bf16 x
streamed from global, 32- or 64-row tiles."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, xb, mt, iters=100):
    x = torch.randn(m, k, device="cuda")
    xb16 = x.bfloat16().contiguous()
    nb = 8
    ws = [(torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() for _ in range(nb)]
    out = torch.zeros(m, n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m = (xb16 if xb else x).data_ptr(), k, m
    a.n, a.k, a.wdt = n, k, L.VV_BF16
    a.out, a.ldo = out.data_ptr(), n
    a.flags = L.LIN_X_BF16 if xb else 0
    lib.vv_tune(b"mfma_mt", mt)
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
    ref = x @ ws[(iters - 1) % nb].float().T
    err = ((out - ref).norm() / ref.norm()).item()
    print(f"m={m} n={n} k={k} xb={xb} mt={mt}: {us:7.2f} us  {n*k*2/us/1e3:7.1f} GB/s  rel err {err:.2e}", flush=True)
for (m, n, k) in ((40, 4608, 1536), (40, 3072, 1536), (20, 4608, 1536)):
    for xb in (0, 1):
        for mt in (1, 2):
            bench(m, n, k, xb, mt)
lib.vv_tune(b"mfma_mt", 0)
