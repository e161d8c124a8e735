import sys, ctypes as C, torch, time
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
def bench(m, n, k, dual, blocks, iters=300, pro=1, mod=True, flags=0, copies=None):
    x = torch.randn(m, k, device="cuda")
    w = (torch.randn(n, k, device="cuda")/k**0.5).bfloat16()
    w2 = (torch.randn(n, k, device="cuda")/k**0.5).bfloat16()
    # several weight copies to defeat the 256MB infinity cache
    nb = copies or max(1, int(600e6 // (w.numel()*2*(2 if dual else 1))))
    ws = [(w.clone(), w2.clone()) for _ in range(nb)]
    nw = torch.ones(k, device="cuda"); sh = torch.zeros(m, k, device="cuda"); sc = torch.zeros(m,k,device="cuda")
    out = torch.zeros(m, n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m = x.data_ptr(), k, m
    a.n, a.k, a.wdt = n, k, L.VV_BF16
    a.out, a.ldo = out.data_ptr(), n
    a.pro, a.norm_w, a.eps = pro, nw.data_ptr(), 1e-5
    a.flags = flags
    if mod: a.mod_shift, a.mod_scale, a.ld_mod = sh.data_ptr(), sc.data_ptr(), k
    if dual: a.act = 2
    lib.vv_tune(b"gemv_blocks", blocks)
    s = torch.cuda.current_stream().cuda_stream
    def run(i):
        a.w = ws[i % nb][0].data_ptr()
        if dual: a.w2 = ws[i % nb][1].data_ptr()
        L.check(lib.vv_linear(C.byref(a), s), "lin")
    for i in range(20): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*1e3/iters
    byts = n*k*2*(2 if dual else 1)
    print(f"m={m} n={n} k={k} dual={dual} pro={pro} mod={mod} flags={flags} copies={nb} blocks={blocks:5d}: {us:7.2f} us  {byts/us/1e3:7.1f} GB/s", flush=True)




for rw in (2, 1):
    lib.vv_tune(b"gemv_small_rw", rw)
    print("small rw", rw)
    for blocks in (0, 128, 256, 512):
        bench(2, 2048, 1536, False, blocks, mod=False)
    for blocks in (0, 192, 384):
        bench(2, 1536, 1536, False, blocks, pro=0, mod=False)
    for blocks in (0, 256, 512, 1024):
        bench(1, 8192, 2048, False, blocks, mod=False)
