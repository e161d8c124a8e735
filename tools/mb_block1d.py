"""In-graph time of one fused Block1D launch (vv_block1d) at the per-frame shapes of the narrow conv stages."""
import sys, time, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
L.check(lib.vv_init(), "init")
st = torch.cuda.Stream(); sp = st.cuda_stream
for C_, T in ((128, 800), (64, 1600), (32, 3200), (128, 3200), (32, 800)):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    p = dict(gamma=r(C_, sc=0.5), ffn_gamma=r(C_, sc=0.5), norm_w=1 + r(C_, sc=0.1), ffn_norm_w=1 + r(C_, sc=0.1), dw_w=r(C_, 7, sc=0.3), dw_b=r(C_, sc=0.1),
             w1=(r(4 * C_, C_) / C_ ** 0.5).bfloat16(), b1=r(4 * C_, sc=0.1), w2=(r(C_, 4 * C_) / (4 * C_) ** 0.5).bfloat16(), b2=r(C_, sc=0.1))
    hist = torch.zeros(6, C_, device="cuda")
    b = L.Block()
    for k, v in p.items(): setattr(b, k, v.data_ptr())
    b.hist = hist.data_ptr()
    xa, xb = r(T, C_), torch.zeros(T, C_, device="cuda")
    n = 20
    with torch.cuda.stream(st):
        L.check(lib.vv_graph_begin(sp), "b")
        for i in range(n):
            src, dst = (xa, xb) if i % 2 == 0 else (xb, xa)
            L.check(lib.vv_block1d(C.byref(b), L.VV_BF16, src.data_ptr(), dst.data_ptr(), T, C_, 1e-5, sp), "blk")
        ge = C.c_void_p(); L.check(lib.vv_graph_end(sp, C.byref(ge)), "e")
        for _ in range(3): lib.vv_graph_launch(ge, sp)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): lib.vv_graph_launch(ge, sp)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 20 / n * 1e6
    print(f"C={C_:4d} T={T:5d}: {dt:6.2f} us per block (in graph, {(T + 31) // 32} workgroups)", flush=True)
