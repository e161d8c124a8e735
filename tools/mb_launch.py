import sys, ctypes as C, torch, time
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
st = torch.cuda.Stream()
x = torch.zeros(4096, device="cuda"); y = torch.zeros(4096, device="cuda")
N = 500
def seq(n_el):
    for i in range(N):
        L.check(lib.vv_affine(x.data_ptr(), 1.0, 0.0, y.data_ptr(), n_el, st.cuda_stream), "aff")
with torch.cuda.stream(st):
    for n_el in (64, 4096):
        seq(n_el); st.synchronize()
        t0 = time.perf_counter(); seq(n_el); st.synchronize(); t1 = time.perf_counter()
        print(f"eager n={n_el}: {(t1-t0)/N*1e6:.2f} us per kernel")
        L.check(lib.vv_graph_begin(st.cuda_stream), "b")
        seq(n_el)
        ge = C.c_void_p(); L.check(lib.vv_graph_end(st.cuda_stream, C.byref(ge)), "e")
        for _ in range(3): lib.vv_graph_launch(ge, st.cuda_stream)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): lib.vv_graph_launch(ge, st.cuda_stream)
        st.synchronize(); t1 = time.perf_counter()
        print(f"graph n={n_el}: {(t1-t0)/10/N*1e6:.2f} us per kernel")
