"""Per-kernel time of vv_linear shapes inside a hipGraph chain of DEPENDENT launches (the way a frame runs them), next to the
synthetic floor of tools/mb_chain.cpp.  x of launch i is the output of launch i-1 (first k columns, ld = n), weights cycle
through `copies` sets (4 = cache-resident like the diffusion head across solver steps, many = streamed from HBM like the LLM)."""
import sys, ctypes as C, torch, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vibevoice_rocm_amd import _lib as L
lib = L.load()
st = torch.cuda.Stream()


def frag_major(w):
    """[N, K] bf16 -> the fragment-major copy [N / 16][K / 32][4][16][8] of vv_gemv_rows.hip (VV_LIN_W_FRAG)"""
    n, k = w.shape
    return w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def chain(m, n, k, dual, copies, pro=1, mod=False, epi=False, flags=0, N=120, tune=(), frag=False):
    for key, val in tune:
        lib.vv_tune(key.encode(), val)
    with torch.cuda.stream(st):
        ld = max(n, k)
        bufs = [torch.randn(m, ld, device="cuda") * 0.5 for _ in range(2)]
        ws = [((torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(), (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() if dual else None)
              for _ in range(copies)]
        if frag:
            ws = [(frag_major(a_), frag_major(b_) if b_ is not None else None) for a_, b_ in ws]
            flags |= L.LIN_W_FRAG
        nw = torch.ones(k, device="cuda"); sh = torch.zeros(m, k, device="cuda"); sc = torch.zeros(m, k, device="cuda")
        gate = torch.full((m, n), 0.5, device="cuda"); res = torch.zeros(m, ld, device="cuda")
        args = []
        for i in range(N):
            a = L.LinArgs()
            a.x, a.ldx, a.m = bufs[i & 1].data_ptr(), ld, m
            a.n, a.k, a.wdt = n, k, L.VV_BF16
            a.out, a.ldo = bufs[(i + 1) & 1].data_ptr(), ld
            a.pro, a.norm_w, a.eps = pro, (nw.data_ptr() if pro == 1 else 0), 1e-5
            a.flags = flags
            if mod:
                a.mod_shift, a.mod_scale, a.ld_mod = sh.data_ptr(), sc.data_ptr(), k
            a.w = ws[i % copies][0].data_ptr()
            if dual:
                a.w2, a.act = ws[i % copies][1].data_ptr(), 2
            if epi:
                a.gate, a.gate_ld, a.res, a.ldres = gate.data_ptr(), n, res.data_ptr(), ld
            args.append(a)
        st.synchronize()
        L.check(lib.vv_graph_begin(st.cuda_stream), "b")
        for a in args:
            L.check(lib.vv_linear(C.byref(a), st.cuda_stream), "lin")
        ge = C.c_void_p(); L.check(lib.vv_graph_end(st.cuda_stream, C.byref(ge)), "e")
        for _ in range(3):
            lib.vv_graph_launch(ge, st.cuda_stream)
        st.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(5):
                lib.vv_graph_launch(ge, st.cuda_stream)
            st.synchronize()
            best = min(best, (time.perf_counter() - t0) / 5 / N * 1e6)
        lib.vv_graph_destroy(ge)
    byts = n * k * 2 * (2 if dual else 1)
    floor = 3.2 + byts / 7.9e6
    print(f"m={m} n={n:5d} k={k:5d} dual={int(dual)} copies={copies:2d} pro={pro} mod={int(mod)} epi={int(epi)} flags={flags} frag={int(frag)} tune={tune}: {best:6.2f} us/kernel  "
          f"({byts / 1e6:5.1f} MB, {byts / best / 1e3:6.0f} GB/s; synthetic floor {floor:5.2f})", flush=True)
    for key, _ in tune:
        lib.vv_tune(key.encode(), 0 if key == "gemv_blocks" else {"gemv_dual_rw": 1, "gemv_small_rw": 2}.get(key, 0))
    return best


def ab(key, vals, *a, rounds=3, **kw):
    """interleaved A/B of one tuning key on one shape, in this process, on this device (min over rounds)"""
    import io, contextlib
    best = {v: 1e9 for v in vals}
    for _ in range(rounds):
        for v in vals:
            with contextlib.redirect_stdout(io.StringIO()):
                best[v] = min(best[v], chain(*a, tune=((key, v),), **kw))
    lib.vv_tune(key.encode(), 1 if key == "gemv_opt" else 0)
    print(f"A/B {key}: " + "  ".join(f"{v}: {best[v]:6.2f} us" for v in vals) + f"   shape {a} {kw}", flush=True)


SHAPES = [
    ("head gate/up", (2, 4608, 1536, True, 4), dict(mod=True, flags=L.LIN_W_REUSED)),
    ("head down", (2, 1536, 4608, False, 4), dict(pro=0, epi=True, flags=L.LIN_W_REUSED)),
    ("head final", (2, 64, 1536, False, 4), dict(mod=True, flags=L.LIN_W_REUSED)),
    ("llm gate/up", (2, 8960, 1536, True, 12), dict()),
    ("llm down", (2, 1536, 8960, False, 24), dict(pro=0, epi=True)),
    ("llm qkv", (2, 2048, 1536, False, 64), dict()),
    ("llm o", (2, 1536, 1536, False, 64), dict(pro=0, epi=True)),
    ("conv lin1", (1, 8192, 2048, False, 16), dict()),
    ("conv lin2", (1, 2048, 8192, False, 16), dict(pro=0, epi=True)),
]

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "ab":
        key, vals = sys.argv[2], [int(v) for v in sys.argv[3].split(",")]
        for name, a, kw in SHAPES:
            print(name, end=": ")
            ab(key, vals, *a, **kw)
        sys.exit(0)
    REUSED = L.LIN_W_REUSED
    print("== head (cache-resident weights)")
    for blocks in (0, 256, 384, 576, 768, 1024):
        chain(2, 4608, 1536, True, 4, mod=True, flags=REUSED, tune=(("gemv_blocks", blocks),))
    chain(2, 4608, 1536, True, 4, mod=True, flags=REUSED, tune=(("gemv_dual_rw", 2),))
    for blocks in (0, 384, 512, 1024):
        chain(2, 1536, 4608, False, 4, pro=0, epi=True, flags=REUSED, tune=(("gemv_blocks", blocks),))
    chain(2, 64, 1536, False, 4, mod=True, flags=REUSED)
    print("== LLM (HBM)")
    for blocks in (0, 256, 560, 768, 1024):
        chain(2, 8960, 1536, True, 12, tune=(("gemv_blocks", blocks),))
    chain(2, 8960, 1536, True, 12, tune=(("gemv_dual_rw", 2),))
    for blocks in (0, 256, 384, 768):
        chain(2, 1536, 8960, False, 24, pro=0, epi=False, tune=(("gemv_blocks", blocks),))
    for blocks in (0, 128, 256, 512):
        chain(2, 2048, 1536, False, 64, tune=(("gemv_blocks", blocks),))
    for blocks in (0, 96, 192, 384):
        chain(2, 1536, 1536, False, 64, pro=0, tune=(("gemv_blocks", blocks),))
    print("== conv stage 0 (HBM)")
    chain(1, 8192, 2048, False, 16)
    chain(1, 2048, 8192, False, 16, pro=0)
