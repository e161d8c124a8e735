"""Phase timing of ffn_in_kernel (vv_convffn.hip) on the middle-stage shapes of a streaming frame.  Builds a debug copy of the library with
-DVV_CF_TIMING into tools/bin (the product library carries no stamps) and runs vv_block_mid on it.

  python tools/convffn_phase.py build     (CPU container or GPU box)
  python tools/convffn_phase.py run       (GPU box)"""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "bin", "libvv_hip_cft.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = sorted(glob.glob(os.path.join(ROOT, "vibevoice_rocm_amd", "csrc", "*.hip")))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DVV_CF_TIMING", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "vibevoice_rocm_amd", "csrc")] + src + ["-o", SO]
    subprocess.check_call(cmd)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from vibevoice_rocm_amd import _lib as L
    lib = C.CDLL(SO)
    assert lib.vv_init() == 0
    lib.vv_block_mid_ws_bytes.restype = C.c_size_t
    for C_, T, rows in ((512, 40, 32), (512, 40, 16), (512, 40, 8), (256, 200, 32), (256, 200, 16), (256, 200, 8), (1024, 8, 8)):
        if C_ != 1024:
            lib.vv_tune(f"convffn_rows{C_}".encode(), rows)
        g = torch.Generator().manual_seed(1)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
        p = dict(gamma=r(C_, sc=0.5), ffn_gamma=r(C_, sc=0.5), norm_w=1 + r(C_, sc=0.1), ffn_norm_w=1 + r(C_, sc=0.1), dw_w=r(C_, 7, sc=0.3), dw_b=r(C_, sc=0.1),
                 w1=(r(4 * C_, C_) / C_ ** 0.5).bfloat16(), b1=r(4 * C_, sc=0.1), w2=(r(C_, 4 * C_) / (4 * C_) ** 0.5).bfloat16(), b2=r(C_, sc=0.1))
        hist = torch.zeros(6, C_, device="cuda")
        b = L.Block()
        for k, v in p.items():
            setattr(b, k, v.data_ptr())
        b.hist = hist.data_ptr()
        ws = torch.empty(lib.vv_block_mid_ws_bytes(T, C_), dtype=torch.uint8, device="cuda")
        x, o = r(T, C_), torch.empty(T, C_, device="cuda")
        t = (C.c_ulonglong * 8)()
        call = lambda: lib.vv_block_mid(C.byref(b), L.VV_BF16, C.c_void_p(x.data_ptr()), C.c_void_p(o.data_ptr()), C.c_void_p(ws.data_ptr()), T, C_, C.c_float(1e-5), None)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        lib.vv_convffn_debug_times(t, 1)
        n = 200
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        lib.vv_convffn_debug_times(t, 1)
        names = ["issue", "loads+stat1", "xn->LDS", "conv+stat2", "xh+y", "mfma", "epilogue"]
        ns = [t[i] * 10.0 / n for i in range(7)]          # wall_clock64: 100 MHz
        print(f"C={C_} T={T} tile {rows}: " + "  ".join(f"{nm} {v:6.0f} ns" for nm, v in zip(names, ns)) + f"   total {sum(ns) / 1e3:.2f} us", flush=True)


def run_narrow():
    import torch
    sys.path.insert(0, ROOT)
    from vibevoice_rocm_amd import _lib as L
    lib = C.CDLL(SO)
    assert lib.vv_init() == 0
    for C_, T in ((128, 800), (64, 1600), (32, 3200)):
        g = torch.Generator().manual_seed(1)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
        p = dict(gamma=r(C_, sc=0.5), ffn_gamma=r(C_, sc=0.5), norm_w=1 + r(C_, sc=0.1), ffn_norm_w=1 + r(C_, sc=0.1), dw_w=r(C_, 7, sc=0.3), dw_b=r(C_, sc=0.1),
                 w1=(r(4 * C_, C_) / C_ ** 0.5).bfloat16(), b1=r(4 * C_, sc=0.1), w2=(r(C_, 4 * C_) / (4 * C_) ** 0.5).bfloat16(), b2=r(C_, sc=0.1))
        hist = torch.zeros(6, C_, device="cuda")
        b = L.Block()
        for k, v in p.items():
            setattr(b, k, v.data_ptr())
        b.hist = hist.data_ptr()
        x, o = r(T, C_), torch.empty(T, C_, device="cuda")
        t = (C.c_ulonglong * 8)()
        call = lambda: lib.vv_block1d(C.byref(b), L.VV_BF16, C.c_void_p(x.data_ptr()), C.c_void_p(o.data_ptr()), T, C_, C.c_float(1e-5), None)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        lib.vv_block1d_debug_times(t, 1)
        n = 200
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        lib.vv_block1d_debug_times(t, 1)
        names = ["loads", "stat+xn", "mixer+norm2", "gemm1+gelu", "gemm2+epi"]
        ns = [t[i] * 10.0 / n for i in range(5)]
        print(f"block1d C={C_} T={T}: " + "  ".join(f"{nm} {v:6.0f} ns" for nm, v in zip(names, ns)) + f"   total {sum(ns) / 1e3:.2f} us", flush=True)


def run_attn_split():
    """phase stamps of workgroup (0, 0, 0) of the grouped decode attention on its split-key path (through an Engine's batch-2 decode step)"""
    import torch
    sys.path.insert(0, ROOT)
    from vibevoice_rocm_amd import _lib as L
    L.LIB_PATH = SO                      # the debug build (stamps compiled in) instead of the product library
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.engine import Engine
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    cfg = VVConfig.preset("1.5b")
    sd = synth_state_dict_torch(cfg, 1, device="cuda:0", dtype=torch.bfloat16)
    eng = Engine(cfg, sd, device="cuda:0", dtype=torch.bfloat16, use_graphs=False)
    lib = eng.lib
    V = cfg.vocab
    for S, smax in ((440, 4096), (12000, 12288), (32000, 32768), (64000, 65536)):
        eng.begin_sequence(smax, [V - 4, V - 3, V - 2, V - 1])
        lib.vv_tune(b"attn_gqa", 2)
        t = (C.c_ulonglong * 8)()
        def step():
            with torch.cuda.stream(eng.stream):
                eng.lens.copy_(torch.tensor([S, S // 3], dtype=torch.int32))
                eng.llm_forward(eng.x2, eng.lens, None, eng.hidden2)
        for _ in range(3):
            step()
        eng.stream.synchronize()
        dbg = C.CDLL(SO).vv_attn_debug_times
        dbg(t, 1)
        n = 10
        for _ in range(n):
            step()
        eng.stream.synchronize()
        dbg(t, 1)
        names = ["requests+pos", "rope+split+new token", "key tiles", "LDS+barrier", "merge+partials", "fence+ticket", "final fold"]
        ns = [t[i] * 10.0 / (n * cfg.layers) for i in range(7)]
        print(f"grouped split S={S} s_max={smax}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, ns)) + f"   total {sum(ns) / 1e3:.2f} us (workgroup 0,0,0)", flush=True)


def run_gemv():
    import torch
    sys.path.insert(0, ROOT)
    from vibevoice_rocm_amd import _lib as L
    lib = C.CDLL(SO)
    assert lib.vv_init() == 0
    for m, n, k, dual, mod, epi, copies in ((2, 4608, 1536, True, True, False, 4), (2, 1536, 4608, False, False, True, 4), (2, 8960, 1536, True, False, False, 12),
                                            (2, 1536, 8960, False, False, True, 24)):
        ws = [((torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16(), (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()) for _ in range(copies)]
        x, out = torch.randn(m, k, device="cuda"), torch.empty(m, n, device="cuda")
        nw, sh, sc = torch.ones(k, device="cuda"), torch.zeros(m, k, device="cuda"), torch.zeros(m, k, device="cuda")
        gate, res = torch.ones(m, n, device="cuda"), torch.zeros(m, n, device="cuda")
        args = []
        for i in range(copies):
            a = L.LinArgs()
            a.x, a.ldx, a.m, a.n, a.k, a.wdt, a.out, a.ldo = x.data_ptr(), k, m, n, k, L.VV_BF16, out.data_ptr(), n
            a.w = ws[i][0].data_ptr()
            if dual:
                a.w2, a.act, a.pro, a.norm_w, a.eps = ws[i][1].data_ptr(), 2, 1, nw.data_ptr(), 1e-5
                if mod:
                    a.mod_shift, a.mod_scale, a.ld_mod = sh.data_ptr(), sc.data_ptr(), k
            if epi:
                a.gate, a.gate_ld, a.res, a.ldres = gate.data_ptr(), n, res.data_ptr(), n
            if copies == 4:
                a.flags = L.LIN_W_REUSED
            args.append(a)
        t = (C.c_ulonglong * 8)()
        for i in range(8):
            assert lib.vv_linear(C.byref(args[i % copies]), None) == 0
        torch.cuda.synchronize()
        lib.vv_gemv_mfma_debug_times(t, 1)
        nrep = 240
        for i in range(nrep):
            lib.vv_linear(C.byref(args[i % copies]), None)
        torch.cuda.synchronize()
        lib.vv_gemv_mfma_debug_times(t, 1)
        names = ["issue", "x+stats", "barrier1", "norm+split", "barrier2", "frags", "mfma(+w wait)", "reduce+epi"]
        ns = [t[i] * 10.0 / nrep for i in range(8)]
        print(f"gemv m={m} n={n} k={k} dual={int(dual)}: " + "  ".join(f"{nm} {v:5.0f}" for nm, v in zip(names, ns)) + f"   total {sum(ns) / 1e3:.2f} us (block 0)", flush=True)


def run_attn(gqa=0):
    import torch
    sys.path.insert(0, ROOT)
    from vibevoice_rocm_amd import _lib as L
    lib = C.CDLL(SO)
    assert lib.vv_init() == 0
    lib.vv_tune(b"attn_gqa", gqa)
    heads, kvh, d, layers, R = 12, 2, 128, 4, 2
    for S in (64, 450, 900):
        s_max = 1024
        k = torch.randn(layers, R, kvh, s_max, d, device="cuda").bfloat16()
        v = torch.randn(layers, R, kvh, s_max, d, device="cuda").bfloat16()
        vt = torch.randn(layers, R, kvh, s_max // 32, d, 32, device="cuda").bfloat16()
        kv = L.KV()
        kv.k, kv.v, kv.layers, kv.rows, kv.kv_heads, kv.s_max, kv.head_dim, kv.kvdt = k.data_ptr(), v.data_ptr(), layers, R, kvh, s_max, d, L.VV_BF16
        kv.vt = vt.data_ptr()
        qkv = torch.randn(R, (heads + 2 * kvh) * d, device="cuda")
        lens = torch.tensor([S, S // 4], dtype=torch.int32, device="cuda")
        inv = (1.0 / (1e6 ** (torch.arange(0, d, 2).float() / d))).cuda()
        rope = torch.empty(R * d, device="cuda")
        out = torch.empty(R, heads * d, device="cuda")
        assert lib.vv_rope_table(C.c_void_p(lens.data_ptr()), C.c_void_p(inv.data_ptr()), R, d, C.c_void_p(rope.data_ptr()), None) == 0
        t = (C.c_ulonglong * 8)()
        call = lambda l: lib.vv_attn_decode(C.c_void_p(qkv.data_ptr()), C.c_int64(qkv.shape[1]), R, heads, C.byref(kv), l, C.c_void_p(rope.data_ptr()),
                                            C.c_void_p(lens.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int64(heads * d), None)
        for i in range(8):
            assert call(i % layers) == 0, lib.vv_last_error()
        torch.cuda.synchronize()
        lib.vv_attn_debug_times(t, 1)
        n = 200
        for i in range(n):
            call(i % layers)
        torch.cuda.synchronize()
        lib.vv_attn_debug_times(t, 1)
        names = ["issue+pos", "q/k/rope", "key batches", "new token", "merge+store"]
        ns = [t[i] * 10.0 / n for i in range(5)]
        print(f"attn S={S}: " + "  ".join(f"{nm} {v:5.0f}" for nm, v in zip(names, ns)) + f"   total {sum(ns) / 1e3:.2f} us (block 0)", flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["attn"]:
        run_attn(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    if sys.argv[1:] == ["attn_split"]:
        run_attn_split()
        sys.exit(0)
    if sys.argv[1:] == ["gemv"]:
        run_gemv()
        sys.exit(0)
    if sys.argv[1:] == ["narrow"]:
        run_narrow()
        sys.exit(0)
    (build if sys.argv[1:] == ["build"] else run)()
