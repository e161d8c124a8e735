"""The 5..8-row matrix-core GEMV (vv_gemv_rows.hip) on the frame's big shapes: numerics against torch (fp64 reference on the bf16 weights)
and per-kernel time inside a dependent hipGraph chain, next to the 2-row VALU kernel and to the two-pass fallback."""
import sys, ctypes as C, torch
sys.argv = ['x'] + sys.argv[1:]
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tools")
import mb_chain_lin as M
L, lib = M.L, M.lib
L.check(lib.vv_init(), "init")
L.check(lib.vv_tune(b"gemv_rows_scratch", 1), "scratch")


def check(m, n, k, dual, pro, mod, epi, bias=False, frag=False):
    torch.manual_seed(m * 1000 + n + k)
    x = torch.randn(m, k, device="cuda") * 1.5
    w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    w2 = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() if dual else None
    nw = torch.rand(k, device="cuda") + 0.5
    sh = torch.randn(m, k, device="cuda") * 0.2; sc = torch.randn(m, k, device="cuda") * 0.2
    gate = torch.randn(m, n, device="cuda"); res = torch.randn(m, n, device="cuda"); b = torch.randn(n, device="cuda")
    out = torch.zeros(m, n, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.m = x.data_ptr(), k, m
    a.n, a.k, a.wdt = n, k, L.VV_BF16
    wf = M.frag_major(w) if frag else w
    a.w = wf.data_ptr()
    a.flags = L.LIN_W_FRAG if frag else 0
    a.out, a.ldo = out.data_ptr(), n
    a.pro, a.eps = pro, 1e-5
    if pro == 1: a.norm_w = nw.data_ptr()
    if mod: a.mod_shift, a.mod_scale, a.ld_mod = sh.data_ptr(), sc.data_ptr(), k
    w2f = (M.frag_major(w2) if frag else w2) if dual else None
    if dual: a.w2, a.act = w2f.data_ptr(), 2
    if epi: a.gate, a.gate_ld, a.res, a.ldres = gate.data_ptr(), n, res.data_ptr(), n
    if bias: a.bias = b.data_ptr()
    L.check(lib.vv_linear(C.byref(a), torch.cuda.current_stream().cuda_stream), "lin")
    torch.cuda.synchronize()
    xd = x.double()
    if pro == 1:
        xd = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-5) * nw.double()
        if mod: xd = xd * (1 + sc.double()) + sh.double()
    y = xd @ w.double().T
    if bias: y = y + b.double()
    if dual: y = torch.nn.functional.silu(y) * (xd @ w2.double().T)
    if epi: y = y * gate.double() + res.double()
    err = ((out.double() - y).norm() / y.norm()).item()
    print(f"  check m={m} n={n} k={k} dual={int(dual)} pro={pro} mod={int(mod)} epi={int(epi)} bias={int(bias)} frag={int(frag)}: rel err {err:.2e}", flush=True)
    return err


SHAPES = (("head gate/up", (4608, 1536, True, 4), dict(mod=True, flags=L.LIN_W_REUSED)), ("head down", (1536, 4608, False, 4), dict(pro=0, epi=True, flags=L.LIN_W_REUSED)),
          ("llm qkv", (2048, 1536, False, 64), {}), ("llm o", (1536, 1536, False, 64), dict(pro=0, epi=True)),
          ("llm gate/up", (8960, 1536, True, 12), {}), ("llm down", (1536, 8960, False, 24), dict(pro=0, epi=True)))
if __name__ == "__main__":
    worst = 0.0
    for m, fr in ((8, True), (5, False), (7, True)):
        worst = max(worst, check(m, 4608, 1536, True, 1, True, False, frag=fr), check(m, 1536, 4608, False, 0, False, True, frag=fr),
                    check(m, 2048, 1536, False, 1, False, False, bias=True, frag=fr), check(m, 1536, 1536, False, 0, False, True, frag=fr),
                    check(m, 8960, 1536, True, 1, False, False, frag=fr), check(m, 1536, 8960, False, 0, False, True, frag=fr),
                    check(m, 4608, 3584, False, 1, False, False, bias=True, frag=fr), check(m, 18944, 3584, True, 1, False, False, frag=fr),
                    check(m, 3584, 18944, False, 0, False, True, frag=fr), check(m, 1000, 1536, False, 1, False, False))
    print("worst rel err", worst)
    if "check" in sys.argv:
        sys.exit(0)
    for name, a, kw in SHAPES:
        print(name)
        M.chain(2, *a, **kw)
        M.chain(8, *a, frag=True, **kw)
        M.chain(8, *a, **kw)
        M.chain(8, *a, tune=(("gemv_rows", 0),), **kw)
        lib.vv_tune(b"gemv_rows", 1)
    print("== timing experiments (wrong results): dbg 2 no ticket merge, 4 no activation loads")
    for name, a, kw in SHAPES:
        print(name)
        for d in (2, 4, 6):
            M.chain(8, *a, frag=True, tune=(("gemv_rows_dbg", d),), **kw)
