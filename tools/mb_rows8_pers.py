"""Persistent-grid sweep of the 8-row SwiGLU GEMV (vv_gemv_rows.hip, tuning hook gemv_rows_pers) on the head's and the LLM's shape."""
import sys
sys.argv = ['x']
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tools")
import mb_chain_lin as M
L, lib = M.L, M.lib
L.check(lib.vv_init(), "init")
L.check(lib.vv_tune(b"gemv_rows_scratch", 1), "scratch")
for name, a, kw, grids in (("head gate/up", (4608, 1536, True, 4), dict(mod=True, flags=L.LIN_W_REUSED), (72, 96, 144)),
                           ("llm gate/up", (8960, 1536, True, 12), {}, (80, 112, 140, 187))):
    print(name)
    for gsz in grids:
        M.chain(8, *a, frag=True, tune=(("gemv_rows_pers", gsz),), **kw)
lib.vv_tune(b"gemv_rows_pers", 256)
