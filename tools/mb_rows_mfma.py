"""Row-count scaling of the two GEMV kernels (VALU vs matrix-core) on the frame's big shapes: what batching dialogues into rows would cost."""
import sys
sys.argv = ['x']
sys.path.insert(0, '/root/repo/tools')
import mb_chain_lin as M
L = M.L
for name, a, kw in (("head gate/up", (4608, 1536, True, 4), dict(mod=True, flags=L.LIN_W_REUSED)), ("head down", (1536, 4608, False, 4), dict(pro=0, epi=True, flags=L.LIN_W_REUSED)),
                    ("llm gate/up", (8960, 1536, True, 12), {}), ("llm down", (1536, 8960, False, 24), dict(pro=0, epi=True))):
    for m in (2, 4):
        for mf in (0, 1):
            print(f"{name} m={m} gemv_mfma={mf}: ", end="")
            M.chain(m, *a, tune=(("gemv_mfma", mf),), **kw)
M.lib.vv_tune(b"gemv_mfma", 0)
