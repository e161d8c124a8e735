"""What the RMSNorm / adaLN prologue costs the decode GEMVs: the same launches with and without it (dependent graph chains)."""
import sys
sys.argv = ['x']
sys.path.insert(0, '/root/repo/tools')
import mb_chain_lin as M
L = M.L
for name, a, kw in (("head gate/up", (2, 4608, 1536, True, 4), dict(flags=L.LIN_W_REUSED)), ("llm gate/up", (2, 8960, 1536, True, 12), {}),
                    ("llm qkv", (2, 2048, 1536, False, 64), {})):
    print(name)
    M.chain(*a, pro=1, mod=(name.startswith("head")), **kw)
    M.chain(*a, pro=1, mod=False, **kw)
    M.chain(*a, pro=0, mod=False, **kw)
