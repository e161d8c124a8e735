import sys
sys.argv=['x']
sys.path.insert(0,'/root/repo/tools')
import mb_chain_lin as M
L=M.L
for m in (1,2,4):
    M.chain(m, 4608, 1536, True, 4, mod=True, flags=L.LIN_W_REUSED)
    M.chain(m, 1536, 4608, False, 4, pro=0, epi=True, flags=L.LIN_W_REUSED)
    M.chain(m, 8960, 1536, True, 12)
    M.chain(m, 1536, 8960, False, 24, pro=0, epi=True)
