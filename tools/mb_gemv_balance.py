"""Row-group balance of the whole-row GEMVs: (blocks, waves per block) so that every wave gets the same number of weight rows AND the grid is
a multiple of the CU count.  head gate/up: 4608 rows = 256 x 6 x 3 = 512 x 3 x 3; LLM gate/up: 8960 rows = 256 x 5 x 7 = 256 x 7 x 5."""
import sys
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from mb_chain_lin import chain, lib, L   # noqa: E402
R = L.LIN_W_REUSED
print("== head gate/up (2 x 4608 x 1536 dual, cache-resident)")
for blocks, waves in ((0, 0), (256, 6), (512, 3), (256, 8), (512, 4), (384, 4), (768, 3), (256, 4)):
    chain(2, 4608, 1536, True, 4, mod=True, flags=R, tune=(("gemv_blocks", blocks), ("gemv_waves", waves)))
print("== llm gate/up (2 x 8960 x 1536 dual, HBM)")
for blocks, waves in ((0, 0), (256, 5), (256, 7), (512, 5), (512, 7), (256, 8), (320, 4), (448, 5)):
    chain(2, 8960, 1536, True, 12, tune=(("gemv_blocks", blocks), ("gemv_waves", waves)))
print("== llm down (2 x 1536 x 8960, K split over 4 waves, HBM): persistent blocks")
for cap in (512, 384, 256, 768):
    chain(2, 1536, 8960, False, 24, pro=0, epi=True, tune=(("gemv_long_cap", cap),))
lib.vv_tune(b"gemv_long_cap", 512)
print("== llm qkv / o")
for blocks, waves in ((0, 0), (128, 8), (256, 4), (512, 4)):
    chain(2, 2048, 1536, False, 64, tune=(("gemv_blocks", blocks), ("gemv_waves", waves)))
for blocks, waves in ((0, 0), (96, 8), (128, 6), (256, 3)):
    chain(2, 1536, 1536, False, 64, pro=0, epi=True, tune=(("gemv_blocks", blocks), ("gemv_waves", waves)))
print("== conv stage 0 W2 (1 x 2048 x 8192)")
for cap in (512, 256, 1024):
    chain(1, 2048, 8192, False, 16, pro=0, epi=True, tune=(("gemv_long_cap", cap),))
lib.vv_tune(b"gemv_long_cap", 512)
