import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
lib = L.load()
heads, kvh, d, layers, smax = 12, 2, 128, 28, 1024
kv = L.KV()
kt = torch.randn(layers, 2, kvh, smax, d, device="cuda").bfloat16()
vt = torch.randn(layers, 2, kvh, smax, d, device="cuda").bfloat16()
kv.k, kv.v, kv.kvdt, kv.layers, kv.rows, kv.kv_heads, kv.s_max, kv.head_dim = kt.data_ptr(), vt.data_ptr(), L.VV_BF16, layers, 2, kvh, smax, d
qkv = torch.randn(2, (heads+2*kvh)*d, device="cuda")
out = torch.zeros(2, heads*d, device="cuda")
inv = (1.0 / (1e6 ** (torch.arange(0, d, 2).float()/d))).cuda()
s = torch.cuda.current_stream().cuda_stream
table = torch.zeros(2, d//2, 2, device="cuda")
for S in (0, 1, 64, 440, 900):
    lens = torch.tensor([S, max(S//2,0)], dtype=torch.int32, device="cuda")
    L.check(lib.vv_rope_table(lens.data_ptr(), inv.data_ptr(), 2, d, table.data_ptr(), s), "table")
    def run(i):
        L.check(lib.vv_attn_decode(qkv.data_ptr(), qkv.shape[1], 2, heads, C.byref(kv), i % layers, table.data_ptr(), lens.data_ptr(), out.data_ptr(), heads*d, s), "attn")
    for i in range(30): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(280): run(i)
    e1.record(); torch.cuda.synchronize()
    print(f"S={S}: {e0.elapsed_time(e1)*1e3/280:.2f} us")
