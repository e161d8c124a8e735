"""Where a streaming frame of the acoustic decoder / semantic encoder spends its time: vv_linear calls by shape (vv_prof, eager,
HIP events) and, when run under rocprofv3 --kernel-trace --stats, every kernel by name."""
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
which = sys.argv[1] if len(sys.argv) > 1 else "both"
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
eng = m.engine; lib = eng.lib; w = eng.w
def dec(): eng._ck(lib.vv_decoder_forward(C.byref(w.dec), eng.latent.data_ptr(), 1, 5.0, -0.05, eng.wav.data_ptr(), eng._dec_ws.data_ptr(), eng.sp), "d")
def sem(): eng._ck(lib.vv_encoder_forward(C.byref(w.sem), eng.wav.data_ptr(), cfg.hop, eng.sem.data_ptr(), eng._sem_ws.data_ptr(), eng.sp), "s")
for name, fn in (("decoder", dec), ("encoder", sem)):
    if which not in ("both", name): continue
    with torch.cuda.stream(eng.stream):
        for _ in range(3): fn()
        eng.stream.synchronize()
        reps = 20
        L.check(lib.vv_prof_begin(20000), "pb")
        for _ in range(reps): fn()
        out = (L.ProfEntry * 256)(); n = C.c_int()
        L.check(lib.vv_prof_end(out, 256, C.byref(n)), "pe")
    rows = sorted(((e.total_ms / reps * 1e3, e.count // reps, e.m, e.n, e.k, e.dual) for e in out[: n.value]), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{name}: vv_linear eager total {tot:.0f} us / frame over {sum(r[1] for r in rows)} calls")
    for us, cnt, mm, nn, kk, dual in rows:
        print(f"   m={mm:5d} n={nn:5d} k={kk:5d} dual={dual} x{cnt:2d}: {us:7.1f} us total, {us/cnt:6.1f} us each")
