"""decoder / semantic frame time (graph replay, unprofiled) under vv_tune settings:  python tools/mb_tune_components.py key=value[,key=value] ..."""
import sys, time, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
eng = m.engine; lib = eng.lib; w = eng.w


def timeit(fn, reps=100):
    with torch.cuda.stream(eng.stream):
        L.check(lib.vv_graph_begin(eng.sp), "b"); fn(); ge = C.c_void_p(); L.check(lib.vv_graph_end(eng.sp, C.byref(ge)), "e")
        for _ in range(3): lib.vv_graph_launch(ge, eng.sp)
        eng.stream.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                lib.vv_graph_launch(ge, eng.sp)
            eng.stream.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        lib.vv_graph_destroy(ge)
    return best


def dec(): eng._ck(lib.vv_decoder_forward(C.byref(w.dec), eng.latent.data_ptr(), 1, 5.0, -0.05, eng.wav.data_ptr(), eng._dec_ws.data_ptr(), eng.sp), "d")
def sem(): eng._ck(lib.vv_encoder_forward(C.byref(w.sem), eng.wav.data_ptr(), cfg.hop, eng.sem.data_ptr(), eng._sem_ws.data_ptr(), eng.sp), "s")


for spec in sys.argv[1:] or ["convffn=1"]:
    kv = [p.split("=") for p in spec.split(",")]
    for k, v in kv:
        lib.vv_tune(k.encode(), int(v))
    print(f"{spec:50s} decoder {timeit(dec):.4f} ms  semantic {timeit(sem):.4f} ms", flush=True)
