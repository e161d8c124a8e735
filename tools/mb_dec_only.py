import sys, ctypes as C, torch
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
def dec(): eng._ck(lib.vv_decoder_forward(C.byref(w.dec), eng.latent.data_ptr(), 1, 5.0, -0.05, eng.wav.data_ptr(), eng._dec_ws.data_ptr(), eng.sp), "d")
with torch.cuda.stream(eng.stream):
    L.check(lib.vv_graph_begin(eng.sp), "b"); dec(); ge = C.c_void_p(); L.check(lib.vv_graph_end(eng.sp, C.byref(ge)), "e")
    for _ in range(20): lib.vv_graph_launch(ge, eng.sp)
    eng.stream.synchronize()
