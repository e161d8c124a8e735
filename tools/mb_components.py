import sys, time, ctypes as C, torch
sys.path.insert(0, "/root/repo")
import bench
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
eng = m.engine; lib = eng.lib; w = eng.w
V = cfg.vocab
eng.begin_sequence(1024, [V-4, V-3, V-2, V-1])
eng.prefill(torch.randn(440, cfg.hidden, device="cuda"), row=0)
eng.prefill(torch.randn(110, cfg.hidden, device="cuda"), row=1)
eng.stream.synchronize()
def timeit(name, fn, reps=50):
    with torch.cuda.stream(eng.stream):
        lens0 = eng.lens.clone()
        L.check(lib.vv_graph_begin(eng.sp), "b"); fn(); ge = C.c_void_p(); L.check(lib.vv_graph_end(eng.sp, C.byref(ge)), "e")
        for _ in range(3): lib.vv_graph_launch(ge, eng.sp)
        eng.stream.synchronize()
        eng.lens.copy_(lens0)
        t0 = time.perf_counter()
        for _ in range(reps):
            lib.vv_graph_launch(ge, eng.sp)
        eng.stream.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        eng.lens.copy_(lens0)
    print(f"{name:28s} {dt:7.3f} ms")
    return dt
tot = 0
tot += timeit("A: llm step(R=2)+token", lambda: eng._seq_A(V-4, V-2), reps=30)
def head(): eng._ck(lib.vv_head_sample(C.byref(w.head), eng.hidden2.data_ptr(), cfg.hidden, eng.noise_dev.data_ptr(), eng.temb.data_ptr(), eng._coefs, eng.n_steps, 2.0, eng.latent.data_ptr(), eng._head_ws.data_ptr(), None, eng.sp), "h")
tot += timeit("B1: head sample (20 steps)", head)
def dec(): eng._ck(lib.vv_decoder_forward(C.byref(w.dec), eng.latent.data_ptr(), 1, 5.0, -0.05, eng.wav.data_ptr(), eng._dec_ws.data_ptr(), eng.sp), "d")
tot += timeit("B2: acoustic decoder frame", dec)
def sem(): eng._ck(lib.vv_encoder_forward(C.byref(w.sem), eng.wav.data_ptr(), cfg.hop, eng.sem.data_ptr(), eng._sem_ws.data_ptr(), eng.sp), "s")
tot += timeit("B3: semantic encoder frame", sem)
def conn():
    eng._ck(lib.vv_connector_forward(C.byref(w.ac_conn), eng.latent.data_ptr(), 1, eng.x2.data_ptr(), 0, eng.conn_ws.data_ptr(), eng.sp), "c")
    eng._ck(lib.vv_connector_forward(C.byref(w.sem_conn), eng.sem.data_ptr(), 1, eng.x2.data_ptr(), 1, eng.conn_ws.data_ptr(), eng.sp), "c")
tot += timeit("B4: connectors", conn)
print("sum", round(tot, 3), "ms/frame")
