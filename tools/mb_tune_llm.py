"""LLM step time (graph replay, unprofiled) under vv_tune settings: python tools/mb_tune_llm.py key=value ..."""
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
eng = m.engine; lib = eng.lib
V = cfg.vocab
eng.begin_sequence(1024, [V - 4, V - 3, V - 2, V - 1])
eng.prefill(torch.randn(440, cfg.hidden, device="cuda"), row=0)
eng.prefill(torch.randn(110, cfg.hidden, device="cuda"), row=1)
eng.stream.synchronize()
for spec in sys.argv[1:] or ["attn_waves=8"]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        lib.vv_tune(k.encode(), int(v))
    with torch.cuda.stream(eng.stream):
        lens0 = eng.lens.clone()
        L.check(lib.vv_graph_begin(eng.sp), "b"); eng._seq_A(V - 4, V - 2); ge = C.c_void_p(); L.check(lib.vv_graph_end(eng.sp, C.byref(ge)), "e")
        for _ in range(3): lib.vv_graph_launch(ge, eng.sp)
        eng.stream.synchronize()
        best = 1e9
        for _ in range(4):
            eng.lens.copy_(lens0)
            t0 = time.perf_counter()
            for _ in range(30):
                lib.vv_graph_launch(ge, eng.sp)
            eng.stream.synchronize()
            best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
        eng.lens.copy_(lens0)
        lib.vv_graph_destroy(ge)
    print(f"{spec:40s} llm step {best:.4f} ms", flush=True)
