"""LLM decode step (graph A) at long contexts with the per-head and the grouped (matrix-core) decode attention: `vv_tune attn_gqa 0/1`."""
import sys, time, ctypes as C, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
model = sys.argv[1] if len(sys.argv) > 1 else "1.5b"
cfg = VVConfig.preset(model)
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
eng = m.engine; lib = eng.lib; V = cfg.vocab
SWEEP = ((440, 1024), (440, 2048), (1800, 2048), (3600, 4096), (5400, 6144), (7200, 8192), (12000, 12288), (16000, 16384), (32000, 32768)) + (((64000, 65536),) if model == '1.5b' else ())
for S, smax in SWEEP:
    eng.begin_sequence(smax, [V-4, V-3, V-2, V-1])
    with torch.cuda.stream(eng.stream):
        eng.lens.copy_(torch.tensor([S, S // 3], dtype=torch.int32))
    res = []
    for gqa in (0, 1, 2) if len(sys.argv) < 3 else (1, 1001, 1002):     # argv[2] = "keys": grouped kernel at 1024 / 512 / 256 keys per split
        lib.vv_tune(b"attn_gqa", 1 if gqa > 2 else gqa)
        lib.vv_tune(b"attn_gqa_keys", {1: 1024, 1001: 512, 1002: 256}.get(gqa, 1024))
        with torch.cuda.stream(eng.stream):
            lens0 = eng.lens.clone()
            L.check(lib.vv_graph_begin(eng.sp), "b"); eng._seq_A(V-4, V-2); ge = C.c_void_p(); L.check(lib.vv_graph_end(eng.sp, C.byref(ge)), "e")
            for _ in range(3):
                lib.vv_graph_launch(ge, eng.sp); eng.lens.copy_(lens0)
            eng.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                lib.vv_graph_launch(ge, eng.sp); eng.lens.copy_(lens0)
            eng.stream.synchronize()
            res.append((time.perf_counter() - t0) / 20 * 1e3)
            lib.vv_graph_destroy(ge)
    print(f"{model} S={S} s_max={eng.kv.s_max}: LLM step per-head {res[0]:.3f} ms, grouped(split only) {res[1]:.3f} ms, grouped(always) {res[2]:.3f} ms", flush=True)
lib.vv_tune(b"attn_gqa", 1)
lib.vv_tune(b"attn_gqa_keys", 1024)
