"""voice-prompt encode time (27 s prompt, no profiler) under vv_tune settings: python tools/mb_voice_tune.py key=value ..."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
wl = bench.build_workload(cfg, 225, 203)
voice = wl["speech_tensors"][0].cuda()
eng = m.engine
for spec in sys.argv[1:] or ["block1d_fused=1"]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        eng.lib.vv_tune(k.encode(), int(v))
    for _ in range(3):
        eng.acoustic_encode(voice)
    eng.stream.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        eng.acoustic_encode(voice)
        eng.stream.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{spec:40s} voice encode {best * 1e3:.3f} ms", flush=True)
