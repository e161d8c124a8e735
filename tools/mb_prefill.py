import sys, time, torch
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
wl = bench.build_workload(cfg, 225, 203)
lib = L.load()
eng = m.engine
x0 = torch.randn(330, cfg.hidden, device="cuda")
voice = wl["speech_tensors"][0].cuda()
for small, sk in ((200, 0), (200, 1), (200, 0), (200, 1)):
    lib.vv_tune(b"mfma_tiled_dual_bk64", sk)
    eng.begin_sequence(1024, [cfg.vocab-4, cfg.vocab-3, cfg.vocab-2, cfg.vocab-1])
    for _ in range(2): eng.prefill(x0, row=0)
    eng.stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): eng.prefill(x0, row=0)
    eng.stream.synchronize()
    tp = (time.perf_counter()-t0)/5*1e3
    for _ in range(2): eng.acoustic_encode(voice)
    eng.stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.acoustic_encode(voice)
    eng.stream.synchronize()
    tv = (time.perf_counter()-t0)/3*1e3
    r = bench.first_chunk_leg(m, wl, 2.0, runs=5)
    print(f"mfma_tiled_small={small} dual_bk64={sk}: prefill {tp:.2f} ms, voice encode {tv:.2f} ms, first chunk {r['p50_ms']}")
