"""Throughput experiment: S independent utterances on one GPU, one Engine + HIP stream + host thread each (weights shared)."""
import sys, time, threading, torch
sys.path.insert(0, "/root/repo")
import bench
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 225
models = [VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16) for _ in range(S)]
for m in models: m.set_ddpm_inference_steps(20)
wls = [bench.build_workload(cfg, frames, 203, seed=1 + i) for i in range(S)]
def run(i, out):
    o = bench.run_generate(models[i], wls[i], 2.0)
    out[i] = o.speech_outputs[0].shape[-1]
for s_active in range(1, S + 1):
    outs = [0] * S
    th = [threading.Thread(target=run, args=(i, outs)) for i in range(s_active)]
    for t in th: t.start()
    for t in th: t.join()          # warm (graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, outs)) for i in range(s_active)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams={s_active}: {sum(outs[:s_active]) / 24000 / dt:.2f} audio-sec/s aggregate ({dt:.2f} s)", flush=True)
