import sys, torch
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
for _ in range(3): eng.acoustic_encode(voice)
eng.stream.synchronize()
