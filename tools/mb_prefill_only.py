import sys, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
eng = m.engine
x0 = torch.randn(330, cfg.hidden, device="cuda")
eng.begin_sequence(1024, [cfg.vocab-4, cfg.vocab-3, cfg.vocab-2, cfg.vocab-1])
for _ in range(4): eng.prefill(x0, row=0)
eng.stream.synchronize()
