"""A short row-batched generate() (4 dialogues, 40 frames) for a kernel trace: rocprofv3 --kernel-trace -- python tools/rb_run.py"""
import sys, types
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
import bench
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
args = types.SimpleNamespace(frames=40, voice_frames=203, cfg_scale=2.0)
print(bench.batched_leg(m, cfg, args, 4, row_batch=True)["value"])
