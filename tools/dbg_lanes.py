"""Why does the second multi-lane leg of bench.py run slow?  Batched generate x4 repeatedly, on the same lanes and on fresh ones."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch

cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 1234, device="cuda:0", dtype=torch.bfloat16)
model = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
model.set_ddpm_inference_steps(20)
args = argparse.Namespace(frames=225, voice_frames=203, cfg_scale=2.0)
for rnd in range(3):
    r = bench.batched_leg(model, cfg, args, 4)
    print("round", rnd, "lanes", [hex(e.stream.cuda_stream) for e in model._lanes], r["value"], flush=True)
    if rnd == 0:
        continue
    if rnd == 1:
        model._lanes = model._lanes[:1]          # fresh lanes next round (old ones freed)
        import gc; gc.collect()
r = bench.batched_leg(model, cfg, args, 8)
print("round x8", [hex(e.stream.cuda_stream) for e in model._lanes], r["value"], flush=True)
