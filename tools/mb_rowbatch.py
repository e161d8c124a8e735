"""One generate() call on 4 (or 3) dialogues of the headline shape: lanes (one launch chain per dialogue) against the row-batched path
(rowbatch.py: one LLM / diffusion-head weight pass per frame for all dialogues).  Prints aggregate audio-sec/s for both."""
import sys, time, types
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
import bench
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 225
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
args = types.SimpleNamespace(frames=frames, voice_frames=203, cfg_scale=2.0)
for batch in [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('4', '3'))]:
    order = (True, False, True, False, True) if (len(sys.argv) > 3 and sys.argv[3] == 'rbfirst') else (False, True, False, True)
    for rb in order:
        r = bench.batched_leg(m, cfg, args, batch, row_batch=rb)
        print(f"batch {batch} row_batch={rb}: {r['value']} audio-sec/s (runs {r['runs']})", flush=True)
