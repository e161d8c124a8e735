"""Several row-batched generate() calls (4 dialogues, 60 frames) under rocprofv3 --kernel-trace, each call bracketed by a marker kernel
(torch.sin_ on a 1024-element tensor) and its aggregate rate printed: tools/rb_modes.py then shows, per call, how many hardware queues
carried conv-tail kernels and how much of the time kernels of different queues overlapped."""
import sys, time, types
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
args = types.SimpleNamespace(frames=60, voice_frames=203, cfg_scale=2.0)
wls = [bench.build_workload(cfg, args.frames, args.voice_frames, seed=201 + i) for i in range(4)]
ids = torch.cat([w["input_ids"] for w in wls])
kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=wls[0]["tok"], cfg_scale=2.0, forced_tokens=[w["forced"] for w in wls],
          noise=torch.stack([w["noise"] for w in wls]), speech_tensors=torch.cat([w["speech_tensors"] for w in wls]).cuda(),
          speech_masks=torch.cat([w["speech_masks"] for w in wls]), speech_input_mask=torch.cat([w["speech_input_mask"] for w in wls]),
          speech_noise=(torch.cat([w["speech_noise"][0] for w in wls]), torch.cat([w["speech_noise"][1] for w in wls])),
          generation_config={"do_sample": False}, show_progress_bar=False, max_length_times=2, row_batch=True)
mark = torch.zeros(1024, device="cuda")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.sin_(mark)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate(**kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"call {i}: {sum(o.shape[-1] for o in out.speech_outputs) / 24000.0 / dt:.1f} audio-sec/s", flush=True)
torch.sin_(mark)
torch.cuda.synchronize()
