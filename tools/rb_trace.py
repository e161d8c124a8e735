"""Kernel trace of replayed row-batched graphs: rocprofv3 --kernel-trace -- python tools/rb_trace.py, then tools/rb_trace_sum.py <csv>.
Replays graph A (8-row LLM step + tail) 20 times, then graph H (8-row diffusion sampling) 20 times, separated by marker kernels."""
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
args = types.SimpleNamespace(frames=30, voice_frames=203, cfg_scale=2.0)
wls = [bench.build_workload(cfg, args.frames, args.voice_frames, seed=201 + i) for i in range(4)]
ids = torch.cat([w["input_ids"] for w in wls])
kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=wls[0]["tok"], cfg_scale=2.0, forced_tokens=[w["forced"] for w in wls],
          noise=torch.stack([w["noise"] for w in wls]), speech_tensors=torch.cat([w["speech_tensors"] for w in wls]).cuda(),
          speech_masks=torch.cat([w["speech_masks"] for w in wls]), speech_input_mask=torch.cat([w["speech_input_mask"] for w in wls]),
          speech_noise=(torch.cat([w["speech_noise"][0] for w in wls]), torch.cat([w["speech_noise"][1] for w in wls])),
          generation_config={"do_sample": False}, show_progress_bar=False, max_length_times=2, row_batch=True)
m.generate(**kw)
rb = m._rowbatch[(4, 0)]
lib = rb.lib
for b in range(4):
    rb.set_active(b, False)
torch.cuda.synchronize()
gA = [g for k, g in rb._graphs.items() if k[0] == "A"][0]
gH = [g for k, g in rb._graphs.items() if k[0] == "H"][0]
mark = torch.zeros(1024, device="cuda")
with torch.cuda.stream(rb.stream):
    for g in (gA, gH):
        torch.sin_(mark)                      # marker
        for _ in range(20):
            lib.vv_graph_launch(g, rb.sp)
    torch.sin_(mark)
torch.cuda.synchronize()
