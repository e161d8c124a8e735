import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden, rel_rms
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.synth import synth_state_dict
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
cfg = VVConfig.preset("tiny")
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()}
g = load_golden("loop_trace_tiny")
ST, E, D, EOS = [int(v) for v in g["special"]]
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.float32, use_graphs=False)
m.set_ddpm_inference_steps(10)
eng = m.engine
ctx = torch.cuda.stream(eng.stream); ctx.__enter__()
_, conn = m._process_speech_inputs(torch.from_numpy(g["voice"]), torch.from_numpy(g["speech_masks"]), torch.from_numpy(g["std_noise"]), torch.from_numpy(g["eps_noise"]))
ids = torch.from_numpy(g["ids"])
eng.cfg_scale = float(g["cfg_scale"])
eng.begin_sequence(64, [ST, E, D, EOS])
x0 = eng.embed_ids(ids)
with torch.cuda.stream(eng.stream):
    x0[torch.from_numpy(g["speech_input_mask"]).cuda()] = conn
forced = g["forced"].tolist()
frame = 0
for step, f in enumerate(forced):
    if step == 0:
        eng.prefill(x0, row=0)
        tok = eng.first_token(ST, D, f)
    else:
        tok = eng.step_decode(ST, D, f)
    eng.stream.synchronize()
    lg = eng.logits[:4].cpu().numpy()
    print(step, tok, "logits rel", rel_rms(lg, g["logits"][step]), "lens", eng.lens.tolist())
    if tok == EOS: break
    if tok == E:
        with torch.cuda.stream(eng.stream): eng.reset_speech_caches()
    if tok == D:
        print("   cond", rel_rms(eng.hidden2[0].cpu().numpy(), g["cond"][frame]), "ncond", rel_rms(eng.hidden2[1].cpu().numpy(), g["ncond"][frame]))
        eng.step_speech(torch.from_numpy(g["noise"][frame]))
        eng.stream.synchronize()
        print("   latent", rel_rms(eng.latent.cpu().numpy(), g["latent"][frame]), "wav", rel_rms(eng.wav.cpu().numpy(), g["wav"][frame]), "sem", rel_rms(eng.sem.cpu().numpy(), g["sem"][frame]))
        frame += 1
    else:
        eng.step_embed()
    eng.stream.synchronize()
    print("   next_embeds", rel_rms(eng.x2[0].cpu().numpy(), g["next_embeds"][step]), rel_rms(eng.x2[1].cpu().numpy(), g["next_embeds"][step]))
