"""End-to-end bf16 'mid' generate against the fp32 oracle with the fused Block1D kernel on and off (per-frame rel RMS)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import rel_rms
from test_hip_parity import _Tok
from oracle import vv_oracle as O
from vibevoice_rocm_amd import _lib as L
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.synth import synth_state_dict
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
cfg = VVConfig.preset("mid")
sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 1234).items()}
sd_o = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()}
V = cfg.vocab; ST, E, D, EOS = V - 4, V - 3, V - 2, V - 1
special = dict(speech_start=ST, speech_end=E, speech_diffusion=D, eos=EOS)
g = torch.Generator().manual_seed(5)
ids = torch.randint(0, V - 8, (40,), generator=g)
forced = [ST] + [D] * 6 + [E, ST] + [D] * 3 + [E, EOS]
noise = torch.randn(9, cfg.latent, generator=g)
voice = 0.1 * torch.randn(1, 2 * cfg.hop + 999, generator=g)
sp_mask = torch.zeros(40, dtype=torch.bool); sp_mask[5:8] = True
speech_masks = torch.ones(1, 3, dtype=torch.bool)
std_noise, eps_noise = torch.randn(1, generator=g), torch.randn(1, 3, cfg.ac_dim, generator=g)
ocfg = cfg.as_dict()
_, conn = O.process_speech_inputs(sd_o, ocfg, voice, speech_masks, std_noise, eps_noise)
ref = O.generate(sd_o, ocfg, ids.tolist(), sp_mask, conn, special, noise, cfg_scale=2.0, n_steps=20, forced_tokens=forced)
ref_wav = torch.cat(ref.audio).numpy()
lib = L.load()
print("filters", cfg.n_filters if hasattr(cfg, "n_filters") else "?", "hop", cfg.hop)
for fused in (1, 0, 1):
    lib.vv_tune(b"block1d_fused", fused)
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.engine.bf16_t_quirk = False; m.engine.n_steps = 0; m.set_ddpm_inference_steps(20)
    out = m.generate(input_ids=ids[None], speech_tensors=voice, speech_masks=speech_masks, speech_input_mask=sp_mask[None],
                     tokenizer=_Tok(ST, E, D, EOS), cfg_scale=2.0, forced_tokens=forced, noise=noise, speech_noise=(std_noise, eps_noise))
    got = out.speech_outputs[0][0].cpu().numpy()
    per = [rel_rms(got[i * cfg.hop:(i + 1) * cfg.hop], ref_wav[i * cfg.hop:(i + 1) * cfg.hop]) for i in range(9)]
    print("fused", fused, "total", rel_rms(got, ref_wav), "per frame", " ".join(f"{p:.4f}" for p in per))
