import sys, torch
sys.path.insert(0, "/root/repo")
from vibevoice_rocm_amd.config import VVConfig
from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
from vibevoice_rocm_amd.synth import synth_state_dict_torch
cfg = VVConfig.preset("1.5b")
sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
m.set_ddpm_inference_steps(20)
V = cfg.vocab
class Tok:
    speech_start_id, speech_end_id, speech_diffusion_id, eos_token_id, bos_token_id, pad_id = V - 4, V - 3, V - 2, V - 1, None, 0
D, E, S, EOS = V - 2, V - 3, V - 4, V - 1
g = torch.Generator().manual_seed(31)
lens = [50, 37]
prompts = [torch.cat([torch.randint(0, 1000, (n - 1,), generator=g), torch.tensor([S])]) for n in lens]
Lp = max(lens)
ids = torch.stack([torch.cat([torch.full((Lp - n,), 0), p]) for n, p in zip(lens, prompts)])
mask = torch.stack([torch.cat([torch.zeros(Lp - n, dtype=torch.long), torch.ones(n, dtype=torch.long)]) for n in lens])
forced = [[D] * 6 + [E, EOS], [D] * 4 + [E, EOS]]
noise = torch.randn(2, 8, cfg.latent, generator=g)
for spec in (True, False):
    for graphs in (True, False):
        m.speculative_frames = spec
        for e in m._lanes: e.use_graphs = graphs
        m._use_graphs = graphs
        out = m.generate(input_ids=ids, attention_mask=mask, tokenizer=Tok(), cfg_scale=2.0, forced_tokens=forced, noise=noise)
        for b in range(2):
            one = m.generate(input_ids=prompts[b][None], tokenizer=Tok(), cfg_scale=2.0, forced_tokens=forced[b], noise=noise[b])
            a, c = out.speech_outputs[b][0].cpu().view(-1, cfg.hop), one.speech_outputs[0][0].cpu().view(-1, cfg.hop)
            bad = [i for i in range(a.shape[0]) if not torch.equal(a[i], c[i])]
            print(f"spec={spec} graphs={graphs} sample {b}: frames {a.shape[0]} differing {bad}", flush=True)
