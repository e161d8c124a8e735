"""GPU parity for the BASELINE.json configurations beyond the 1-speaker benchmark and for the branches they reach:

  * voice-prompt prefill with S > 1 ragged, padded voices (reference fixture)                      -> cfg 3 / 4
  * multi-chunk LLM prefill (L0 > chunk): `mid` fp32, and 1 040 / 1 500-token prompts at 1.5B        -> cfg 3 / 4 / 5 (512-token chunks)
  * the shipped bf16 path with its timestep quirk against the reference's OWN bf16 run (fixture)   -> every bf16 config
  * 4-speaker dialogue with `speech_end, speech_start` turn switches: `mid` vs oracle, 1.5B props  -> cfg 3
  * VibeVoice-7B shapes: head sampling at 20 and 50 steps, batch-2 decode (28 / 4 heads), streaming
    decoder + semantic frames, in bf16 and with weight-only fp8                                     -> cfg 4 / 5
  * the drop-in loading path: reference-layout checkpoint directory -> `vibevoice.*` import paths -> demo call sequence

Every assert message carries the measured error."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_rms

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


class _Tok:
    def __init__(self, v):
        self.speech_start_id, self.speech_end_id, self.speech_diffusion_id, self.eos_token_id = v - 4, v - 3, v - 2, v - 1
        self.bos_token_id, self.pad_id = None, 0


def _special(v):
    return dict(speech_start=v - 4, speech_end=v - 3, speech_diffusion=v - 2, eos=v - 1)


def _cpu(sd, prefix):
    return {k: v.float().cpu() for k, v in sd.items() if k.startswith(prefix) or k.startswith("model.speech_")}


# ---------------------------------------------------------------------------------------------------------------
# (i) voice-prompt prefill, two ragged voices, against the reference fixture
# ---------------------------------------------------------------------------------------------------------------
def test_process_speech_inputs_two_ragged_voices_vs_reference(tiny_cfg, tiny_weights):
    """modeling._process_speech_inputs on the GPU against the reference's own `_process_speech_inputs`
    (modeling_vibevoice_inference.py:149-163) for S = 2 voices of 3 hops + 700 samples and 2 hops, padded to the longer one with
    padded speech_masks, the two gaussian draws injected (tests/golden/speech_inputs_tiny.npz)."""
    _need_gpu()
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    g = load_golden("speech_inputs_tiny")
    m = VibeVoiceForConditionalGenerationInference(tiny_cfg, tiny_weights, device="cuda:0", torch_dtype=torch.float32)
    masks = torch.from_numpy(g["masks"])
    feats, conn = m._process_speech_inputs(torch.from_numpy(g["wav"]), masks, torch.from_numpy(g["std_noise"]), torch.from_numpy(g["eps_noise"]))
    m.engine.stream.synchronize()
    assert masks.sum(-1).tolist() == [4, 2] and tuple(conn.shape) == (6, tiny_cfg.hidden)
    e_f = rel_rms(feats.cpu()[masks].numpy(), g["feats"][g["masks"]])
    e_c = rel_rms(conn.cpu().numpy(), g["connected"])
    assert e_f < 1e-4, f"scaled latents of the valid frames vs reference: rel RMS {e_f:.3e}"
    assert e_c < 1e-4, f"connected voice embeddings vs reference: rel RMS {e_c:.3e}"


# ---------------------------------------------------------------------------------------------------------------
# (v) the shipped bf16 path (bf16_t_quirk on) against the reference's own bf16 run
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset", ["tiny", "mid"])
def test_bf16_head_sampling_vs_reference_bf16_run(preset):
    """Engine in bf16 with its DEFAULT timestep handling (t and the sinusoid rounded to bf16, engine.py: bf16_t_quirk) against
    tests/golden/sample_bf16_*.npz = the reference's `sample_speech_tokens` run in bf16 on the CPU.  Calibration carried by the fixture:
    the reference's own fp32-vs-bf16 difference on the same weights is 1e-2 .. 2e-2, the exact-arithmetic oracle with the same t rounding
    sits 6e-3 .. 1e-2 from the bf16 run (tests/test_oracle_vs_golden.py).  Bars: 1.5e-2 against the reference's bf16 run (its own
    rounding noise), 5e-3 against the oracle with `bf16_t` (what is left is our bf16 activation rounding in the matrix-core GEMMs)."""
    _need_gpu()
    from oracle import vv_oracle as O
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.engine import Engine
    from vibevoice_rocm_amd.synth import synth_state_dict
    cfg = VVConfig.preset(preset)
    sd = {k: torch.from_numpy(v).to(torch.bfloat16).float() for k, v in synth_state_dict(cfg, 1234).items()}   # as a bf16 checkpoint stores them
    g = load_golden(f"sample_bf16_{preset}")
    eng = Engine(cfg, sd, device="cuda:0", dtype=torch.bfloat16, use_graphs=False)
    assert eng.bf16_t_quirk
    cond, ncond, noise = (torch.from_numpy(g[k]) for k in ("cond", "ncond", "noise"))
    worst = 0.0
    for n in (10, 20):
        eng.set_steps(n)
        for cs in (1.3, 2.0):
            with torch.cuda.stream(eng.stream):
                eng.hidden2[0].copy_(cond[0].cuda()); eng.hidden2[1].copy_(ncond[0].cuda()); eng.noise_dev.copy_(noise[0].cuda())
                eng._ck(eng.lib.vv_head_sample(C.byref(eng.w.head), eng.hidden2.data_ptr(), cfg.hidden, eng.noise_dev.data_ptr(), eng.temb.data_ptr(),
                                               eng._coefs, n, cs, eng.latent.data_ptr(), eng._head_ws.data_ptr(), None, eng.sp), "vv_head_sample")
            eng.stream.synchronize()
            got = eng.latent.cpu().numpy()
            ref_b, ref_f = g[f"latent_bf16_n{n}_cfg{cs}"][0], g[f"latent_fp32_n{n}_cfg{cs}"][0]
            want = O.sample_speech_tokens(sd, cfg.as_dict(), cond, ncond, noise, cs, n, bf16_t=True)[0].numpy()
            e_ref, e_or, floor = rel_rms(got, ref_b), rel_rms(got, want), rel_rms(ref_f, ref_b)
            worst = max(worst, e_ref)
            assert e_ref < 1.5e-2, f"{preset} n={n} cfg={cs}: HIP bf16 vs reference bf16 run {e_ref:.3e} (reference fp32 vs bf16: {floor:.3e})"
            assert e_or < 5e-3, f"{preset} n={n} cfg={cs}: HIP bf16 vs oracle(bf16_t) {e_or:.3e}"
    print(f"[{preset}] worst HIP-bf16 vs reference-bf16 rel RMS {worst:.3e}")


# ---------------------------------------------------------------------------------------------------------------
# (ii) multi-chunk prefill
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mid():
    _need_gpu()
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.synth import synth_state_dict
    cfg = VVConfig.preset("mid")
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 4321).items()}
    return cfg, sd


def test_multichunk_prefill_mid_fp32_vs_oracle(mid):
    """Engine.prefill(chunk=64) over a 200-token prompt = chunks of 64, 64, 64 and 8 rows (the last one below the prefill-path
    threshold): every chunk attends to the rows cached by the earlier ones (reference: one forward over the whole prompt,
    modeling_vibevoice_inference.py:478).  Last hidden state and K/V of every layer vs the oracle, fp32."""
    from oracle import vv_oracle as O
    from vibevoice_rocm_amd.engine import Engine
    cfg, sd = mid
    eng = Engine(cfg, sd, device="cuda:0", dtype=torch.float32, use_graphs=False)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg.vocab - 8, (200,), generator=g)
    kv = O.KVCache(cfg.layers)
    h_ref = O.llm_forward(sd, cfg.as_dict(), sd["model.language_model.embed_tokens.weight"][ids], kv, 0)
    eng.begin_sequence(256, [cfg.vocab - 4, cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1])
    eng.prefill(eng.embed_ids(ids), row=0, chunk=64)
    eng.stream.synchronize()
    assert eng.lens.tolist()[0] == 200
    e = rel_rms(eng.hidden2[0].cpu().numpy(), h_ref[-1].numpy())
    assert e < 1e-4, f"last hidden after 4 chunks: rel RMS {e:.3e}"
    for layer in range(cfg.layers):
        ek = rel_rms(eng._kv_t[0][layer, 0, :, :200].cpu().numpy(), kv.k[layer].numpy())
        ev = rel_rms(eng._kv_t[1][layer, 0, :, :200].cpu().numpy(), kv.v[layer].numpy())
        assert ek < 1e-4 and ev < 1e-4, f"layer {layer}: K {ek:.3e} V {ev:.3e}"
    single = Engine(cfg, sd, device="cuda:0", dtype=torch.float32, use_graphs=False)
    single.begin_sequence(256, [cfg.vocab - 4, cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1])
    single.prefill(single.embed_ids(ids), row=0)
    single.stream.synchronize()
    e = rel_rms(eng.hidden2[0].cpu().numpy(), single.hidden2[0].cpu().numpy())
    assert e < 1e-5, f"chunked vs single-chunk prefill: rel RMS {e:.3e}"


@pytest.fixture(scope="module")
def big():
    _need_gpu()
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    cfg = VVConfig.preset("1.5b")
    sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(20)
    torch.set_num_threads(16)
    return cfg, sd, m


def test_multichunk_prefill_1p5b_1040_and_1500_tokens_vs_oracle(big):
    """cfg 3's 1 040-token prompt with the default 1 024-row chunks (1 024 + 16) and cfg 4/5's 1 500-token prompt with 512-row chunks
    (512 + 512 + 476) at 1.5B shapes, bf16: last hidden + K/V of layers 0 and 27 against ONE oracle forward over 1 500 tokens (causal:
    its first 1 040 positions are the shorter prompt's)."""
    from oracle import vv_oracle as O
    cfg, sd, m = big
    eng = m.engine
    W = _cpu(sd, "model.language_model.")
    g = torch.Generator().manual_seed(17)
    ids = torch.randint(0, 1000, (1500,), generator=g)
    kv = O.KVCache(cfg.layers)
    h_ref = O.llm_forward(W, cfg.as_dict(), W["model.language_model.embed_tokens.weight"][ids], kv, 0)
    for n, chunk in ((1040, 1024), (1500, 512)):
        eng.begin_sequence(2048, [cfg.vocab - 4, cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1])
        eng.prefill(eng.embed_ids(ids[:n]), row=0, chunk=chunk)
        eng.stream.synchronize()
        assert eng.lens.tolist()[0] == n
        e = rel_rms(eng.hidden2[0].cpu().numpy(), h_ref[n - 1].numpy())
        assert e < 2e-2, f"L0={n} chunk={chunk}: last hidden rel RMS {e:.3e}"
        for layer in (0, cfg.layers - 1):
            ek = rel_rms(eng._kv_t[0][layer, 0, :, :n].float().cpu().numpy(), kv.k[layer][:, :n].numpy())
            ev = rel_rms(eng._kv_t[1][layer, 0, :, :n].float().cpu().numpy(), kv.v[layer][:, :n].numpy())
            assert ek < 2e-2 and ev < 2e-2, f"L0={n} layer {layer}: K {ek:.3e} V {ev:.3e}"
    # one decode step on top of the 1 500-token cache (the context length of cfg 4's first frames)
    x = 0.05 * torch.randn(1, cfg.hidden, generator=g)
    p_ref = O.llm_forward(W, cfg.as_dict(), x, kv, kv.length)[0]
    with torch.cuda.stream(eng.stream):
        eng.x2[0].copy_(x[0].cuda()); eng.x2[1].copy_(x[0].cuda())
        eng.llm_forward(eng.x2, eng.lens, None, eng.hidden2)
    eng.stream.synchronize()
    e = rel_rms(eng.hidden2[0].cpu().numpy(), p_ref.numpy())
    assert e < 2e-2, f"decode step at S=1500: rel RMS {e:.3e}"


# ---------------------------------------------------------------------------------------------------------------
# (iv) 4 speakers, turn switches
# ---------------------------------------------------------------------------------------------------------------
def _four_speaker_inputs(cfg, g, frames_per_voice=(3, 2, 4, 2), text_tokens=30):
    """A prompt shaped like the processor's: text, then per speaker [speech_start, placeholders, speech_end], then text + speech_start."""
    V = cfg.vocab
    ST, SE, SD = V - 4, V - 3, V - 2
    ids, mask, voices = [], [], []
    ids += torch.randint(0, V - 8, (text_tokens,), generator=g).tolist(); mask += [False] * text_tokens
    for f in frames_per_voice:
        ids += [7, ST] + [SD] * f + [SE, 9]; mask += [False, False] + [True] * f + [False, False]
        voices.append(0.1 * torch.randn(f * cfg.hop - 321, generator=g))
    ids += torch.randint(0, V - 8, (text_tokens,), generator=g).tolist() + [ST]; mask += [False] * (text_tokens + 1)
    T, Fm = max(v.shape[0] for v in voices), max(frames_per_voice)
    wav = torch.zeros(len(voices), T)
    sm = torch.zeros(len(voices), Fm, dtype=torch.bool)
    for i, (v, f) in enumerate(zip(voices, frames_per_voice)):
        wav[i, : v.shape[0]] = v
        sm[i, :f] = True
    return torch.tensor(ids), torch.tensor(mask), wav, sm


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 2e-2)])
def test_four_speaker_turn_switching_mid_vs_oracle(mid, dtype, tol):
    """cfg 3 at `mid` shapes: 4 ragged voice prompts through _process_speech_inputs, then 4 turns of 5 frames separated by
    `speech_end, speech_start` (conv caches zeroed, negative branch reset; modeling_vibevoice_inference.py:540-563), CFG 2, 20 steps,
    whole generate() against the oracle."""
    from oracle import vv_oracle as O
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg, sd = mid
    sd_o = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()} if dtype == torch.bfloat16 else sd
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(23)
    ids, sp_mask, wav, sm = _four_speaker_inputs(cfg, g)
    forced = ([SD] * 5 + [SE, ST]) * 3 + [SD] * 5 + [SE, EOS]
    noise = torch.randn(20, cfg.latent, generator=g)
    std_noise, eps_noise = torch.randn(4, generator=g), torch.randn(4, 4, cfg.ac_dim, generator=g)
    _, conn = O.process_speech_inputs(sd_o, cfg.as_dict(), wav, sm, std_noise, eps_noise)
    ref = O.generate(sd_o, cfg.as_dict(), ids.tolist(), sp_mask, conn, _special(V), noise, cfg_scale=2.0, n_steps=20, forced_tokens=forced,
                     bf16_t=(dtype == torch.bfloat16))
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=dtype)
    m.set_ddpm_inference_steps(20)
    out = m.generate(input_ids=ids[None], speech_tensors=wav, speech_masks=sm, speech_input_mask=sp_mask[None], tokenizer=_Tok(V), cfg_scale=2.0,
                     forced_tokens=forced, noise=noise, speech_noise=(std_noise, eps_noise))
    assert out.sequences[0, ids.shape[0]:].tolist() == forced
    got, want = out.speech_outputs[0][0].cpu().numpy(), torch.cat(ref.audio).numpy()
    assert got.shape == want.shape == (20 * cfg.hop,)
    per_turn = [rel_rms(got[i * 5 * cfg.hop:(i + 1) * 5 * cfg.hop], want[i * 5 * cfg.hop:(i + 1) * 5 * cfg.hop]) for i in range(4)]
    e = rel_rms(got, want)
    assert e < tol and max(per_turn) < 2 * tol, f"4-speaker {dtype}: waveform rel RMS {e:.3e} (bar {tol}), per turn {['%.2e' % x for x in per_turn]}"
    assert m.engine.lens.tolist() == [ids.shape[0] + len(forced) - 1, 5]      # negative context = the frames since the last speech_start


def test_four_speaker_dialogue_properties_1p5b(big):
    """cfg 3's dialogue shape at 1.5B: L0 = 1 040 (two prefill chunks), 4 turns of 75 frames with `speech_end, speech_start` between
    them (300 frames).  Size-independent properties: token schedule reproduced, frame count, finiteness, two runs bit-identical,
    speculative == non-speculative launch bit for bit, positions after the run, and every turn starts from zeroed conv caches: the
    waveform of a turn does not change when the PREVIOUS turn's noise changes only through the LLM context (checked structurally:
    the first frame after a switch differs from the first frame of the run, i.e. the positive KV context carried over)."""
    cfg, sd, m = big
    V = cfg.vocab
    tok = _Tok(V)
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(29)
    ids = torch.cat([torch.randint(0, 1000, (1039,), generator=g), torch.tensor([ST])])
    forced = ([SD] * 75 + [SE, ST]) * 3 + [SD] * 75 + [SE, EOS]
    noise = torch.randn(300, cfg.latent, generator=g)
    kw = dict(input_ids=ids[None], tokenizer=tok, cfg_scale=2.0, forced_tokens=forced, noise=noise)
    a = m.generate(**kw)
    assert a.sequences[0, 1040:].tolist() == forced
    wa = a.speech_outputs[0]
    assert tuple(wa.shape) == (1, 300 * cfg.hop) and bool(torch.isfinite(wa).all())
    assert m.engine.lens.tolist() == [1040 + len(forced) - 1, 75]
    b = m.generate(**kw)
    assert torch.equal(wa, b.speech_outputs[0]), "two identical calls must be bit-identical"
    m.speculative_frames = False
    try:
        c = m.generate(**kw)
    finally:
        m.speculative_frames = True
    assert torch.equal(wa, c.speech_outputs[0]), "speculative frame launch must equal the plain loop across turn switches"
    w = wa[0].cpu().numpy().reshape(300, cfg.hop)
    assert rel_rms(w[75], w[0]) > 1e-2 and float(np.abs(w).max()) < 1e3


# ---------------------------------------------------------------------------------------------------------------
# (iii) VibeVoice-7B shapes (cfg 4: bf16, 20 steps; cfg 5: 50 steps, weight-only fp8)
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def seven():
    _need_gpu()
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    cfg = VVConfig.preset("7b")
    sd = synth_state_dict_torch(cfg, 777, device="cuda:0", dtype=torch.bfloat16)
    torch.set_num_threads(16)
    return cfg, sd


def _head_sample(eng, cfg, cond, ncond, noise, n, cs):
    eng.set_steps(n)
    with torch.cuda.stream(eng.stream):
        eng.hidden2[0].copy_(cond[0].cuda()); eng.hidden2[1].copy_(ncond[0].cuda()); eng.noise_dev.copy_(noise[0].cuda())
        eng._ck(eng.lib.vv_head_sample(C.byref(eng.w.head), eng.hidden2.data_ptr(), cfg.hidden, eng.noise_dev.data_ptr(), eng.temb.data_ptr(),
                                       eng._coefs, n, cs, eng.latent.data_ptr(), eng._head_ws.data_ptr(), None, eng.sp), "vv_head_sample")
    eng.stream.synchronize()
    return eng.latent.cpu().numpy().copy()


@pytest.mark.parametrize("quant", [None, "fp8"])
def test_7b_components_vs_oracle(seven, quant):
    """VVConfig.preset("7b") (H 3584, I 18944, 28 / 4 heads, untied lm_head, head 3584 / 10752) on the HIP path against the oracle:
    head sampling at 20 AND 50 solver steps (CFG 2), prompt prefill (80 rows) + batch-2 decode step incl. the constrained logits from the
    untied lm_head, 3 streaming decoder + semantic-encoder frames.  quant="fp8": the same with weight-only e4m3 companions on the
    streaming GEMVs, the oracle on the effective (dequantised) matrices."""
    from oracle import vv_oracle as O
    from vibevoice_rocm_amd.engine import Engine
    from vibevoice_rocm_amd.weights import fp8_effective_state_dict
    cfg, sd = seven
    eng = Engine(cfg, sd, device="cuda:0", dtype=torch.bfloat16, use_graphs=False, weight_quant=quant)
    sd_o = fp8_effective_state_dict(cfg, sd) if quant else sd
    ocfg = cfg.as_dict()
    g = torch.Generator().manual_seed(41)
    tag = f"7B {quant or 'bf16'}"
    # --- diffusion head: 20 and 50 steps
    W = _cpu(sd_o, "model.prediction_head.")
    cond, ncond, noise = torch.randn(1, cfg.hidden, generator=g), torch.randn(1, cfg.hidden, generator=g), torch.randn(1, cfg.latent, generator=g)
    for n in (20, 50):
        ref = O.sample_speech_tokens(W, ocfg, cond, ncond, noise, 2.0, n, bf16_t=True)[0].numpy()
        e = rel_rms(_head_sample(eng, cfg, cond, ncond, noise, n, 2.0), ref)
        assert e < 2e-2, f"{tag} head sampling, {n} steps: rel RMS {e:.3e}"
    del W
    # --- LLM: prefill + batch-2 decode + constrained logits (untied lm_head)
    W = _cpu(sd_o, "model.language_model.")
    W["lm_head.weight"] = sd_o["lm_head.weight"].float().cpu()
    ids = torch.randint(0, 1000, (80,), generator=g)
    emb = W["model.language_model.embed_tokens.weight"]
    kv, nkv = O.KVCache(cfg.layers), O.KVCache(cfg.layers)
    h_ref = O.llm_forward(W, ocfg, emb[ids], kv, 0)[-1]
    O.llm_forward(W, ocfg, emb[ids[:7]], nkv, 0)
    valid = [cfg.vocab - 4, cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1]
    eng.begin_sequence(256, valid)
    eng.prefill(eng.embed_ids(ids), row=0)
    eng.prefill(eng.embed_ids(ids[:7]), row=1)
    eng.stream.synchronize()
    e = rel_rms(eng.hidden2[0].cpu().numpy(), h_ref.numpy())
    assert e < 2e-2, f"{tag} prefill(80) last hidden: rel RMS {e:.3e}"
    x = 0.05 * torch.randn(1, cfg.hidden, generator=g)
    p_ref = O.llm_forward(W, ocfg, x, kv, kv.length)[0]
    n_ref = O.llm_forward(W, ocfg, x, nkv, nkv.length)[0]
    with torch.cuda.stream(eng.stream):
        eng.x2[0].copy_(x[0].cuda()); eng.x2[1].copy_(x[0].cuda())
        eng.llm_forward(eng.x2, eng.lens, None, eng.hidden2)
        eng._logits()
    eng.stream.synchronize()
    ep, en = rel_rms(eng.hidden2[0].cpu().numpy(), p_ref.numpy()), rel_rms(eng.hidden2[1].cpu().numpy(), n_ref.numpy())
    assert ep < 2e-2 and en < 2e-2, f"{tag} batch-2 decode: positive {ep:.3e} negative {en:.3e}"
    lg_ref = (p_ref @ O.lm_head_weight(W, ocfg)[valid].t()).numpy()
    el = rel_rms(eng.logits[:4].cpu().numpy(), lg_ref)
    assert not cfg.tie and el < 2e-2, f"{tag} constrained logits (untied lm_head): rel RMS {el:.3e}"
    del W, kv, nkv
    # --- streaming decoder + semantic encoder frames (stage 0 runs as fp8 GEMVs in fp8 mode)
    Wd, Ws = _cpu(sd_o, "model.acoustic_tokenizer.decoder."), _cpu(sd_o, "model.semantic_tokenizer.encoder.")
    st_d, st_s = O.ConvState(), O.ConvState()
    with torch.cuda.stream(eng.stream):
        eng.reset_speech_caches()
    for f in range(3):
        lat = torch.randn(cfg.ac_dim, generator=g)
        wav_ref = O.tokenizer_decoder(Wd, ocfg, lat[:, None], st_d)[0]
        sem_ref = O.semantic_encode(Ws, ocfg, wav_ref[None], st_s)[0]
        with torch.cuda.stream(eng.stream):
            ld, wr = lat.cuda(), wav_ref.cuda()
            eng._ck(eng.lib.vv_decoder_forward(C.byref(eng.w.dec), ld.data_ptr(), 1, 1.0, 0.0, eng.wav.data_ptr(), eng._dec_ws.data_ptr(), eng.sp), "dec")
            eng._ck(eng.lib.vv_encoder_forward(C.byref(eng.w.sem), wr.data_ptr(), cfg.hop, eng.sem.data_ptr(), eng._sem_ws.data_ptr(), eng.sp), "sem")
        eng.stream.synchronize()
        ed, es = rel_rms(eng.wav.cpu().numpy(), wav_ref.numpy()), rel_rms(eng.sem.cpu().numpy(), sem_ref.numpy())
        assert ed < 2e-2 and es < 2e-2, f"{tag} frame {f}: decoder {ed:.3e} semantic {es:.3e}"
    eng.close()


def test_7b_generate_two_speakers_properties(seven):
    """cfg 4's shape on the whole loop: 7B, 2 speakers (one turn switch), 1 500-token prompt (two prefill chunks), CFG 2, 20 steps, bf16.
    Properties: schedule reproduced, frame count, finite, graphs == eager bit for bit, and the first frame equals the composition of the
    component calls checked against the oracle above (structural: same engine code path)."""
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg, sd = seven
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(20)
    g = torch.Generator().manual_seed(43)
    ids = torch.cat([torch.randint(0, 1000, (1499,), generator=g), torch.tensor([ST])])
    forced = [SD] * 12 + [SE, ST] + [SD] * 12 + [SE, EOS]
    noise = torch.randn(24, cfg.latent, generator=g)
    kw = dict(input_ids=ids[None], tokenizer=_Tok(V), cfg_scale=2.0, forced_tokens=forced, noise=noise)
    a = m.generate(**kw)
    assert a.sequences[0, 1500:].tolist() == forced
    wa = a.speech_outputs[0]
    assert tuple(wa.shape) == (1, 24 * cfg.hop) and bool(torch.isfinite(wa).all())
    m.engine.use_graphs = False
    try:
        b = m.generate(**kw)
    finally:
        m.engine.use_graphs = True
    assert torch.equal(wa, b.speech_outputs[0]), "hipGraph replay must equal eager launches at 7B shapes"
    m.engine.close()


# ---------------------------------------------------------------------------------------------------------------
# the drop-in loading path: reference-layout checkpoint directory -> vibevoice.* imports -> demo/inference_from_file.py's calls
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tie", [True, False])
def test_drop_in_checkpoint_dir_through_reference_import_paths(tmp_path, tie):
    """Writes a checkpoint directory in the reference's layout - config.json with the reference's schema incl. the `vibepod_*` model_type
    keys (vibevoice/configs/qwen2.5_1.5b_64k.json:37,75,80,104), safetensors shards + index, preprocessor_config.json
    (scripts/convert_nnscaler_checkpoint_to_transformers.py:92-123) - with a tied and an untied-lm_head variant, then drives exactly
    demo/inference_from_file.py:340-427's call sequence through the `vibevoice.*` import paths and compares with the same weights
    handed to the constructor directly."""
    _need_gpu()
    import dataclasses
    from vibevoice.modular.modeling_vibevoice_inference import VibeVoiceForConditionalGenerationInference      # the reference's import paths
    from vibevoice.processor.vibevoice_processor import VibeVoiceProcessor
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import save_checkpoint_dir
    from vibevoice_rocm_amd.processor import SyntheticTokenizer, load_wav
    from vibevoice_rocm_amd.synth import synth_state_dict
    cfg = dataclasses.replace(VVConfig.preset("mid"), tie=tie)
    sd = {k: torch.from_numpy(v).to(torch.bfloat16) if v.ndim >= 2 else torch.from_numpy(v) for k, v in synth_state_dict(cfg, 555).items()}
    assert ("lm_head.weight" in sd) == (not tie)
    path = str(tmp_path / "ckpt")
    save_checkpoint_dir(path, cfg, sd, max_shard_bytes=4 * 2 ** 20)
    shards = [f for f in os.listdir(path) if f.endswith(".safetensors")]
    assert len(shards) >= 3 and os.path.exists(os.path.join(path, "model.safetensors.index.json"))
    j = json.load(open(os.path.join(path, "config.json")))
    assert j["model_type"] == "vibepod" and j["acoustic_tokenizer_config"]["model_type"] == "vibepod_acoustic_tokenizer"
    assert j["diffusion_head_config"]["model_type"] == "vibepod_diffusion_head" and j["decoder_config"]["model_type"] == "qwen2"
    # --- demo/inference_from_file.py:340-427
    tok = SyntheticTokenizer(cfg.vocab)
    processor = VibeVoiceProcessor.from_pretrained(path, tokenizer=tok)           # no Qwen vocabulary offline: the stand-in tokenizer is injected
    model = VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.bfloat16, device_map="cuda",
                                                                       attn_implementation="flash_attention_2")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=10)
    assert model.model.language_model.config._attn_implementation == "flash_attention_2" and model.ddpm_inference_steps == 10
    rng = np.random.Generator(np.random.PCG64(5))
    voices = [(0.05 * rng.standard_normal(2 * cfg.hop + 500)).astype(np.float32), (0.05 * rng.standard_normal(3 * cfg.hop)).astype(np.float32)]
    script = "Speaker 1: Hello there.\nSpeaker 2: Hi, how are you?"
    inputs = processor(text=[script], voice_samples=[voices], padding=True, return_tensors="pt", return_attention_mask=True)
    for k, v in inputs.items():
        if torch.is_tensor(v):
            inputs[k] = v.to("cuda")
    V = cfg.vocab
    forced = [V - 2] * 4 + [V - 3, V - 1]
    noise = torch.randn(4, cfg.latent, generator=torch.Generator().manual_seed(1))
    sn = (torch.randn(2, generator=torch.Generator().manual_seed(2)), torch.randn(2, 3, cfg.ac_dim, generator=torch.Generator().manual_seed(3)))
    outputs = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=processor.tokenizer, generation_config={"do_sample": False},
                             verbose=False, forced_tokens=forced, noise=noise, speech_noise=sn)
    wav = outputs.speech_outputs[0]
    assert tuple(wav.shape) == (1, 4 * cfg.hop) and bool(torch.isfinite(wav).all())
    assert outputs.sequences.shape[1] == inputs["input_ids"].shape[1] + len(forced)
    out_wav = str(tmp_path / "out" / "generated.wav")
    processor.save_audio(wav, output_path=out_wav)
    y = load_wav(out_wav)
    assert y.shape == (4 * cfg.hop,) and float(np.abs(y - np.clip(wav[0].float().cpu().numpy(), -1, 1)).max()) < 1e-4 + 1 / 32768
    # --- the same weights handed over directly
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference as Direct
    direct = Direct(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    direct.set_ddpm_inference_steps(10)
    ref = direct.generate(**inputs, cfg_scale=1.3, tokenizer=tok, forced_tokens=forced, noise=noise, speech_noise=sn)
    assert torch.equal(ref.speech_outputs[0], wav), "checkpoint directory and direct construction must give the same waveform bit for bit"
    assert torch.equal(ref.sequences, outputs.sequences)
    if not tie:     # the untied head is really used: logits differ from what the embedding matrix would give
        e = model.engine
        assert e.w.lm_head.data_ptr() != e.w.embed.data_ptr()
