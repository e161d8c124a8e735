"""GPU parity of the row-batched decode path (dialogues batched into the row dimension of the LLM / diffusion-head weight passes):
csrc/vv_gemv_rows.hip against torch, the batched composites against their single-dialogue forms, and generate(row_batch=True) against the
lanes.  The matrix-core GEMV carries activations as bf16 hi + lo parts (2^-17 per element) where the 1..4-row kernels keep fp32, so the
comparisons are to a tolerance, not bit for bit; the measured errors go into the parity record (conftest.rel_rms `what=`)."""
import ctypes as C

import pytest
import torch

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vibevoice_rocm_amd import _lib as L
    lb = L.load()
    L.check(lb.vv_init(), "vv_init")
    return lb


@pytest.fixture(scope="module")
def big():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    cfg = VVConfig.preset("1.5b")
    sd = synth_state_dict_torch(cfg, 2024, device="cuda:0", dtype=torch.bfloat16)
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(20)
    return cfg, sd, m


class _Tok:
    def __init__(self, vocab):
        self.speech_start_id, self.speech_end_id, self.speech_diffusion_id, self.eos_token_id = vocab - 4, vocab - 3, vocab - 2, vocab - 1
        self.bos_token_id = None
        self.pad_id = vocab - 5


def _frag(w):
    from vibevoice_rocm_amd.weights import DeviceWeights
    return DeviceWeights.frag_major(w)


@pytest.mark.parametrize("m_rows", [4, 5, 8])
@pytest.mark.parametrize("frag", [False, True])
@pytest.mark.parametrize("shape", [(4608, 1536, True, 1, True, False), (1536, 4608, False, 0, False, True), (2048, 1536, False, 1, False, False),
                                   (1536, 1536, False, 0, False, True), (8960, 1536, True, 1, False, False), (1536, 8960, False, 0, False, True),
                                   (18944, 3584, True, 1, False, False), (3584, 18944, False, 0, False, True), (1008, 1536, False, 1, False, False)])
def test_rows_gemv_vs_torch(lib, shape, frag, m_rows):
    """vv_linear with 5..8 rows on the matrix-core GEMV (process-wide split-K scratch switched on for the public entry point) against an fp64
    torch reference on the same bf16 weights: RMSNorm (+ adaLN modulate) prologues, SwiGLU / gate / residual / bias epilogues, whole-row,
    persistent and split-K (ticket) forms, row-major and fragment-major weights, N not a multiple of the 16-row tile."""
    from vibevoice_rocm_amd import _lib as L
    n, k, dual, pro, mod, epi = shape
    if frag and n % 16:
        pytest.skip("fragment-major copies need N % 16 == 0")
    L.check(lib.vv_tune(b"gemv_rows_scratch", 1), "scratch")
    try:
        m = m_rows
        torch.manual_seed(m * 1000 + n + k)
        x = torch.randn(m, k, device="cuda") * 1.5
        w = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
        w2 = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16() if dual else None
        nw = torch.rand(k, device="cuda") + 0.5
        sh, sc = torch.randn(m, k, device="cuda") * 0.2, torch.randn(m, k, device="cuda") * 0.2
        gate, res, b = torch.randn(m, n, device="cuda"), torch.randn(m, n, device="cuda"), torch.randn(n, device="cuda")
        out = torch.zeros(m, n, device="cuda")
        wf, w2f = (_frag(w) if frag else w), ((_frag(w2) if frag else w2) if dual else None)
        a = L.LinArgs()
        a.x, a.ldx, a.m = x.data_ptr(), k, m
        a.n, a.k, a.wdt = n, k, L.VV_BF16
        a.w = wf.data_ptr()
        a.flags = L.LIN_W_FRAG if frag else 0
        a.out, a.ldo = out.data_ptr(), n
        a.pro, a.eps = pro, 1e-5
        if pro == 1:
            a.norm_w = nw.data_ptr()
        if mod:
            a.mod_shift, a.mod_scale, a.ld_mod = sh.data_ptr(), sc.data_ptr(), k
        if dual:
            a.w2, a.act = w2f.data_ptr(), 2
        if epi:
            a.gate, a.gate_ld, a.res, a.ldres = gate.data_ptr(), n, res.data_ptr(), n
        else:
            a.bias = b.data_ptr()
        for _ in range(2):          # twice: the tickets of the split-K form must be left ready for the next launch
            L.check(lib.vv_linear(C.byref(a), torch.cuda.current_stream().cuda_stream), "vv_linear")
        torch.cuda.synchronize()
        xd = x.double()
        if pro == 1:
            xd = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-5) * nw.double()
            if mod:
                xd = xd * (1 + sc.double()) + sh.double()
        y = xd @ w.double().T
        if not epi:
            y = y + b.double()
        if dual:
            y = torch.nn.functional.silu(y) * (xd @ w2.double().T)
        if epi:
            y = y * gate.double() + res.double()
        err = rel_rms(out.double().cpu().numpy(), y.cpu().numpy())
        assert err < 2e-5, f"rows GEMV m={m} n={n} k={k} dual={dual} frag={frag}: rel RMS {err:.3e}"
        if epi:
            # residual in place (res == out), as the composites call it: long K then adds every K slice to `out` with fp32 atomics
            for atomic in (1, 0):
                lib.vv_tune(b"gemv_rows_atomic", atomic)
                out2 = res.clone()
                a.res, a.out = out2.data_ptr(), out2.data_ptr()
                L.check(lib.vv_linear(C.byref(a), torch.cuda.current_stream().cuda_stream), "vv_linear")
                torch.cuda.synchronize()
                err = rel_rms(out2.double().cpu().numpy(), y.cpu().numpy())
                assert err < 2e-5, f"rows GEMV in place (atomic={atomic}) m={m} n={n} k={k} frag={frag}: rel RMS {err:.3e}"
    finally:
        lib.vv_tune(b"gemv_rows_atomic", 1)
        lib.vv_tune(b"gemv_rows_scratch", 0)


def test_head_sample_batch_vs_single(big):
    """vv_head_sample_batch (4 utterances, 8 rows per head matrix pass) against vv_head_sample per utterance."""
    cfg, sd, m = big
    eng = m.engine
    lb = eng.lib
    eng.w.ensure_frag()
    B = 4
    g = torch.Generator().manual_seed(11)
    cond = torch.randn(2 * B, cfg.hidden, generator=g).cuda()
    noise = torch.randn(B, cfg.latent, generator=g).cuda()
    with torch.cuda.stream(eng.stream):
        ws = torch.empty(lb.vv_head_ws_bytes_batch(C.byref(eng.w.head), 20, B), dtype=torch.uint8, device="cuda")
        lat = torch.zeros(B, cfg.latent, device="cuda")
        eng._ck(lb.vv_head_sample_batch(C.byref(eng.w.head), cond.data_ptr(), cfg.hidden, noise.data_ptr(), cfg.latent, eng.temb.data_ptr(), eng._coefs, 20, 2.0,
                                        lat.data_ptr(), cfg.latent, B, ws.data_ptr(), eng.sp), "vv_head_sample_batch")
        one = torch.zeros(B, cfg.latent, device="cuda")
        for b in range(B):
            eng._ck(lb.vv_head_sample(C.byref(eng.w.head), cond[2 * b:].data_ptr(), cfg.hidden, noise[b].data_ptr(), eng.temb.data_ptr(), eng._coefs, 20, 2.0,
                                      one[b].data_ptr(), eng._head_ws.data_ptr(), None, eng.sp), "vv_head_sample")
    eng.stream.synchronize()
    for b in range(B):
        err = rel_rms(lat[b].cpu().numpy(), one[b].cpu().numpy(), what=f"row-batched head sampling 1.5B, utterance {b} of 4, vs the single-utterance sampler")
        assert err < 2e-3, f"utterance {b}: rel RMS {err:.3e}"


def test_decode_step_8_rows_vs_batch2(big):
    """One row-batched decode step (8 rows, one KV cache with 8 rows, vv_llm_tail_batch) against the batch-2 step of every dialogue on its
    own engine state: hidden rows, constrained logits, tokens, positions."""
    from vibevoice_rocm_amd.rowbatch import RowBatch
    cfg, sd, m = big
    tok = _Tok(cfg.vocab)
    ST, SD = tok.speech_start_id, tok.speech_diffusion_id
    valid = [ST, tok.speech_end_id, SD, tok.eos_token_id]
    B = 4
    lanes = [m._lane(b) for b in range(B)]
    rb = RowBatch(lanes)
    g = torch.Generator().manual_seed(17)
    lens = [50, 37, 44, 29]
    prompts = [torch.cat([torch.randint(0, 1000, (n - 1,), generator=g), torch.tensor([ST])]) for n in lens]
    rb.begin(128, valid, 2.0)
    ref_h, ref_tok, ref_logits, ref_lens = [], [], [], []
    eng = m.engine
    for b in range(B):
        eng.begin_sequence(128, valid)
        x0 = eng.embed_ids(prompts[b])
        eng.prefill(x0, row=0, pos0=0)
        t0 = eng.first_token(ST, SD, SD)
        eng.prefill(eng.embed_ids(torch.tensor([ST])), row=1, pos0=0)
        with torch.cuda.stream(eng.stream):
            xin = torch.randn(cfg.hidden, generator=g).cuda() * 0.5
            eng.x2[0].copy_(xin); eng.x2[1].copy_(xin)
        t1 = eng.step_decode(ST, SD, None)
        eng.stream.synchronize()
        ref_h.append(eng.hidden2.clone()); ref_tok.append(t1); ref_logits.append(eng.logits[:4].clone()); ref_lens.append(eng.lens.clone())
        # the same on the shared cache
        rb.prefill(b, x0)
        assert rb.first_token(b, SD) == t0 == SD
        rb.prefill(b, eng.embed_ids(torch.tensor([ST])), neg=True)
        with torch.cuda.stream(rb.stream):
            rb.x[2 * b].copy_(xin); rb.x[2 * b + 1].copy_(xin)
    rb.decode_begin(ST, SD, {b: None for b in range(B)})
    toks = rb.decode_end()
    rb.synchronize()
    for b in range(B):
        eh = rel_rms(rb.hidden[2 * b: 2 * b + 2].cpu().numpy(), ref_h[b].cpu().numpy(), what=f"row-batched decode step 1.5B, dialogue {b} of 4: hidden rows vs the batch-2 step")
        el = rel_rms(rb.logits[b, :4].cpu().numpy(), ref_logits[b].cpu().numpy())
        assert eh < 5e-3 and el < 5e-3, f"dialogue {b}: hidden {eh:.3e} logits {el:.3e}"
        assert toks[b] == ref_tok[b]
        assert rb.lens[2 * b: 2 * b + 2].tolist() == ref_lens[b].tolist()
    # a finished dialogue keeps its positions
    rb.set_active(2, False)
    before = rb.lens.clone()
    rb.decode_begin(ST, SD, {b: SD for b in range(B)})
    rb.decode_end()
    rb.synchronize()
    after = rb.lens
    assert after[4:6].tolist() == before[4:6].tolist() and after[0].item() == before[0].item() + 1 and after[6].item() == before[6].item() + 1
    rb.close()


def test_generate_row_batch_vs_lanes(big):
    """generate() on 4 left-padded dialogues with different token schedules (one ends early, one switches turns: mis-speculated frames are
    rolled back per dialogue): the row-batched path against the lanes - same sequences, same chunk delivery, waveforms to the rounding of the
    matrix-core GEMV through the autoregressive loop."""
    from vibevoice_rocm_amd.streamer import AudioStreamer
    cfg, sd, m = big
    tok = _Tok(cfg.vocab)
    D, E, S, EOS = tok.speech_diffusion_id, tok.speech_end_id, tok.speech_start_id, tok.eos_token_id
    g = torch.Generator().manual_seed(31)
    lens = [50, 37, 44, 29]
    prompts = [torch.cat([torch.randint(0, 1000, (n - 1,), generator=g), torch.tensor([S])]) for n in lens]
    Lp = max(lens)
    ids = torch.stack([torch.cat([torch.full((Lp - n,), tok.pad_id), p]) for n, p in zip(lens, prompts)])
    mask = torch.stack([torch.cat([torch.zeros(Lp - n, dtype=torch.long), torch.ones(n, dtype=torch.long)]) for n in lens])
    forced = [[D] * 6 + [E, EOS], [D] * 2 + [E, EOS], [D] * 3 + [E, S] + [D] * 2 + [E, EOS], [D] * 5 + [E, EOS]]
    noise = torch.randn(4, 8, cfg.latent, generator=g)
    outs, events = {}, {}
    for rbm in (False, True):
        st = AudioStreamer(batch_size=4)
        ev = []
        put0 = st.put

        def spy(chunks, idx, put0=put0, ev=ev):
            ev.append([int(i) for i in idx])
            put0(chunks, idx)
        st.put = spy
        outs[rbm] = m.generate(input_ids=ids, attention_mask=mask, tokenizer=tok, cfg_scale=2.0, forced_tokens=forced, noise=noise, audio_streamer=st,
                               row_batch=rbm)
        events[rbm] = ev
        if rbm:
            for b in range(4):
                got = torch.cat([c.reshape(-1) for c in st.get_stream(b)])
                assert torch.equal(got, outs[rbm].speech_outputs[b][0].cpu()), f"sample {b}: streamed chunks"
            assert st.finished_flags == [True] * 4
    assert (4, 0) in m._rowbatch
    assert outs[True].sequences.tolist() == outs[False].sequences.tolist()
    assert events[True] == events[False]
    for b in range(4):
        a, r = outs[True].speech_outputs[b], outs[False].speech_outputs[b]
        assert a.shape == r.shape
        err = rel_rms(a.float().cpu().numpy(), r.float().cpu().numpy(), what=f"generate() on 4 dialogues 1.5B bf16, row-batched vs lanes, waveform of dialogue {b}")
        assert err < 1e-2, f"dialogue {b}: waveform rel RMS {err:.3e}"
    # three dialogues: 6 rows
    out3 = m.generate(input_ids=ids[:3], attention_mask=mask[:3], tokenizer=tok, cfg_scale=2.0, forced_tokens=forced[:3], noise=noise[:3], row_batch=True)
    for b in range(3):
        err = rel_rms(out3.speech_outputs[b].float().cpu().numpy(), outs[False].speech_outputs[b].float().cpu().numpy())
        assert err < 1e-2, f"batch of 3, dialogue {b}: waveform rel RMS {err:.3e}"


def test_generate_row_batch_7b_shapes_vs_lanes():
    """VibeVoice-7B shapes (hidden 3584, 28 / 4 heads, head D = 3584): K = 3584 / 18944 take other forms of the GEMV (K split over two blocks
    with the SwiGLU epilogue in the ticket merge, the ticket with the RMSNorm scalar factored out, 13-way split of the down projection; the
    head's modulated SwiGLU GEMV falls back to the streaming kernels) - the row-batched call must still agree with the lanes."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    cfg = VVConfig.preset("7b")
    sd = synth_state_dict_torch(cfg, 777, device="cuda:0", dtype=torch.bfloat16)
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(10)
    tok = _Tok(cfg.vocab)
    D, E, S, EOS = tok.speech_diffusion_id, tok.speech_end_id, tok.speech_start_id, tok.eos_token_id
    g = torch.Generator().manual_seed(41)
    ids = torch.stack([torch.cat([torch.randint(0, 1000, (23,), generator=g), torch.tensor([S])]) for _ in range(3)])
    forced = [[D] * 3 + [E, EOS], [D] * 2 + [E, EOS], [D] * 3 + [E, EOS]]
    noise = torch.randn(3, 4, cfg.latent, generator=g)
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=tok, cfg_scale=2.0, forced_tokens=forced, noise=noise)
    lanes = m.generate(row_batch=False, **kw)
    rows = m.generate(row_batch=True, **kw)
    assert (3, 0) in m._rowbatch and rows.sequences.tolist() == lanes.sequences.tolist()
    for b in range(3):
        err = rel_rms(rows.speech_outputs[b].float().cpu().numpy(), lanes.speech_outputs[b].float().cpu().numpy(),
                      what=f"generate() on 3 dialogues at 7B shapes bf16, row-batched vs lanes, waveform of dialogue {b}")
        assert err < 2e-2, f"7B shapes, dialogue {b}: waveform rel RMS {err:.3e}"
    del m
    torch.cuda.empty_cache()


def test_generate_row_batch_mid_bf16_vs_oracle():
    """The row-batched path against the CPU oracle (not only against the lanes): `mid` shapes (hidden 512, 3 layers, head D 512: the 8-row GEMV's
    whole-row, split-K ticket and atomic forms all occur), bf16 weights, 3 dialogues of the 4-speaker prompt with different turn schedules and
    noise in ONE generate() call; every dialogue against its own oracle run at the bf16 bar of the single-dialogue test (2e-2), with the lanes'
    error next to it in the parity record."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import vv_oracle as O
    from test_hip_configs import _four_speaker_inputs, _special
    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict
    cfg = VVConfig.preset("mid")
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 4321).items()}
    sd_o = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()}
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(29)
    ids, sp_mask, wav, sm = _four_speaker_inputs(cfg, g)
    forced = [([SD] * 4 + [SE, ST]) * 2 + [SD] * 4 + [SE, EOS], [SD] * 7 + [SE, ST] + [SD] * 3 + [SE, EOS], [SD] * 5 + [SE, EOS]]
    noise = torch.randn(3, 12, cfg.latent, generator=g)
    std_noise, eps_noise = torch.randn(4, generator=g), torch.randn(4, 4, cfg.ac_dim, generator=g)
    _, conn = O.process_speech_inputs(sd_o, cfg.as_dict(), wav, sm, std_noise, eps_noise)
    refs = [O.generate(sd_o, cfg.as_dict(), ids.tolist(), sp_mask, conn, _special(V), noise[b], cfg_scale=2.0, n_steps=20, forced_tokens=forced[b], bf16_t=True)
            for b in range(3)]
    m = VibeVoiceForConditionalGenerationInference(cfg, sd, device="cuda:0", torch_dtype=torch.bfloat16)
    m.set_ddpm_inference_steps(20)
    tok = _Tok(V)
    tok.pad_id = 0
    kw = dict(input_ids=ids[None].repeat(3, 1), attention_mask=torch.ones(3, ids.shape[0], dtype=torch.long), speech_tensors=wav.repeat(3, 1), speech_masks=sm.repeat(3, 1),
              speech_input_mask=sp_mask[None].repeat(3, 1), tokenizer=tok, cfg_scale=2.0, forced_tokens=forced, noise=noise,
              speech_noise=(std_noise.repeat(3), eps_noise.repeat(3, 1, 1)))
    outs = {rbm: m.generate(row_batch=rbm, **kw) for rbm in (True, False)}
    assert (3, 0) in m._rowbatch
    for b in range(3):
        want = torch.cat(refs[b].audio).numpy()
        assert outs[True].sequences[b, ids.shape[0]: ids.shape[0] + len(forced[b])].tolist() == forced[b]
        e_rows = rel_rms(outs[True].speech_outputs[b][0].float().cpu().numpy(), want, what=f"generate() 3 dialogues, mid bf16, ROW-BATCHED vs oracle, dialogue {b}")
        e_lane = rel_rms(outs[False].speech_outputs[b][0].float().cpu().numpy(), want, what=f"generate() 3 dialogues, mid bf16, lanes vs oracle, dialogue {b}")
        assert e_rows < 2e-2, f"row-batched dialogue {b}: waveform rel RMS {e_rows:.3e} vs oracle (lanes: {e_lane:.3e})"


@pytest.mark.parametrize("B", [6, 8, 9])
def test_generate_two_row_batches_vs_lanes(big, B):
    """5..16 dialogues: ceil(B / 4) row batches (3 + 3, 4 + 4, 3 + 3 + 3) inside one lock-step loop, both on the main stream, their conv tails on the three side
    streams (two lanes per stream, one launch worker per stream) - against the lanes: same sequences, waveforms to the bf16 noise floor."""
    cfg, sd, m = big
    tok = _Tok(cfg.vocab)
    D, E, EOS, S = tok.speech_diffusion_id, tok.speech_end_id, tok.eos_token_id, tok.speech_start_id
    g = torch.Generator().manual_seed(37 + B)
    L = 40
    ids = torch.stack([torch.cat([torch.randint(0, 1000, (L - 1,), generator=g), torch.tensor([S])]) for _ in range(B)])
    forced = [[D] * (3 + (b % 3)) + ([E, S, D, D] if b == 1 else []) + [E, EOS] for b in range(B)]
    noise = torch.randn(B, 8, cfg.latent, generator=g)
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=tok, cfg_scale=2.0, forced_tokens=forced, noise=noise)
    rows = m.generate(row_batch=True, **kw)
    lanes = m.generate(row_batch=False, **kw)
    n_groups = -(-B // 4)
    sizes = [B // n_groups + (1 if g < B % n_groups else 0) for g in range(n_groups)]
    assert all((n, sum(sizes[:g]), "side") in m._rowbatch for g, n in enumerate(sizes))      # ceil(B / 4) row batches, every conv tail beside the main stream
    assert rows.sequences.tolist() == lanes.sequences.tolist()
    for b in range(B):
        assert rows.speech_outputs[b].shape == lanes.speech_outputs[b].shape
        err = rel_rms(rows.speech_outputs[b].float().cpu().numpy(), lanes.speech_outputs[b].float().cpu().numpy(),
                      what=f"generate() on {B} dialogues (two row batches) 1.5B bf16 vs lanes, waveform of dialogue {b}")
        assert err < 1e-2, f"{B} dialogues, dialogue {b}: waveform rel RMS {err:.3e}"


def test_frag_flag_is_rejected_where_nothing_reads_that_layout(lib):
    """VV_LIN_W_FRAG on a call the 5..8-row GEMV does not take (2 rows; no split-K scratch for a long-K shape) is an error, never a silent read of
    a fragment-major matrix as if it were row-major."""
    from vibevoice_rocm_amd import _lib as L
    x = torch.randn(8, 8960, device="cuda")
    w = _frag((torch.randn(1536, 8960, device="cuda") / 90).bfloat16())
    out = torch.zeros(8, 1536, device="cuda")
    a = L.LinArgs()
    a.x, a.ldx, a.n, a.k, a.wdt = x.data_ptr(), 8960, 1536, 8960, L.VV_BF16
    a.w, a.flags, a.out, a.ldo = w.data_ptr(), L.LIN_W_FRAG, out.data_ptr(), 1536
    for m in (2, 8):      # 2 rows: never this kernel; 8 rows: no split-K scratch for the long-K shape
        a.m = m
        assert lib.vv_linear(C.byref(a), torch.cuda.current_stream().cuda_stream) != 0, f"m={m}"
        assert b"FRAG" in lib.vv_last_error()


def test_generate_row_batch_1p5b_bf16_vs_oracle(big):
    """The row-batched path at the benchmark's REAL shapes against the CPU oracle: VibeVoice-1.5B, bf16 weights (the oracle computes in fp32 on
    the same bf16-rounded matrices, bf16 timestep quirk on), ONE generate() call on 3 dialogues - a shared 2-frame voice prompt, a 40-token
    prompt, different schedules (a turn switch with a rolled-back speculative frame, an early end) and noise - each dialogue against its own
    oracle run at the bf16 bar (2e-2), the lanes' error next to it in the parity record."""
    from oracle import vv_oracle as O
    cfg, sd, m = big
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(53)
    ids = torch.cat([torch.randint(0, 1000, (39,), generator=g), torch.tensor([ST])])
    forced = [[SD, SD, SE, ST, SD, SE, EOS], [SD, SD, SD, SE, EOS], [SD, SE, EOS]]
    noise = torch.randn(3, 4, cfg.latent, generator=g)
    voice = 0.1 * torch.randn(1, 2 * cfg.hop - 321, generator=g)
    sp_mask = torch.zeros(40, dtype=torch.bool)
    sp_mask[7:9] = True
    speech_masks = torch.ones(1, 2, dtype=torch.bool)
    std_noise, eps_noise = torch.randn(1, generator=g), torch.randn(1, 2, cfg.ac_dim, generator=g)
    tok = _Tok(V)
    tok.pad_id = 0
    kw = dict(input_ids=ids[None].repeat(3, 1), attention_mask=torch.ones(3, 40, dtype=torch.long), speech_tensors=voice.repeat(3, 1),
              speech_masks=speech_masks.repeat(3, 1), speech_input_mask=sp_mask[None].repeat(3, 1), tokenizer=tok, cfg_scale=2.0, forced_tokens=forced,
              noise=noise, speech_noise=(std_noise.repeat(3), eps_noise.repeat(3, 1, 1)))
    outs = {rbm: m.generate(row_batch=rbm, **kw) for rbm in (True, False)}
    torch.set_num_threads(16)
    sd_o = {k: v.float().cpu() for k, v in sd.items()}
    ocfg = cfg.as_dict()
    _, conn = O.process_speech_inputs(sd_o, ocfg, voice, speech_masks, std_noise, eps_noise)
    special = dict(speech_start=ST, speech_end=SE, speech_diffusion=SD, eos=EOS)
    for b in range(3):
        ref = O.generate(sd_o, ocfg, ids.tolist(), sp_mask, conn, special, noise[b], cfg_scale=2.0, n_steps=20, forced_tokens=forced[b], bf16_t=True)
        want = torch.cat(ref.audio).numpy()
        got = outs[True].speech_outputs[b][0].float().cpu().numpy()
        assert got.shape == want.shape
        e_rows = rel_rms(got, want, what=f"generate() 3 dialogues, 1.5B bf16, ROW-BATCHED vs oracle, dialogue {b}")
        e_lane = rel_rms(outs[False].speech_outputs[b][0].float().cpu().numpy(), want, what=f"generate() 3 dialogues, 1.5B bf16, lanes vs oracle, dialogue {b}")
        assert e_rows < 2e-2, f"row-batched dialogue {b} at 1.5B: waveform rel RMS {e_rows:.3e} vs oracle (lanes: {e_lane:.3e})"
