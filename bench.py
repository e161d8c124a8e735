#!/usr/bin/env python3
"""bench.py — audio-seconds generated per wall-second on MI355X for the VibeVoice hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1], SURVEY.md §8d cfg 2): VibeVoice-1.5B shapes, 1 speaker, 27 s synthetic voice
prompt (203 frames), prompt of ~330 tokens, forced schedule of 225 speech_diffusion frames (30 s of 24 kHz audio)
then speech_end, eos; CFG = 2.0, 20 DPM-Solver++ steps, bf16 weights / fp32 accumulate, random-init weights.
A "step" is one whole generate() of that script (voice-prompt encode + LLM prefill + 225 frames): exactly what the
reference's demo times for its RTF (demo/inference_from_file.py:383-405).  Inputs are resident in HBM before the
timed region.  With N GPUs every rank synthesises its own dialogue (weak scaling, no data-path collective);
weights are broadcast from rank 0 and waveforms gathered to rank 0 over RCCL.

The JSON line also carries
  roofline      the dominant kernel (the streamed-weight GEMV) timed with HIP events on its launch stream,
                algorithmic bytes = its weight matrix, against the 8 TB/s HBM3E peak;
  frame         algorithmic bytes per frame (SURVEY.md §8d formula) / measured seconds per frame;
  cpu_baseline  the CPU oracle (oracle/vv_oracle.py, kind "port") timed on this host on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: one hardware queue per lane of the batched / concurrent legs

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


WORKLOADS = {
    # BASELINE.json configs[1..4] restated synthetically: model preset, speakers, voice frames per speaker, text
    # tokens, frames, frames per turn (speech_end, speech_start between turns), solver steps, weight quantisation, prefill chunk
    "cfg2": dict(model="1.5b", speakers=1, voice_frames=203, text=88, frames=225, turn=0, steps=20, quant=None, chunk=1024),
    "cfg3": dict(model="1.5b", speakers=4, voice_frames=203, text=168, frames=450, turn=75, steps=20, quant=None, chunk=1024),
    "cfg4": dict(model="7b", speakers=2, voice_frames=203, text=1042, frames=2250, turn=75, steps=20, quant=None, chunk=1024),
    "cfg5": dict(model="7b", speakers=2, voice_frames=203, text=1042, frames=2250, turn=75, steps=50, quant="fp8", chunk=512),
}


def build_workload(cfg, frames, voice_frames, seed=1, speakers=1, text=88, turn=0):
    """Synthetic processor output: ids uniform in [0,1000), `speakers` voice prompts laid out as the processor does
    (" Speaker i:" prefix, speech_start, placeholders, speech_end, newline), forced schedule of `frames` speech_diffusion tokens with
    `speech_end, speech_start` every `turn` frames (0: one turn), then speech_end, eos."""
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(seed)
    lim = min(1000, V - 8)
    parts, masks = [torch.randint(0, lim, (30,), generator=g)], [torch.zeros(30, dtype=torch.bool)]
    for _ in range(speakers):
        pre = torch.randint(0, lim, (6,), generator=g)
        parts += [pre, torch.tensor([ST]), torch.full((voice_frames,), SD), torch.tensor([SE])]
        masks += [torch.zeros(7, dtype=torch.bool), torch.ones(voice_frames, dtype=torch.bool), torch.zeros(1, dtype=torch.bool)]
    parts += [torch.randint(0, lim, (text,), generator=g), torch.tensor([ST])]
    masks += [torch.zeros(text + 1, dtype=torch.bool)]
    ids, mask = torch.cat(parts), torch.cat(masks)
    gv = torch.Generator().manual_seed(seed + 1)
    voice = torch.randn(speakers, voice_frames * cfg.hop - 1234, generator=gv)
    voice = voice * (10 ** (-25 / 20) / (voice.pow(2).mean(-1, keepdim=True).sqrt() + 1e-6))          # -25 dBFS like AudioNormalizer
    forced = []
    for f in range(frames):
        if turn and f and f % turn == 0:
            forced += [SE, ST]
        forced.append(SD)
    forced += [SE, EOS]
    gn = torch.Generator().manual_seed(seed + 2)
    noise = torch.randn(frames, cfg.latent, generator=gn)
    gs = torch.Generator().manual_seed(seed + 3)
    speech_noise = (torch.randn(speakers, generator=gs), torch.randn(speakers, voice_frames, cfg.ac_dim, generator=gs))

    class Tok:
        speech_start_id, speech_end_id, speech_diffusion_id, eos_token_id, bos_token_id, pad_id = ST, SE, SD, EOS, None, 0

    return dict(input_ids=ids[None], speech_input_mask=mask[None], speech_tensors=voice,
                speech_masks=torch.ones(speakers, voice_frames, dtype=torch.bool), forced=forced, noise=noise,
                speech_noise=speech_noise, tok=Tok(), special=dict(speech_start=ST, speech_end=SE, speech_diffusion=SD, eos=EOS))


def bytes_per_frame(cfg, n_steps, mean_ctx, wbytes=2, gemv_wbytes=None):
    """b*(W_llm + N*W_head + W_dec + W_sem + W_conn) + 2*S*KVb + state.  gemv_wbytes: bytes per weight of the LLM and head
    matrices when they stream as fp8 codes (weight_quant="fp8"; the adaLN matrices and norms stay bf16, counted at 1 byte here too: an
    under-count of < 15 % of the head)."""
    from vibevoice_rocm_amd.synth import state_dict_shapes
    tot = dict(llm=0, head=0, dec=0, sem=0, conn=0)
    for n, s in state_dict_shapes(cfg).items():
        k = int(np.prod(s)) if len(s) else 1
        if n.startswith("model.language_model.layers.") or n == "model.language_model.norm.weight":
            tot["llm"] += k
        elif n.startswith("model.prediction_head."):
            tot["head"] += k
        elif n.startswith("model.acoustic_tokenizer.decoder."):
            tot["dec"] += k
        elif n.startswith("model.semantic_tokenizer.encoder."):
            tot["sem"] += k
        elif "_connector." in n:
            tot["conn"] += k
    kvb = cfg.layers * 2 * cfg.kv_heads * cfg.head_dim * wbytes
    gw = gemv_wbytes or wbytes
    b = gw * (tot["llm"] + n_steps * tot["head"]) + wbytes * (tot["dec"] + tot["sem"] + tot["conn"]) + 2 * mean_ctx * kvb
    b_resident_head = gw * (tot["llm"] + tot["head"]) + wbytes * (tot["dec"] + tot["sem"] + tot["conn"]) + 2 * mean_ctx * kvb
    return b, b_resident_head, tot


def run_generate(model, wl, cfg_scale, n_frames=None, use_voice=True):
    forced = wl["forced"] if n_frames is None else [wl["special"]["speech_diffusion"]] * n_frames + wl["forced"][-2:]
    # generate() caps a dialogue at 2x the prompt length (max_length_times, reference :420): the synthetic prompts are shorter than
    # a real script of that duration, so the cap is lifted to fit the forced schedule
    mlt = max(2, -(-len(forced) // wl["input_ids"].shape[1]) + 1)
    kw = dict(input_ids=wl["input_ids"], tokenizer=wl["tok"], cfg_scale=cfg_scale, forced_tokens=forced, noise=wl["noise"],
              generation_config={"do_sample": False}, show_progress_bar=False, max_length_times=mlt)
    if use_voice:
        kw.update(speech_tensors=wl["speech_tensors"], speech_masks=wl["speech_masks"], speech_input_mask=wl["speech_input_mask"],
                  speech_noise=wl["speech_noise"])
    return model.generate(**kw)


def roofline_leg(model, wl, cfg_scale, frames=12):
    """Eager (no graph) pass with HIP-event timing around every vv_linear launch; returns the dominant kernel's stats."""
    from vibevoice_rocm_amd import _lib as L
    lib = L.load()
    eng = model.engine
    was = eng.use_graphs
    eng.use_graphs = False
    frames = min(frames, wl["noise"].shape[0])
    try:
        run_generate(model, wl, cfg_scale, n_frames=2, use_voice=False)            # warm
        L.check(lib.vv_prof_begin(400000), "vv_prof_begin")
        run_generate(model, wl, cfg_scale, n_frames=frames, use_voice=False)
        out = (L.ProfEntry * 256)()
        n = C.c_int(0)
        L.check(lib.vv_prof_end(out, 256, C.byref(n)), "vv_prof_end")
    finally:
        eng.use_graphs = was
    ents = []
    for i in range(n.value):
        e = out[i]
        if e.m > 8:
            continue                                                                 # prefill GEMMs are not the per-frame path
        wb = {L.VV_BF16: 2, L.VV_FP8: 1}.get(e.wdt, 4) * e.n * e.k * (2 if e.dual else 1)
        ents.append(dict(m=e.m, n=e.n, k=e.k, dual=bool(e.dual), count=e.count, total_ms=e.total_ms, avg_us=1e3 * e.total_ms / e.count,
                         weight_bytes=wb, gbs=wb / (e.total_ms / e.count * 1e-3) / 1e9))
    ents.sort(key=lambda d: -d["total_ms"])
    return ents


def kernel_source_sha():
    """Content hash of the kernel sources + C ABI header: ties a PMC traffic file to the code it was measured on (the GPU box has no
    .git, and the driver's bench box neither)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vibevoice_rocm_amd", "csrc")
    for fn in sorted(os.listdir(d)) + ["../../include/vv_hip.h"]:
        path = os.path.join(d, fn)
        if os.path.isfile(path) and fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(shape_key):
    """HBM-side bytes per launch of a shape from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE inside graph replays of real
    frames, x2 gfx950 correction; tools/pmc_frames.py) - only when that file was measured on THIS kernel source, else None."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tp) as f:
            j = json.load(f)
        if j.get("kernel_src_sha") != kernel_source_sha():
            return None, f"profiles/pmc_traffic.json was measured on kernel source {j.get('kernel_src_sha')}, this tree is {kernel_source_sha()}"
        return j.get("per_launch_bytes", {}).get(shape_key), j.get("note")
    except Exception as e:      # noqa: BLE001
        return None, repr(e)


def in_graph_durations():
    """Per-shape in-graph kernel durations from profiles/r03_per_shape.csv (rocprofv3 --kernel-trace over hipGraph replays of real generate()
    frames, tools/frame_trace.py) - only when profiles/r03_per_shape.sha names THIS kernel source, else ({}, reason)."""
    import csv
    base = os.path.join(ROOT, "profiles", "r03_per_shape")
    try:
        sha = open(base + ".sha").read().strip()
        if sha != kernel_source_sha():
            return {}, f"profiles/r03_per_shape.csv was measured on kernel source {sha}, this tree is {kernel_source_sha()}"
        out = {}
        for r in csv.DictReader(open(base + ".csv")):
            if int(r["m"] or 0) > 0:
                out[(int(r["m"]), int(r["n"]), int(r["k"]))] = float(r["avg_us"])
        return out, "profiles/r03_per_shape.csv (rocprofv3 --kernel-trace, in-graph replays of real frames)"
    except Exception as e:      # noqa: BLE001
        return {}, repr(e)


def llm_prefill_flops(cfg, L0):
    """2 x M x N x K over the Qwen2 linears of an L0-row prefill + causal attention (QK^T and PV, half the square)."""
    H, I, qd, kvd = cfg.hidden, cfg.inter, cfg.q_dim, cfg.kv_dim
    lin = 2.0 * L0 * (H * (qd + 2 * kvd) + qd * H + 3 * H * I)
    att = 2.0 * 2.0 * cfg.heads * cfg.head_dim * L0 * (L0 + 1) / 2
    return cfg.layers * (lin + att)


def encoder_flops(cfg, T):
    """Dense conv + Block1D FFN flops of a whole-utterance acoustic encode of T samples (depthwise taps and norms ignored)."""
    rr = list(reversed(cfg.ac_ratios))
    fl, t, c_in = 0.0, T, 1
    for i, depth in enumerate(cfg.ac_depths):
        c = cfg.ac_filters * 2 ** i
        k, s = (7, 1) if i == 0 else (2 * rr[i - 1], rr[i - 1])
        t = -(-t // s)
        fl += 2.0 * t * c * k * c_in
        fl += depth * 2.0 * t * (2 * 4 * c * c)
        c_in = c
    return fl + 2.0 * t * cfg.ac_dim * 7 * c_in


def first_chunk_parts(model, wl, cfg, runs=5):
    """The two matrix-core legs of the first-chunk latency timed with HIP events on the engine stream, against the dense bf16 MFMA
    peak (2.5 PFLOP/s, MI355X_MICROARCH.md): voice-prompt encode (acoustic encoder over the whole prompt + connector) and LLM prefill."""
    eng = model.engine
    dev = eng.device
    vt = wl["speech_tensors"].to(dev).float()
    L0 = wl["input_ids"].shape[1]
    x0 = torch.randn(L0, cfg.hidden, device=dev) * 0.02
    V = cfg.vocab
    eng.begin_sequence(L0 + 64, [V - 4, V - 3, V - 2, V - 1])

    def timed(fn):
        ts = []
        for _ in range(runs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eng.stream):
                e0.record(eng.stream)
                fn()
                e1.record(eng.stream)
            eng.stream.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]
    ms_v = timed(lambda: model._process_speech_inputs(vt, wl["speech_masks"], *wl["speech_noise"]))
    ms_p = timed(lambda: eng.prefill(x0, row=0, pos0=0))
    fv = sum(encoder_flops(cfg, int(vt.shape[1])) for _ in range(vt.shape[0]))
    fp = llm_prefill_flops(cfg, L0)
    peak = 2500.0
    return {"voice_encode": {"ms": round(ms_v, 3), "gflop": round(fv / 1e9, 1), "tflops": round(fv / ms_v / 1e9, 1), "frac_of_mfma_peak": round(fv / ms_v / 1e9 / peak, 4)},
            "llm_prefill": {"ms": round(ms_p, 3), "rows": L0, "gflop": round(fp / 1e9, 1), "tflops": round(fp / ms_p / 1e9, 1), "frac_of_mfma_peak": round(fp / ms_p / 1e9 / peak, 4)},
            "mfma_peak_tflops": peak, "note": "HIP events on the engine stream, median of %d; dense bf16 MFMA peak 2.5 PFLOP/s" % runs}


def first_chunk_leg(model, wl, cfg_scale, runs=5, gen_kw=None):
    """p50 latency from generate() entry to the first 3200-sample chunk on the host (voice encode + prefill + 1 frame),
    through the AudioStreamer path the reference's streaming callers use."""
    from vibevoice_rocm_amd.streamer import AudioStreamer

    class Timer(AudioStreamer):
        def __init__(self):
            super().__init__(batch_size=1)
            self.t_first = None

        def put(self, audio_chunks, sample_indices):
            audio_chunks[0].detach().cpu()                     # the chunk is on the host, as a consumer would have it
            if self.t_first is None:
                self.t_first = time.perf_counter()

    lat = []
    for _ in range(runs):
        st = Timer()
        forced = [wl["special"]["speech_diffusion"]] * 2 + wl["forced"][-2:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(input_ids=wl["input_ids"], tokenizer=wl["tok"], cfg_scale=cfg_scale, forced_tokens=forced, noise=wl["noise"],
                       speech_tensors=wl["speech_tensors"], speech_masks=wl["speech_masks"], speech_input_mask=wl["speech_input_mask"],
                       speech_noise=wl["speech_noise"], audio_streamer=st, generation_config={"do_sample": False}, **(gen_kw or {}))
        lat.append(1e3 * (st.t_first - t0))
    lat.sort()
    return dict(p50_ms=round(lat[len(lat) // 2], 2), min_ms=round(lat[0], 2), max_ms=round(lat[-1], 2), runs=runs,
                note="generate() entry -> first 3200-sample chunk on host: voice-prompt encode + LLM prefill + first frame")


def concurrent_leg(model, cfg, sd, dtype, device, args, streams=3):
    """Serving-throughput extra (NOT the headline value): `streams` independent dialogues on ONE GPU, each with its own Engine,
    HIP stream and host thread, sharing the resident weights.  A single dialogue is a latency-bound chain of ~580 small
    kernels per frame, so a second and third chain fill the bubbles."""
    import gc
    import threading
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    model.release_lanes()                    # lanes / row batches of an earlier batched leg hand their HIP streams back (engine.py, _IDLE_STREAMS)
    gc.collect()
    models = [model] + [VibeVoiceForConditionalGenerationInference(cfg, sd, device=device, torch_dtype=dtype, use_graphs=not args.no_graphs)
                        for _ in range(streams - 1)]
    for mm in models:
        mm.set_ddpm_inference_steps(args.ddpm_steps)
    wls = [build_workload(cfg, args.frames, args.voice_frames, seed=101 + i) for i in range(streams)]
    outs = [0] * streams

    def run(i):
        outs[i] = run_generate(models[i], wls[i], args.cfg_scale).speech_outputs[0].shape[-1]

    dt = None
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(i,)) for i in range(streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    del models, mm
    gc.collect()               # the extra engines hand their HIP streams back for the next leg (engine.py, _IDLE_STREAMS)
    return dict(streams=streams, value=round(sum(outs) / 24000.0 / dt, 3), unit="audio-sec/s", seconds=round(dt, 3),
                note="aggregate over independent dialogues run concurrently on one GPU (one Engine/stream/thread each, shared weights); "
                     "not the headline metric, which is one dialogue per GPU")


def batched_leg(model, cfg, args, batch=4, row_batch=None):
    """Extra (NOT the headline value): ONE generate() call on a batch of `batch` dialogues of the headline shape - the reference's own
    batch dimension (modeling_vibevoice_inference.py:459-653).  The samples advance in lock step under one host loop: 3-4 dialogues batched
    into the row dimension of the LLM / diffusion-head weight passes (rowbatch.py), otherwise (or row_batch=False) on one engine lane each
    (own HIP stream, KV cache and conv state, shared weights)."""
    wls = [build_workload(cfg, args.frames, args.voice_frames, seed=201 + i) for i in range(batch)]
    ids = torch.cat([w["input_ids"] for w in wls])
    kw = dict(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=wls[0]["tok"], cfg_scale=args.cfg_scale,
              forced_tokens=[w["forced"] for w in wls], noise=torch.stack([w["noise"] for w in wls]),
              speech_tensors=torch.cat([w["speech_tensors"] for w in wls]).to(model.engine.device),
              speech_masks=torch.cat([w["speech_masks"] for w in wls]), speech_input_mask=torch.cat([w["speech_input_mask"] for w in wls]),
              speech_noise=(torch.cat([w["speech_noise"][0] for w in wls]), torch.cat([w["speech_noise"][1] for w in wls])),
              generation_config={"do_sample": False}, show_progress_bar=False,
              max_length_times=max(2, -(-len(wls[0]["forced"]) // ids.shape[1]) + 1))
    if row_batch is not None:
        kw["row_batch"] = row_batch
    n, dts = 0, []
    for timed in (False, False, True, True, True):   # two warm-up calls (lanes / graphs; the allocator's per-stream pools), then three timed ones:
        torch.cuda.synchronize()                     # the loop enqueues ~1000 graph nodes per step from a few host threads and a call's rate
        t0 = time.perf_counter()                     # moves by ~10 % with how they interleave - the median is reported, every run listed
        out = model.generate(**kw)
        torch.cuda.synchronize()
        if timed:
            dts.append(time.perf_counter() - t0)
        n = sum(o.shape[-1] for o in out.speech_outputs)
    assert n == batch * args.frames * cfg.hop, (n, batch, args.frames)
    dt = sorted(dts)[len(dts) // 2]
    rowb = (model.row_batch if row_batch is None else row_batch) and 2 <= batch <= 16
    return dict(batch=batch, value=round(n / 24000.0 / dt, 3), unit="audio-sec/s", seconds=round(dt, 3), row_batched=bool(rowb),
                runs=[round(n / 24000.0 / d, 2) for d in dts],
                note="one generate() call on a batch of dialogues of the headline shape, in lock step; " +
                     ("the dialogues are batched into the row dimension of the LLM and diffusion-head weight passes (one pass per frame for all of them, "
                      "vibevoice_rocm_amd/rowbatch.py), conv tokenizers per dialogue; lanes_value = the same call with one engine lane per dialogue; "
                      if rowb else "one engine lane per dialogue; ") +
                     "aggregate audio seconds per wall second, not the headline metric (one dialogue per GPU)")


def fp8_leg(cfg, sd, device, args, wl):
    """Extra (NOT the headline value, which is bf16): the same workload with weight-only fp8 (e4m3fn codes + power-of-two row scales)
    on the per-frame weight-streaming GEMVs - LLM linears, head SwiGLU matrices, the 1-row conv stage."""
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    m8 = VibeVoiceForConditionalGenerationInference(cfg, sd, device=device, torch_dtype=torch.bfloat16, use_graphs=not args.no_graphs,
                                                    weight_quant="fp8")
    m8.set_ddpm_inference_steps(args.ddpm_steps)
    n = 0
    for timed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_generate(m8, wl, args.cfg_scale).speech_outputs[0].shape[-1]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dict(value=round(n / 24000.0 / dt, 3), unit="audio-sec/s", seconds=round(dt, 3), weights="e4m3fn + power-of-two row scales",
                note="same workload, weight-only fp8 on the decode GEMVs (bf16 activations, fp32 accumulate); an optional mode "
                     "(weight_quant='fp8'), not the headline metric")


def cpu_baseline_leg(sd_dev, cfg, frames=4, timed_runs=3):
    """BASELINE.md section 3 / SURVEY.md section 8d cfg 1 restated on the CPU oracle (kind "port"): VibeVoice-1.5B shapes, fp32, CFG = 1.0,
    10 solver steps, 1 speaker with a 70-frame voice prompt through process_speech_inputs, a ~170-token prompt, forced speech_diffusion
    frames.  One warm-up + `timed_runs` timed passes of the frame loop (median), per-component seconds per frame; the once-per-utterance
    legs (voice encode, prefill) are timed once.  Bounded to a few frames so that the default bench run stays within minutes: the
    per-frame cost does not depend on how many frames follow (KV growth over 4 frames is negligible)."""
    from oracle import vv_oracle as O
    cores = min(len(os.sched_getaffinity(0)), 16)       # a 1-GPU box's CPU share is 16 cores even when the mask lists the whole host
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads, copying weights to host (fp32)")
    sd = {k: v.detach().float().cpu() for k, v in sd_dev.items()}
    V = cfg.vocab
    special = dict(speech_start=V - 4, speech_end=V - 3, speech_diffusion=V - 2, eos=V - 1)
    wl = build_workload(cfg, frames, 70, seed=7, speakers=1, text=57)           # 30 + 6 + 1 + 70 + 1 + 57 + 1 = 166 prompt tokens
    ocfg = cfg.as_dict()
    ids = wl["input_ids"][0].tolist()
    t0 = time.time()
    _, conn = O.process_speech_inputs(sd, ocfg, wl["speech_tensors"], wl["speech_masks"], *wl["speech_noise"])
    t_voice = time.time() - t0
    # per-component timers around the oracle's own functions
    acc = {}
    names = dict(llm_forward="llm", sample_speech_tokens="head", tokenizer_decoder="decode", semantic_encode="semantic", connector="connectors")
    orig = {n: getattr(O, n) for n in names}

    def wrap(n):
        f = orig[n]

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            acc[names[n]] = acc.get(names[n], 0.0) + time.perf_counter() - t
            return r
        return g
    for n in names:
        setattr(O, n, wrap(n))
    runs = []
    try:
        for i in range(1 + timed_runs):
            acc.clear()
            t0 = time.time()
            res = O.generate(sd, ocfg, ids, wl["speech_input_mask"][0], conn, special, wl["noise"], cfg_scale=1.0, n_steps=10, forced_tokens=wl["forced"])
            dt = time.time() - t0
            runs.append((dt, dict(acc)))
            log(f"cpu baseline pass {i}: {dt:.1f} s")
    finally:
        for n in names:
            setattr(O, n, orig[n])
    timed = sorted(runs[1:], key=lambda r: r[0])
    dt, comp = timed[len(timed) // 2]
    n_fr = len(res.audio)
    # the first llm_forward call of a pass is the prompt prefill: measured apart from the per-frame steps
    k0 = O.KVCache(cfg.layers)
    t0 = time.perf_counter()
    O.llm_forward(sd, ocfg, sd["model.language_model.embed_tokens.weight"][torch.tensor(ids)], k0, 0)
    t_prefill = time.perf_counter() - t0
    per_frame = {k: round((v - (t_prefill if k == "llm" else 0.0)) / n_fr, 4) for k, v in comp.items()}
    frame_s = (dt - t_prefill) / n_fr
    return dict(value=round((cfg.hop / 24000.0) / frame_s, 4), unit="audio-sec/s", cores=cores, kind="port",
                seconds_per_frame=round(frame_s, 4), per_frame_seconds=per_frame, prefill_seconds=round(t_prefill, 2), voice_encode_seconds=round(t_voice, 2),
                sample=f"BASELINE.md section 3 (cfg 1 restated): oracle/vv_oracle.py on 1.5B shapes, fp32, CFG=1.0, 10 DPM-Solver++ steps, {len(ids)}-token prompt with a "
                       f"70-frame voice prompt, {n_fr} forced frames per pass, 1 warm-up + {timed_runs} timed passes (median {dt:.1f} s incl. {t_prefill:.1f} s prefill); "
                       "value = steady-state frames only (per-frame LLM pos+neg, head, decode, semantic, connectors)")


_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, timeout=None):
    """Parent of a self-launched N-rank run (no torchrun): one fresh child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rank 0's stdout relayed to ours, every rank's stderr inherited.  The parent never touches the GPU (no exec from a process that
    has).  Returns the exit code: 0 only if every rank returned 0; the first failure ends the other ranks (exact PIDs)."""
    import signal
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []

    def reap(*_):            # the parent was told to stop (driver timeout, Ctrl-C): the ranks must not outlive it holding the GPUs
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except Exception:      # noqa: BLE001
                p.kill()
        raise SystemExit(143)
    signal.signal(signal.SIGTERM, reap)
    signal.signal(signal.SIGINT, reap)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_WORLD_SIZE=str(n))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    log(f"launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    import threading
    lines = []

    def relay():
        for ln in procs[0].stdout:
            lines.append(ln)          # held back until every rank has returned 0: a failed run prints no result line
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    t0, rc = time.time(), 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0:
                log(f"launcher: rank {r} exited with {c}")
                rc = rc or (c if c > 0 else 1)
        if rc or (timeout and time.time() - t0 > timeout):
            rc = rc or 124
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(10)
                except Exception:      # noqa: BLE001
                    procs[r].kill()
            break
        time.sleep(0.05)
    th.join(5)
    if rc == 0 and not any(ln.strip().startswith(b"{") for ln in lines):
        log("launcher: rank 0 printed no JSON line")
        rc = 1
    for ln in lines:          # stdout carries the result line only; library chatter on rank 0's stdout (gloo's connection notes) goes to stderr
        out = sys.stdout if rc == 0 and ln.lstrip().startswith(b"{") else sys.stderr
        out.write(ln.decode(errors="replace"))
        out.flush()
    return rc


_RESULT_OUT = None


def claim_stdout():
    """fd 1 carries the result line and nothing else: native code of the collectives libraries writes to stdout (RCCL prints a five-line
    version banner at init, gloo its connection notes), so a rank process parks the real stdout, points fd 1 at stderr for everybody
    else, and emit_result() writes the one JSON line to the parked descriptor."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_result(obj):
    out = _RESULT_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def init_ranks(gpus):
    """RANK / LOCAL_RANK / WORLD_SIZE from the environment (set by torch.distributed.run or by launch_ranks).  WORLD_SIZE must equal
    --gpus in every case.  Returns (rank, local_rank, world, dist module or None, backend name or None)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (self-launching) or with "
                         f"torch.distributed.run --nproc-per-node N")
    if world == 1 and os.environ.get("VV_DIST_FORCE") != "1":
        return rank, local_rank, world, None, None
    # VV_DIST_FORCE=1: a one-rank process group, so that the RCCL branch (broadcast, gather, max-reduce, barrier) runs end to end on a 1-GPU box
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    backend = os.environ.get("VV_DIST_BACKEND", "nccl")          # "gloo" lets ranks rehearse on one GPU / on the CPU
    if os.environ.get("VV_ALL_RANKS_ONE_GPU") == "1":
        local_rank = 0
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert dist.get_world_size() == world and dist.get_rank() == rank
    return rank, local_rank, world, dist, backend


def stub_body(args, rank, world, dist, backend):
    """VV_BENCH_STUB=1: the launcher / rendezvous / timing skeleton with no GPU work (CPU test of the N-rank path)."""
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if dist is not None:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if os.environ.get("VV_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    if rank == 0:
        emit_result({"metric": "stub", "n_gpus": world, "max_rank_plus_1": float(t.item()),
                     "rccl": {"ranks": dist.get_world_size() if dist is not None else 1, "backend": backend}})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="BASELINE.json configs[1..4]; cfg2 (1.5B, 1 speaker, 30 s) is the headline the metric is quoted on")
    ap.add_argument("--model", default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--voice-frames", type=int, default=None)
    ap.add_argument("--cfg-scale", type=float, default=2.0)
    ap.add_argument("--ddpm-steps", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-fp8-leg", dest="fp8_leg", action="store_false", help="skip the extra weight-only-fp8 leg")
    ap.add_argument("--concurrent", type=int, default=3, help="extra leg: N independent dialogues concurrently on one GPU (0/1 = skip)")
    ap.add_argument("--batched", type=int, default=4, help="extra leg: one generate() call on a batch of N dialogues (0/1 = skip)")
    ap.add_argument("--first-chunk-runs", type=int, default=5)
    args = ap.parse_args()
    W = WORKLOADS[args.workload]
    args.model = args.model or W["model"]
    args.frames = args.frames or W["frames"]
    args.voice_frames = args.voice_frames or W["voice_frames"]
    args.ddpm_steps = args.ddpm_steps or W["steps"]
    headline = args.workload == "cfg2"

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process has made no GPU call yet (argparse only) and never will - it starts N
        # fresh rank processes, relays rank 0's JSON line and leaves with the worst exit code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    claim_stdout()
    rank, local_rank, world, dist, backend = init_ranks(args.gpus)
    if os.environ.get("VV_BENCH_STUB") == "1":
        return stub_body(args, rank, world, dist, backend)
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    from vibevoice_rocm_amd.config import VVConfig
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    from vibevoice_rocm_amd import distributed as vd

    cfg = VVConfig.preset(args.model)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # rank 0 owns the (random-init) checkpoint; replicas receive it over RCCL/xGMI
    sd = synth_state_dict_torch(cfg, 1234, device=device, dtype=dtype) if rank == 0 else None
    rccl = None
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        sd = vd.broadcast_state_dict(sd, cfg, dtype, device, src=0)
        torch.cuda.synchronize()
        nb = sum(v.numel() * v.element_size() for v in sd.values())
        rccl = {"ranks": dist.get_world_size(), "backend": dist.get_backend(), "bcast": os.environ.get("VV_BCAST", "broadcast"),
                "bcast_s": round(time.perf_counter() - tb, 4), "bcast_GB": round(nb / 1e9, 3), "gather_ms": []}
    log("weights ready")
    model = VibeVoiceForConditionalGenerationInference(cfg, sd, device=device, torch_dtype=dtype, use_graphs=not args.no_graphs,
                                                       weight_quant=W["quant"] if dtype == torch.bfloat16 else None)
    log(f"engine ready ({model.engine.w.nbytes() / 1e9:.2f} GB resident)")
    single = world == 1            # the diagnostic legs run at N = 1 only (other ranks would just wait at the final barrier)
    keep_sd = rank == 0 and single and headline and (not args.no_cpu_baseline or args.concurrent > 1 or args.fp8_leg)
    if not keep_sd:
        del sd
        sd = None
    model.set_ddpm_inference_steps(args.ddpm_steps)
    wl = build_workload(cfg, args.frames, args.voice_frames, seed=1 + rank, speakers=W["speakers"], text=W["text"], turn=W["turn"])     # every rank its own dialogue
    wl["speech_tensors"] = wl["speech_tensors"].to(device)
    gen_kw = dict(prefill_chunk=W["chunk"])

    def gen():
        forced = wl["forced"]
        mlt = max(2, -(-len(forced) // wl["input_ids"].shape[1]) + 1)
        return model.generate(input_ids=wl["input_ids"], tokenizer=wl["tok"], cfg_scale=args.cfg_scale, forced_tokens=forced, noise=wl["noise"],
                              generation_config={"do_sample": False}, show_progress_bar=False, max_length_times=mlt,
                              speech_tensors=wl["speech_tensors"], speech_masks=wl["speech_masks"], speech_input_mask=wl["speech_input_mask"],
                              speech_noise=wl["speech_noise"], **gen_kw)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for i in range(args.warmup):
        tw = time.perf_counter()
        out = gen()
        torch.cuda.synchronize()
        log(f"warmup {i}: {time.perf_counter() - tw:.2f} s")
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = gen()
        if dist is not None:
            tg = time.perf_counter()
            vd.gather_waveforms(out.speech_outputs[0], dst=0)
            rccl["gather_ms"].append(round(1e3 * (time.perf_counter() - tg), 3))        # includes waiting for the slowest rank's dialogue
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_samples = out.speech_outputs[0].shape[-1]
    audio_s = n_samples / 24000.0
    assert n_samples == args.frames * cfg.hop, (n_samples, args.frames)
    assert bool(torch.isfinite(out.speech_outputs[0]).all())
    value = world * args.steps * audio_s / dt
    log(f"timed: {args.steps} steps in {dt:.2f} s -> {value:.2f} audio-sec/s")

    result = None
    if rank == 0:
        L0 = wl["input_ids"].shape[1]
        mean_ctx = L0 + len(wl["forced"]) / 2
        wb = 2 if dtype == torch.bfloat16 else 4
        bpf, bpf_res, _ = bytes_per_frame(cfg, args.ddpm_steps, mean_ctx, wb, 1 if (W["quant"] and dtype == torch.bfloat16) else None)
        s_per_frame = dt / (args.steps * args.frames)
        result = {
            "metric": "audio-sec/s", "value": round(value, 4), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype if not W["quant"] else f"{args.dtype} activations / fp8 e4m3 decode weights",
            "data": "synthetic (random-init weights, seeded prompt/voice/noise, forced token schedule)",
            "config": {"workload": f"{args.workload}: VibeVoice-{args.model.upper()} shapes, {W['speakers']} speaker(s) x {args.voice_frames}-frame voice prompt, "
                                   f"{L0}-token prompt, {args.frames} frames ({audio_s:.1f} s audio)"
                                   + (f", speech_end/speech_start every {W['turn']} frames" if W["turn"] else "") +
                                   f", CFG={args.cfg_scale}, {args.ddpm_steps} DPM-Solver++ steps, whole generate() incl. voice encode + prefill"
                                   + (f" in {W['chunk']}-token chunks" if W["chunk"] != 1024 else ""),
                       "global_batch": world, "parallelism": f"replicas x{world} (one dialogue per GPU)", "hipgraph": not args.no_graphs},
            "frame": {"ms_per_frame_incl_prefill": round(1e3 * s_per_frame, 4), "algorithmic_GB_per_frame": round(bpf / 1e9, 4),
                      "achieved_GBps": round(bpf / s_per_frame / 1e9, 1), "frac_of_hbm_peak": round(bpf / s_per_frame / 1e9 / HBM_PEAK_GBS, 4),
                      "frac_if_head_weights_counted_once": round(bpf_res / s_per_frame / 1e9 / HBM_PEAK_GBS, 4),
                      "note": "whole-frame HBM figure: the diffusion head's weights stay in the 256 MB Infinity Cache across solver steps, so "
                              "frac_if_head_weights_counted_once is the honest HBM utilisation"},
        }
    if rank == 0 and rccl is not None:
        rccl["bcast_GBps"] = round(rccl["bcast_GB"] / max(rccl["bcast_s"], 1e-9), 1)
        rccl["note"] = ("one broadcast of the packed checkpoint before the timed region, one ragged gather of the fp32 waveforms per step inside "
                        "it (wall time on rank 0: includes waiting for the slowest rank); no collective inside generate()")
        result["rccl"] = rccl
    if rank == 0 and single:
        result["first_chunk_latency"] = first_chunk_leg(model, wl, args.cfg_scale, runs=args.first_chunk_runs, gen_kw=gen_kw)
        log(f"first-chunk latency p50 {result['first_chunk_latency']['p50_ms']} ms")
        try:
            result["first_chunk_latency"]["matrix_core_legs"] = first_chunk_parts(model, wl, cfg)
        except Exception as e:      # noqa: BLE001
            result["first_chunk_latency"]["matrix_core_legs"] = {"error": repr(e)}
    if rank == 0 and single and not args.no_roofline:
        ents = roofline_leg(model, wl, args.cfg_scale)
        log("roofline leg done")
        D, F = cfg.head_hidden, cfg.head_ffn
        head_shapes = {(2, F, D), (2, D, F), (2, cfg.latent, D)}          # re-read by every solver step: served from the Infinity Cache

        ig, ig_note = in_graph_durations()

        def entry(e):
            cached = (e["m"], e["n"], e["k"]) in head_shapes
            traffic, tnote = pmc_traffic(f"{e['m']}x{e['n']}x{e['k']}")
            us = ig.get((e["m"], e["n"], e["k"]))
            in_graph = ({"avg_us": us, "achieved": round(e["weight_bytes"] / us / 1e3, 1), "frac": round(e["weight_bytes"] / us / 1e3 / HBM_PEAK_GBS, 4), "source": ig_note}
                        if us else {"avg_us": None, "frac": None, "source": ig_note})
            return {"bound": "hbm", "in_graph": in_graph, "kernel": f"gemv_stream_kernel (vv_linear m={e['m']} n={e['n']} k={e['k']} dual={e['dual']})",
                    "achieved": round(e["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(e["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_note": tnote, "avg_us": round(e["avg_us"], 2), "bytes_per_launch": e["weight_bytes"], "launches": e["count"],
                    "served_from": "infinity cache (weights re-read by every solver step; HBM sees them once per frame)" if cached else "hbm",
                    "timing": "HIP events on the launch stream around every eager launch of the timed frames (includes ~2-3 us of eager launch overhead "
                              "a graph replay does not pay; `in_graph` carries the rocprofv3 duration of the same launch inside a hipGraph replay)"}
        result["roofline"] = entry(ents[0])
        hbm_ents = [e for e in ents if (e["m"], e["n"], e["k"]) not in head_shapes]
        if hbm_ents:
            result["roofline_hbm_stream"] = entry(hbm_ents[0])
        result["kernels"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items()} for e in ents[:8]]
    # the extra legs never take the headline line down with them
    if rank == 0 and single and headline and args.batched > 1:
        try:
            result["batched_generate"] = batched_leg(model, cfg, args, args.batched)
            log(f"batched x{args.batched}: {result['batched_generate']['value']} audio-sec/s aggregate")
            if model.row_batch and 2 <= args.batched <= 16:
                # the same call with the dialogues on one engine lane each (one LLM / head weight pass per dialogue and frame), for comparison
                lanes = batched_leg(model, cfg, args, args.batched, row_batch=False)
                result["batched_generate"]["lanes_value"] = lanes["value"]
                log(f"batched x{args.batched} on lanes: {lanes['value']} audio-sec/s aggregate")
            if model.row_batch and args.batched == 4:
                # 8 dialogues in one call: two row batches in one loop, every conv tail beside the main stream (modeling._generate_rowbatch)
                result["batched_generate_x8"] = batched_leg(model, cfg, args, 8)
                log(f"batched x8: {result['batched_generate_x8']['value']} audio-sec/s aggregate")
        except Exception as e:      # noqa: BLE001
            result["batched_generate"] = {"error": repr(e)}
            log(f"batched leg failed: {e!r}")
    if rank == 0 and single and headline and args.concurrent > 1:
        try:
            result["concurrent_streams"] = concurrent_leg(model, cfg, sd, dtype, device, args, args.concurrent)
            log(f"concurrent x{args.concurrent}: {result['concurrent_streams']['value']} audio-sec/s aggregate")
        except Exception as e:      # noqa: BLE001
            result["concurrent_streams"] = {"error": repr(e)}
            log(f"concurrent leg failed: {e!r}")
    if rank == 0 and single and headline and args.fp8_leg and dtype == torch.bfloat16 and sd is not None:
        try:
            result["fp8_weights"] = fp8_leg(cfg, sd, device, args, wl)
            log(f"fp8 weights: {result['fp8_weights']['value']} audio-sec/s")
        except Exception as e:      # noqa: BLE001
            result["fp8_weights"] = {"error": repr(e)}
            log(f"fp8 leg failed: {e!r}")
    if rank == 0 and single and not args.no_cpu_baseline:
        if sd is None or args.model != "1.5b":
            cfg15 = VVConfig.preset("1.5b")
            sd15 = synth_state_dict_torch(cfg15, 1234, device=device, dtype=dtype)
        else:
            cfg15, sd15 = cfg, sd
        result["cpu_baseline"] = cpu_baseline_leg(sd15, cfg15)
        log("cpu baseline done")
    if rank == 0:
        emit_result(result)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
