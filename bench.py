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

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def build_workload(cfg, frames, voice_frames, seed=1):
    """Synthetic processor output: ids uniform in [0,1000), one voice prompt, forced schedule."""
    V = cfg.vocab
    ST, SE, SD, EOS = V - 4, V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(seed)
    lim = min(1000, V - 8)
    sys_t = torch.randint(0, lim, (30,), generator=g)
    pre = torch.randint(0, lim, (6,), generator=g)
    txt = torch.randint(0, lim, (88,), generator=g)
    ids = torch.cat([sys_t, pre, torch.tensor([ST]), torch.full((voice_frames,), SD), torch.tensor([SE]), txt, torch.tensor([ST])])
    mask = torch.zeros(ids.shape[0], dtype=torch.bool)
    mask[37: 37 + voice_frames] = True
    gv = torch.Generator().manual_seed(seed + 1)
    voice = torch.randn(voice_frames * cfg.hop - 1234, generator=gv)
    voice = voice * (10 ** (-25 / 20) / (voice.pow(2).mean().sqrt() + 1e-6))          # -25 dBFS like AudioNormalizer
    forced = [SD] * frames + [SE, EOS]
    gn = torch.Generator().manual_seed(seed + 2)
    noise = torch.randn(frames, cfg.latent, generator=gn)
    gs = torch.Generator().manual_seed(seed + 3)
    speech_noise = (torch.randn(1, generator=gs), torch.randn(1, voice_frames, cfg.ac_dim, generator=gs))

    class Tok:
        speech_start_id, speech_end_id, speech_diffusion_id, eos_token_id, bos_token_id, pad_id = ST, SE, SD, EOS, None, 0

    return dict(input_ids=ids[None], speech_input_mask=mask[None], speech_tensors=voice[None],
                speech_masks=torch.ones(1, voice_frames, dtype=torch.bool), forced=forced, noise=noise,
                speech_noise=speech_noise, tok=Tok(), special=dict(speech_start=ST, speech_end=SE, speech_diffusion=SD, eos=EOS))


def bytes_per_frame(cfg, n_steps, mean_ctx, wbytes=2):
    """b*(W_llm + N*W_head + W_dec + W_sem + W_conn) + 2*S*KVb + state."""
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
    b = wbytes * (tot["llm"] + n_steps * tot["head"] + tot["dec"] + tot["sem"] + tot["conn"]) + 2 * mean_ctx * kvb
    b_resident_head = wbytes * (tot["llm"] + tot["head"] + tot["dec"] + tot["sem"] + tot["conn"]) + 2 * mean_ctx * kvb
    return b, b_resident_head, tot


def run_generate(model, wl, cfg_scale, n_frames=None, use_voice=True):
    forced = wl["forced"] if n_frames is None else [wl["special"]["speech_diffusion"]] * n_frames + wl["forced"][-2:]
    kw = dict(input_ids=wl["input_ids"], tokenizer=wl["tok"], cfg_scale=cfg_scale, forced_tokens=forced, noise=wl["noise"],
              generation_config={"do_sample": False}, show_progress_bar=False)
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
        wb = (2 if e.wdt == L.VV_BF16 else 4) * e.n * e.k * (2 if e.dual else 1)
        ents.append(dict(m=e.m, n=e.n, k=e.k, dual=bool(e.dual), count=e.count, total_ms=e.total_ms, avg_us=1e3 * e.total_ms / e.count,
                         weight_bytes=wb, gbs=wb / (e.total_ms / e.count * 1e-3) / 1e9))
    ents.sort(key=lambda d: -d["total_ms"])
    return ents


def first_chunk_leg(model, wl, cfg_scale, runs=5):
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
                       speech_noise=wl["speech_noise"], audio_streamer=st, generation_config={"do_sample": False})
        lat.append(1e3 * (st.t_first - t0))
    lat.sort()
    return dict(p50_ms=round(lat[len(lat) // 2], 2), min_ms=round(lat[0], 2), max_ms=round(lat[-1], 2), runs=runs,
                note="generate() entry -> first 3200-sample chunk on host: voice-prompt encode + LLM prefill + first frame")


def concurrent_leg(model, cfg, sd, dtype, device, args, streams=3):
    """Serving-throughput extra (NOT the headline value): `streams` independent dialogues on ONE GPU, each with its own Engine,
    HIP stream and host thread, sharing the resident weights.  A single dialogue is a latency-bound chain of ~580 small
    kernels per frame, so a second and third chain fill the bubbles."""
    import threading
    from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference
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
    return dict(streams=streams, value=round(sum(outs) / 24000.0 / dt, 3), unit="audio-sec/s", seconds=round(dt, 3),
                note="aggregate over independent dialogues run concurrently on one GPU (one Engine/stream/thread each, shared weights); "
                     "not the headline metric, which is one dialogue per GPU")


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


def cpu_baseline_leg(model, cfg, cfg_scale, n_steps, frames=12, prompt=64):
    """The CPU oracle on a bounded sample of the same workload (kind: port)."""
    from oracle import vv_oracle as O
    # a 1-GPU box's CPU share is 16 cores even when the affinity mask lists the whole host: more threads only thrash
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads, copying weights to host")
    sd = {}
    # the oracle computes in fp32 on the bf16-rounded weights the GPU path streams
    from vibevoice_rocm_amd.synth import synth_state_dict_torch
    src = model._bench_sd
    for k, v in src.items():
        sd[k] = v.detach().float().cpu()
    V = cfg.vocab
    special = dict(speech_start=V - 4, speech_end=V - 3, speech_diffusion=V - 2, eos=V - 1)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, min(1000, V - 8), (prompt,), generator=g).tolist() + [special["speech_start"]]
    noise = torch.randn(frames, cfg.latent, generator=g)
    forced = [special["speech_diffusion"]] * frames + [special["speech_end"], special["eos"]]
    ocfg = cfg.as_dict()
    log("cpu baseline: weights on host, running the oracle")
    t0 = time.time()
    res = O.generate(sd, ocfg, ids, None, None, special, noise, cfg_scale=cfg_scale, n_steps=n_steps, forced_tokens=forced)
    dt = time.time() - t0
    audio_s = len(res.audio) * cfg.hop / 24000.0
    return dict(value=audio_s / dt, unit="audio-sec/s", cores=cores, kind="port",
                sample=f"oracle/vv_oracle.py generate(): {prompt + 1}-token prompt (no voice prompt), {frames} frames, CFG={cfg_scale}, "
                       f"{n_steps} steps, fp32 torch-CPU on the bf16-rounded weights, {dt:.1f} s wall")


_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="1.5b")
    ap.add_argument("--frames", type=int, default=225)
    ap.add_argument("--voice-frames", type=int, default=203)
    ap.add_argument("--cfg-scale", type=float, default=2.0)
    ap.add_argument("--ddpm-steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-fp8-leg", dest="fp8_leg", action="store_false", help="skip the extra weight-only-fp8 leg")
    ap.add_argument("--concurrent", type=int, default=3, help="extra leg: N independent dialogues concurrently on one GPU (0/1 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VV_DIST_BACKEND", "nccl")          # "gloo" lets two ranks rehearse on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("VV_ALL_RANKS_ONE_GPU") == "1":
        local_rank = 0
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
    if world > 1:
        sd = vd.broadcast_state_dict(sd, cfg, dtype, device, src=0)
    log("weights ready")
    model = VibeVoiceForConditionalGenerationInference(cfg, sd, device=device, torch_dtype=dtype, use_graphs=not args.no_graphs)
    log(f"engine ready ({model.engine.w.nbytes() / 1e9:.2f} GB resident)")
    model._bench_sd = sd if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    model._bench_sd_all = sd if (rank == 0 and world == 1 and (args.concurrent > 1 or args.fp8_leg)) else None
    if model._bench_sd is None and model._bench_sd_all is None:
        del sd
    model.set_ddpm_inference_steps(args.ddpm_steps)
    wl = build_workload(cfg, args.frames, args.voice_frames, seed=1 + rank)     # every rank its own dialogue
    for k in ("input_ids", "speech_input_mask", "speech_tensors", "speech_masks"):
        wl[k] = wl[k].to(device) if k == "speech_tensors" else wl[k]

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for i in range(args.warmup):
        tw = time.perf_counter()
        out = run_generate(model, wl, args.cfg_scale)
        torch.cuda.synchronize()
        log(f"warmup {i}: {time.perf_counter() - tw:.2f} s")
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run_generate(model, wl, args.cfg_scale)
        if dist is not None:
            vd.gather_waveforms(out.speech_outputs[0], dst=0)
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
        mean_ctx = L0 + args.frames / 2
        wb = 2 if dtype == torch.bfloat16 else 4
        bpf, bpf_res, _ = bytes_per_frame(cfg, args.ddpm_steps, mean_ctx, wb)
        s_per_frame = dt / (args.steps * args.frames)
        result = {
            "metric": "audio-sec/s", "value": round(value, 4), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, seeded prompt/voice/noise, forced token schedule)",
            "config": {"workload": f"VibeVoice-{args.model.upper()} shapes, 1 speaker, {args.voice_frames}-frame voice prompt, "
                                   f"{L0}-token prompt, {args.frames} frames ({audio_s:.1f} s audio), CFG={args.cfg_scale}, "
                                   f"{args.ddpm_steps} DPM-Solver++ steps, whole generate() incl. voice encode + prefill",
                       "global_batch": world, "parallelism": f"replicas x{world} (one dialogue per GPU)", "hipgraph": not args.no_graphs},
            "frame": {"ms_per_frame_incl_prefill": round(1e3 * s_per_frame, 4), "algorithmic_GB_per_frame": round(bpf / 1e9, 4),
                      "achieved_GBps": round(bpf / s_per_frame / 1e9, 1), "frac_of_hbm_peak": round(bpf / s_per_frame / 1e9 / HBM_PEAK_GBS, 4),
                      "frac_if_head_weights_counted_once": round(bpf_res / s_per_frame / 1e9 / HBM_PEAK_GBS, 4)},
        }
    single = world == 1            # the diagnostic legs run at N = 1 only (other ranks would just wait at the final barrier)
    if rank == 0 and single:
        result["first_chunk_latency"] = first_chunk_leg(model, wl, args.cfg_scale)
        log(f"first-chunk latency p50 {result['first_chunk_latency']['p50_ms']} ms")
    if rank == 0 and single and not args.no_roofline:
        ents = roofline_leg(model, wl, args.cfg_scale)
        log("roofline leg done")
        top = ents[0]
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                with open(tp) as f:
                    traffic = json.load(f).get(f"{top['m']}x{top['n']}x{top['k']}")
            except Exception:
                traffic = None
        result["roofline"] = {"bound": "hbm", "kernel": f"gemv_stream_kernel (vv_linear m={top['m']} n={top['n']} k={top['k']} dual={top['dual']})",
                              "achieved": round(top["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(top["gbs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                              "avg_us": round(top["avg_us"], 2), "bytes_per_launch": top["weight_bytes"], "launches": top["count"]}
        result["kernels"] = [{k: (round(v, 2) if isinstance(v, float) else v) for k, v in e.items()} for e in ents[:8]]
    # the extra legs never take the headline line down with them
    if rank == 0 and single and args.concurrent > 1:
        try:
            result["concurrent_streams"] = concurrent_leg(model, cfg, model._bench_sd_all, dtype, device, args, args.concurrent)
            log(f"concurrent x{args.concurrent}: {result['concurrent_streams']['value']} audio-sec/s aggregate")
        except Exception as e:      # noqa: BLE001
            result["concurrent_streams"] = {"error": repr(e)}
            log(f"concurrent leg failed: {e!r}")
    if rank == 0 and single and args.fp8_leg and dtype == torch.bfloat16 and model._bench_sd_all is not None:
        try:
            result["fp8_weights"] = fp8_leg(cfg, model._bench_sd_all, device, args, wl)
            log(f"fp8 weights: {result['fp8_weights']['value']} audio-sec/s")
        except Exception as e:      # noqa: BLE001
            result["fp8_weights"] = {"error": repr(e)}
            log(f"fp8 leg failed: {e!r}")
    if rank == 0 and single and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_leg(model, cfg, args.cfg_scale, args.ddpm_steps)
        log("cpu baseline done")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
