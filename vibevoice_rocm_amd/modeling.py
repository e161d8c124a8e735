"""Host-side mirror of the reference's inference class on top of the MI355X engine.

`VibeVoiceForConditionalGenerationInference` keeps the call surface `demo/inference_from_file.py` (and the Gradio
apps) use — `from_pretrained(path, torch_dtype=, device_map=, attn_implementation=)`, `.eval()`,
`.set_ddpm_inference_steps(num_steps=)`, `.generate(**processor_outputs, max_new_tokens=None, cfg_scale=,
tokenizer=, generation_config={'do_sample': False}, verbose=, audio_streamer=, stop_check_fn=, ...)`,
`.model.language_model.config._attn_implementation`, `.ddpm_inference_steps`, `.device`, `.model.noise_scheduler` —
and returns the same `VibeVoiceGenerationOutput(sequences, speech_outputs, reach_max_step_sample)`
(reference: vibevoice/modular/modeling_vibevoice_inference.py:38-51,68-147,326-693).
The loop below restates :364-693 for one utterance per engine; a batch is served utterance by utterance
(they are independent: SURVEY.md §8e).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .config import VVConfig
from .engine import Engine
from .synth import state_dict_shapes

LANES_IN_FLIGHT = 4       # lock-step batches: lanes (one HIP stream each) enqueued concurrently; see _generate_lockstep


@dataclass
class VibeVoiceGenerationOutput:
    sequences: torch.LongTensor = None
    speech_outputs: Optional[List[Optional[torch.Tensor]]] = None
    reach_max_step_sample: Optional[torch.BoolTensor] = None


def load_state_dict_from_dir(path: str) -> Dict[str, torch.Tensor]:
    """HF sharded-safetensors checkpoint as the reference's converter writes it
    (vibevoice/scripts/convert_nnscaler_checkpoint_to_transformers.py:116-123)."""
    from safetensors.torch import load_file
    idx = os.path.join(path, "model.safetensors.index.json")
    files = []
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(path, "model.safetensors")):
        files = ["model.safetensors"]
    else:
        raise FileNotFoundError(f"no model.safetensors[.index.json] under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for fn in files:
        sd.update(load_file(os.path.join(path, fn)))
    return sd


def save_checkpoint_dir(path: str, config: VVConfig, state_dict: Dict[str, torch.Tensor], max_shard_bytes: int = 2 * 10 ** 9,
                        language_model_pretrained_name: str = "Qwen/Qwen2.5-1.5B") -> None:
    """Write a checkpoint directory in the layout the reference's converter produces
    (vibevoice/scripts/convert_nnscaler_checkpoint_to_transformers.py:92-123): `config.json` in the reference's schema (incl. the
    `vibepod_*` model_type keys of vibevoice/configs/*.json), `preprocessor_config.json`, and safetensors shards of at most
    `max_shard_bytes` with `model.safetensors.index.json` (a single `model.safetensors` when everything fits one shard).
    Tied checkpoints carry no `lm_head.weight` (HF save_pretrained drops the alias)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    j = config.to_reference_json()
    j["model_type"] = "vibepod"
    j["acoustic_tokenizer_config"]["model_type"] = "vibepod_acoustic_tokenizer"
    j["semantic_tokenizer_config"]["model_type"] = "vibepod_semantic_tokenizer"
    j["diffusion_head_config"]["model_type"] = "vibepod_diffusion_head"
    j["torch_dtype"] = j["decoder_config"]["torch_dtype"] = "bfloat16"
    j["tie_word_embeddings"] = bool(config.tie)             # the converter passes it to VibeVoiceConfig (:46-50)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(j, f, indent=2, sort_keys=True)
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:
        json.dump({"processor_class": "VibeVoiceProcessor", "speech_tok_compress_ratio": config.hop, "db_normalize": True,
                   "audio_processor": {"feature_extractor_type": "VibeVoiceTokenizerProcessor", "sampling_rate": 24000,
                                       "normalize_audio": True, "target_dB_FS": -25, "eps": 1e-6},
                   "language_model_pretrained_name": language_model_pretrained_name}, f, indent=2)
    names = [k for k in state_dict if not (config.tie and k == "lm_head.weight")]
    shards, cur, cur_bytes = [], {}, 0
    for k in names:
        t = state_dict[k].detach().cpu().contiguous()
        nb = t.numel() * t.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = t
        cur_bytes += nb
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
        return
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(path, fn), metadata={"format": "pt"})
        for k, t in sh.items():
            weight_map[k] = fn
            total += t.numel() * t.element_size()
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)


def _make_sampler(gen_cfg: dict):
    """do_sample path (modeling_vibevoice_inference.py:491-494): softmax over the constrained logits + multinomial, with the
    HF warpers the reference's callers configure (temperature, top_k, top_p; main.py:1187-1196) applied in HF order.
    Runs on the host over the 4-5 valid logits; draws from torch's CPU generator."""
    temperature = float(gen_cfg.get("temperature", 1.0) or 1.0)
    top_k = int(gen_cfg.get("top_k", 0) or 0)
    top_p = float(gen_cfg.get("top_p", 1.0) or 1.0)

    def sample(logits: torch.Tensor, ids):
        z = logits.double() / temperature
        if 0 < top_k < z.numel():
            kth = torch.topk(z, top_k).values[-1]
            z = torch.where(z < kth, torch.full_like(z, float("-inf")), z)
        if top_p < 1.0:
            sz, order = torch.sort(z, descending=False)
            cum = torch.softmax(sz, -1).cumsum(-1)
            remove = cum <= (1 - top_p)
            remove[-1] = False
            z[order[remove]] = float("-inf")
        p = torch.softmax(z, -1)
        return ids[int(torch.multinomial(p.float(), 1))]
    return sample


class VibeVoiceForConditionalGenerationInference:
    def __init__(self, config: VVConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0", torch_dtype=torch.bfloat16,
                 attn_implementation: str = "hip_gfx950", use_graphs: bool = True, weight_quant: Optional[str] = None):
        missing = [k for k in state_dict_shapes(config) if k not in state_dict]
        if missing:
            raise KeyError(f"state dict is missing {len(missing)} tensors, e.g. {missing[:4]}")
        self.config = config
        self.dtype = torch_dtype
        # launch a frame's diffusion tail right behind the LLM step while the host still waits for the token (rolled back when the
        # token is not speech_diffusion); results are identical either way (tests/test_hip_parity.py)
        self.speculative_frames = True
        # batches of 2..8 dialogues: one weight pass per frame for all of them (rowbatch.py) instead of one per dialogue (lanes)
        self.row_batch = os.environ.get("VV_ROW_BATCH", "1") != "0"
        self.row_batch_min = int(os.environ.get("VV_ROW_BATCH_MIN", "2"))      # 2 dialogues: 64 vs 58 audio-sec/s on the lanes
        self._rowbatch = {}
        # weight_quant="fp8": weight-only e4m3 companions for the per-frame weight-streaming GEMVs
        self.weight_quant = weight_quant
        self.engine = Engine(config, state_dict, device=device, dtype=torch_dtype, use_graphs=use_graphs, weight_quant=weight_quant)
        self.device = self.engine.device
        # batches run in lock step on one Engine per sample (own HIP stream, KV cache and conv state; matrices already in the streamed
        # dtype on the device are shared, not copied): lanes beyond the first are built on first use from this state dict
        self._lanes = [self.engine]
        self._use_graphs = use_graphs
        self.ddpm_inference_steps = config.ddpm_infer
        # attribute paths the reference's callers read
        lm_cfg = SimpleNamespace(_attn_implementation=attn_implementation, hidden_size=config.hidden,
                                 max_position_embeddings=config.max_pos)
        self.model = SimpleNamespace(language_model=SimpleNamespace(config=lm_cfg), noise_scheduler=self.engine.scheduler)

    # ---- construction ----------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.bfloat16, device_map=None,
                        attn_implementation: Optional[str] = None, **kw):
        """Load a local checkpoint directory (config.json + safetensors shards).  `attn_implementation` is accepted for
        call compatibility; attention always runs on the hand-written gfx950 kernel."""
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise OSError(f"{path!r} is not a local checkpoint directory (this build has no hub access)")
        cfg = VVConfig.from_pretrained(path)
        device = device_map if isinstance(device_map, (str, torch.device)) and str(device_map) not in ("auto", "cpu") else "cuda:0"
        if str(device) == "cuda":
            device = "cuda:0"
        return cls(cfg, load_state_dict_from_dir(path), device=device, torch_dtype=torch_dtype,
                   attn_implementation=attn_implementation or "hip_gfx950", use_graphs=kw.get("use_graphs", True),
                   weight_quant=kw.get("weight_quant"))

    @classmethod
    def from_synthetic(cls, config: VVConfig, seed: int = 1234, device="cuda:0", torch_dtype=torch.bfloat16, numpy_weights=False, **kw):
        """Random-init weights of the architecture (no checkpoints exist offline)."""
        from .synth import synth_state_dict, synth_state_dict_torch
        if numpy_weights:
            sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(config, seed).items()}
        else:
            sd = synth_state_dict_torch(config, seed, device=device, dtype=torch_dtype)
        return cls(config, sd, device=device, torch_dtype=torch_dtype, **kw)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def set_ddpm_inference_steps(self, num_steps=None):
        self.ddpm_inference_steps = num_steps or self.config.ddpm_infer
        # main.py:543-548 swaps the scheduler first: model.model.noise_scheduler = ....from_config(cfg, algorithm_type="sde-dpmsolver++", ...)
        if self.model.noise_scheduler is not self.engine.scheduler:
            self.engine.scheduler = self.model.noise_scheduler
        self.engine.set_steps(self.ddpm_inference_steps)

    @property
    def noise_scheduler(self):
        return self.engine.scheduler

    # ---- voice-prompt prefill --------------------------------------------------------------------------------
    def _process_speech_inputs(self, speech_tensors: torch.Tensor, speech_masks: torch.Tensor,
                               std_noise: Optional[torch.Tensor] = None, eps_noise: Optional[torch.Tensor] = None):
        """modeling_vibevoice_inference.py:149-163.  Returns (acoustic_features [S,F,64], connected [sum(mask), H])."""
        eng, cfg = self.engine, self.config
        with torch.cuda.stream(eng.stream):      # every torch op of the prefill rides the engine stream
            return self._process_speech_inputs_impl(speech_tensors, speech_masks, std_noise, eps_noise)

    def _process_speech_inputs_impl(self, speech_tensors, speech_masks, std_noise, eps_noise):
        eng, cfg = self.engine, self.config
        S, Tmax = speech_tensors.shape
        Fm = speech_masks.shape[1]
        n_frames = speech_masks.sum(-1).tolist()
        means = torch.zeros(S, Fm, cfg.ac_dim, dtype=torch.float32, device=self.device)
        # causal encoder: frames < n_frames[i] only see samples < n_frames[i]*hop, so the zero tail of the padded batch
        # row beyond that boundary never matters; when the boundary exceeds Tmax the reference pads features instead.
        t_len = [min(int(n_frames[i]) * cfg.hop, Tmax) for i in range(S)]
        live = [i for i in range(S) if t_len[i] > 0]
        with torch.cuda.stream(eng.stream):
            st_dev = speech_tensors.to(self.device)
        encoded = eng.acoustic_encode_many([st_dev[i, :t_len[i]] for i in live],      # the S voices run concurrently, on the lanes' streams if there are lanes
                                           side_streams=[e.stream for e in self._lanes[1:]])
        with torch.cuda.stream(eng.stream):
            for i, m in zip(live, encoded):
                means[i, : m.shape[0]] = m
        with torch.cuda.stream(eng.stream):
            if cfg.ac_std_dist == "gaussian":
                if std_noise is None:
                    std_noise = torch.randn(S, device=self.device, dtype=self.dtype)
                    eps_noise = torch.randn_like(torch.empty(S, cfg.ac_dim, Fm, device=self.device, dtype=self.dtype).permute(0, 2, 1))
                std = std_noise.to(self.device).float() * (cfg.ac_fix_std / 0.8)
                lat = means + std[:, None, None] * eps_noise.to(self.device).float()
            else:
                lat = means
            feats = (lat + eng.w.speech_bias) * eng.w.speech_scale
            sel = feats[speech_masks.to(self.device)]
        conn = eng.connector("acoustic", sel)
        return feats, conn

    # ---- generation --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                 speech_tensors: Optional[torch.Tensor] = None, speech_masks: Optional[torch.Tensor] = None,
                 speech_input_mask: Optional[torch.Tensor] = None, tokenizer=None, generation_config=None,
                 max_new_tokens: Optional[int] = None, cfg_scale: float = 1.0, audio_streamer=None,
                 stop_check_fn: Optional[Callable[[], bool]] = None, return_speech: bool = True, **kwargs):
        self.engine.sync_in()
        with torch.cuda.stream(self.engine.stream):   # all torch glue (gathers, scatters, copies) rides the engine stream
            out = self._generate(input_ids, attention_mask, speech_tensors, speech_masks, speech_input_mask, tokenizer,
                                 generation_config, max_new_tokens, cfg_scale, audio_streamer, stop_check_fn, return_speech, **kwargs)
        self.engine.stream.synchronize()
        return out

    def _generate(self, input_ids, attention_mask, speech_tensors, speech_masks, speech_input_mask, tokenizer, generation_config,
                  max_new_tokens, cfg_scale, audio_streamer, stop_check_fn, return_speech, **kwargs):
        if tokenizer is None:
            raise ValueError("generate() needs tokenizer= (for the speech_start/end/diffusion and eos ids)")
        gen_cfg = dict(generation_config or {})
        sample_fn = _make_sampler(gen_cfg) if gen_cfg.get("do_sample", False) else None
        verbose = kwargs.get("verbose", False)
        max_length_times = kwargs.get("max_length_times", 2)
        refresh_negative = bool(kwargs.get("refresh_negative", True))     # False: reference :501-515 (no caller of the reference uses it)
        self._prefill_chunk = int(kwargs.get("prefill_chunk", 1024))   # extension: rows per prefill launch sequence (cfg 5: 512-token chunks)
        forced_tokens = kwargs.get("forced_tokens")          # extension: bench / fixtures drive the token schedule
        noise = kwargs.get("noise")                          # extension: injected diffusion noise [F, latent]
        speech_noise = kwargs.get("speech_noise")            # extension: (std_noise [S], eps_noise [S, F, 64])
        sde_noise = kwargs.get("sde_noise")                  # extension: injected variance noise of the SDE solver [F, n_steps, latent]
        input_ids = torch.as_tensor(input_ids)
        in_dev = input_ids.device                            # callers may hand over device tensors (the reference moves them itself, modeling_vibevoice_inference.py:288,305-307)
        input_ids = input_ids.cpu()                          # ids / masks drive host-side bookkeeping only
        if input_ids.dim() == 1:
            input_ids = input_ids[None]
        B, Lp = input_ids.shape
        attention_mask = torch.ones_like(input_ids) if attention_mask is None else torch.as_tensor(attention_mask).cpu()
        if speech_input_mask is not None:
            speech_input_mask = torch.as_tensor(speech_input_mask).cpu()
        special = dict(speech_start=tokenizer.speech_start_id, speech_end=tokenizer.speech_end_id,
                       speech_diffusion=tokenizer.speech_diffusion_id, eos=tokenizer.eos_token_id,
                       bos=getattr(tokenizer, "bos_token_id", None))
        conn_all = None
        if speech_tensors is not None and speech_masks is not None:
            sn = speech_noise or (None, None)
            _, conn_all = self._process_speech_inputs(torch.as_tensor(speech_tensors).float(), torch.as_tensor(speech_masks).bool(), *sn)
        if B > 1:
            if not refresh_negative:
                # with refresh_negative=False the reference's batched loop couples the samples (a non-diffusing sample's negative step is
                # dropped only in steps where some OTHER sample diffuses, :588-622): served per sample here would not reproduce that
                raise NotImplementedError("refresh_negative=False is built for batch size 1 only")
            rb = kwargs.get("row_batch", self.row_batch)
            fn = self._generate_lockstep
            if rb and self.row_batch_min <= B <= 16 and sample_fn is None and self.dtype == torch.bfloat16 and self.weight_quant is None and not self.engine.sde:
                fn = self._generate_rowbatch      # dialogues batched into the row dimension of the LLM / diffusion-head weight passes (rowbatch.py)
            return fn(input_ids, attention_mask, speech_input_mask, conn_all, special, cfg_scale, max_new_tokens, max_length_times,
                                           forced_tokens, None if noise is None else torch.as_tensor(noise), None if sde_noise is None else torch.as_tensor(sde_noise),
                                           audio_streamer, stop_check_fn, verbose, sample_fn, tokenizer, return_speech, in_dev)
        seqs, audios, reach = [], [], []
        off = 0
        for b in range(B):
            keep = attention_mask[b].bool()
            ids_b = input_ids[b][keep]                       # left padding carries no information (position_ids = cumsum(mask)-1)
            sp_b = None
            conn_b = None
            if speech_input_mask is not None:
                sp_b = torch.as_tensor(speech_input_mask)[b][keep].bool()
                n_b = int(sp_b.sum())
                if n_b and conn_all is not None:
                    conn_b = conn_all[off: off + n_b]
                    off += n_b
            r = self._generate_one(ids_b, sp_b, conn_b, special, cfg_scale, max_new_tokens, max_length_times, forced_tokens,
                                   None if noise is None else torch.as_tensor(noise), audio_streamer, stop_check_fn, b, verbose, sample_fn,
                                   None if sde_noise is None else torch.as_tensor(sde_noise), refresh_negative=refresh_negative)
            seqs.append(torch.cat([input_ids[b][~keep], r["sequence"]]))
            audios.append(r["audio"])
            reach.append(r["reach_max"])
        if audio_streamer is not None:
            audio_streamer.end()
        mx = max(s.shape[0] for s in seqs)
        pad_id = getattr(tokenizer, "pad_id", None)
        if pad_id is None:
            pad_id = special["eos"]
        seq_t = torch.full((B, mx), int(pad_id), dtype=torch.long)
        for b, s in enumerate(seqs):
            seq_t[b, : s.shape[0]] = s
        return VibeVoiceGenerationOutput(sequences=seq_t.to(in_dev), speech_outputs=audios if return_speech else None,
                                         reach_max_step_sample=torch.tensor(reach, dtype=torch.bool))


    # ---- batches: lock step over samples (modeling_vibevoice_inference.py:430-673 with batch_size > 1) ---------------------
    def release_lanes(self) -> None:
        """Drop the engines of earlier batched calls (lanes, row batches): their HIP streams go back to the recycle pool (engine.py, _IDLE_STREAMS)."""
        for rb in self._rowbatch.values():
            rb.close()
        self._rowbatch = {}
        for e in self._lanes[1:]:
            e.stream.synchronize()
            e.close()
        self._lanes = self._lanes[:1]

    def _lane(self, b: int) -> Engine:
        while len(self._lanes) <= b:
            n = len(self._lanes)             # lanes past LANES_IN_FLIGHT share the stream of lane n % LANES_IN_FLIGHT: see _generate_lockstep
            eng = Engine(self.config, None, device=self.device, dtype=self.dtype, use_graphs=self._use_graphs, weight_quant=self.weight_quant,
                         stream=self._lanes[n % LANES_IN_FLIGHT].stream if n >= LANES_IN_FLIGHT else None, weights_from=self.engine)
            eng.scheduler = self.engine.scheduler
            self._lanes.append(eng)
        eng = self._lanes[b]
        if eng.scheduler is not self.engine.scheduler or eng.n_steps != self.engine.n_steps:
            eng.scheduler = self.engine.scheduler
            eng.set_steps(self.engine.n_steps)
        return eng

    def _generate_lockstep(self, input_ids, attention_mask, speech_input_mask, conn_all, special, cfg_scale, max_new_tokens, max_length_times,
                           forced_tokens, noise, sde_noise, audio_streamer, stop_check_fn, verbose, sample_fn, tokenizer, return_speech, in_dev):
        """One host loop drives B engines in lock step, as the reference's batched generate() does: every live sample takes its LLM step
        (phase A of all of them is enqueued before any token is awaited, so the B dependent chains fill each other's bubbles on the GPU),
        tokens are handled per sample (:517-563), the samples that emitted speech_diffusion are sampled / decoded / re-embedded (:571-670)
        and their chunks reach the AudioStreamer together, once per step (:644-653).  A sample is the same computation as in a batch of
        one - bit for bit with injected noise; the random draws follow the reference's order (randn(2 n, latent) per step for the n
        diffusing samples, rows [:n] used, :699).  `forced_tokens` / `noise` / `sde_noise` may be given per sample (list / leading batch
        dimension) or once for all."""
        cfg = self.config
        B, Lp = input_ids.shape
        ST, SE, SD, EOS = special["speech_start"], special["speech_end"], special["speech_diffusion"], special["eos"]
        valid = [ST, SE, SD, EOS] + ([special["bos"]] if special.get("bos") is not None else [])
        lanes = [self._lane(b) for b in range(B)]
        for e in lanes[1:]:
            e.sync_in()
        keep = attention_mask.bool()
        L0 = keep.sum(-1).tolist()
        max_length = cfg.max_pos if max_new_tokens is None else Lp + int(max_new_tokens)            # :370-371 (padded length, as the reference)
        max_steps = min(max_length - Lp, int(max_length_times * Lp))                                # :420
        max_step_per_sample = [min(max_length - l, int(max_length_times * l)) for l in L0]          # :421
        per_list = forced_tokens is not None and len(forced_tokens) > 0 and isinstance(forced_tokens[0], (list, tuple))
        ftok = [(forced_tokens[b] if per_list else forced_tokens) for b in range(B)]
        nz = [(noise[b] if (noise is not None and noise.dim() == 3) else noise) for b in range(B)]
        snz = [(sde_noise[b] if (sde_noise is not None and sde_noise.dim() == 4) else sde_noise) for b in range(B)]
        x0s, off = [], 0
        for b in range(B):
            eng = lanes[b]
            ids_b = input_ids[b][keep[b]]
            eng.cfg_scale = float(cfg_scale)
            eng.begin_sequence(L0[b] + max(max_steps, 1) + 8, valid)
            with torch.cuda.stream(eng.stream):
                x0 = eng.embed_ids(ids_b)
                if speech_input_mask is not None and conn_all is not None:
                    sp_b = speech_input_mask[b][keep[b]].bool()
                    n_b = int(sp_b.sum())
                    if n_b:
                        eng.stream.wait_stream(self.engine.stream)                                   # conn_all was produced on lane 0's stream
                        x0[sp_b.to(self.device)] = conn_all[off: off + n_b]                         # :221-224
                        off += n_b
            x0s.append(x0)
        seq = [input_ids[b][keep[b]].tolist() for b in range(B)]
        chunks = [[] for _ in range(B)]
        frame = [0] * B
        finished = [False] * B
        reach = [False] * B
        prev_tok = [None] * B
        pending = []                      # (sample, ring slot) of chunks not yet handed to the streamer
        ours = [False] * max(B, getattr(audio_streamer, "batch_size", B) if audio_streamer is not None else B)   # streams ended by this loop
        speculate = self.speculative_frames and sample_fn is None and self._use_graphs
        sde = lanes[0].sde

        def deliver():
            if audio_streamer is None or not pending:
                pending.clear()
                return
            idx = [b for b, _ in pending]
            audio_streamer.put(torch.stack([lanes[b].take_chunk(k)[None] for b, k in pending]), torch.tensor(idx))   # one put per step, all samples (:644-653)
            pending.clear()

        def draw(n):
            """the reference's draws for a step with n diffusing samples: randn(2 n, latent), rows [:n]; SDE: n_steps more of the same"""
            a = torch.randn(2 * n, cfg.latent)[:n]
            s_ = torch.stack([torch.randn(2 * n, cfg.latent)[:n] for _ in range(lanes[0].n_steps)], dim=1) if sde else None
            return a, s_

        pool = None
        if B > 1 and os.environ.get("VV_LANE_THREADS", "1") != "0":
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=min(B, LANES_IN_FLIGHT))
        for step in range(max_steps):
            if stop_check_fn is not None and stop_check_fn():                                       # :432-438
                if verbose:
                    print(f"Generation stopped externally at step {step + 1}")
                deliver()
                if audio_streamer is not None:
                    audio_streamer.end()
                break
            if audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and \
                    any(f and not ours[i] for i, f in enumerate(audio_streamer.finished_flags)):
                break       # :441-445 "stopped externally".  Deviation, on purpose: the reference tests any(finished_flags), which its own
                            # end(new_eos_indices) at :526 also sets - a batch with a streamer then stops at the FIRST sample's EOS; here
                            # only streams ended by someone else stop the batch, streams this loop ended itself (EOS / max length) do not
            if all(finished):
                break
            if Lp + step >= max_length:                                                             # :452-457
                for b in range(B):
                    reach[b] = reach[b] or not finished[b]
                break
            live = [b for b in range(B) if not finished[b]]
            forced = {b: (ftok[b][step] if (ftok[b] is not None and step < len(ftok[b])) else None) for b in live}
            toks = {}
            speculated = set()
            if step == 0:
                for b in live:
                    lanes[b].prefill(x0s[b], row=0, pos0=0, chunk=getattr(self, "_prefill_chunk", 1024), neg_embed=lanes[b].embed_ids(torch.tensor([ST])))
                for b in live:
                    toks[b] = lanes[b].first_token(ST, SD, forced[b], sample_fn)
                    if toks[b] == SD:
                        lanes[b].commit_negative_prompt()
            elif sample_fn is not None:
                for b in live:
                    toks[b] = lanes[b].step_decode(ST, SD, forced[b], sample_fn)
            else:
                # phase A of every live sample goes out before any token is awaited; a sample in the steady state of a dialogue gets its
                # diffusion tail enqueued speculatively behind it when its noise is injected (drawn noise depends on how many samples
                # diffuse in this step, which is only known once the tokens are)
                specs = {}
                for b in live:
                    specs[b] = None
                    if speculate and prev_tok[b] == SD and nz[b] is not None and frame[b] < len(nz[b]) and (not sde or (snz[b] is not None and frame[b] < len(snz[b]))):
                        specs[b] = (nz[b][frame[b]], snz[b][frame[b]] if sde else None)
                        speculated.add(b)
                # a frame is ~600 graph nodes and the runtime enqueues them node by node: the lanes' launches go out from one host
                # thread each (the HIP calls release the GIL), or the host becomes the bottleneck at batch > 2
                # at most LANES_IN_FLIGHT lanes run at once: more streams than that serialise badly on MI355X (8 streams in flight are
                # slower than 4), so lane b of a larger batch shares the HIP stream of lane b % LANES_IN_FLIGHT and stream order queues
                # it behind that lane's frame.  One host thread per stream (never two threads on one stream: first use captures graphs).
                def begin(s_):
                    for b in live:
                        if b % LANES_IN_FLIGHT == s_:
                            lanes[b].decode_begin(ST, SD, forced[b], specs[b])
                slots = sorted({b % LANES_IN_FLIGHT for b in live})
                if pool is not None and len(slots) > 1:
                    list(pool.map(begin, slots))
                else:
                    for s_ in slots:
                        begin(s_)
                deliver()                  # the previous step's chunks: their copies completed long before this step's tokens
                for b in live:
                    toks[b] = lanes[b].decode_end()
            diffusing = []
            for b in live:
                tok = toks[b]
                if b in speculated and tok != SD:
                    lanes[b].rollback_speech_state()
                prev_tok[b] = tok
                seq[b].append(tok)
                if tok == EOS:                                                                      # :517-526
                    finished[b] = True
                    if verbose:
                        print(f"Samples [{b}] reached EOS token at step {step + 1}.", flush=True)
                    if audio_streamer is not None:
                        deliver()
                        ours[b] = True
                        audio_streamer.end(torch.tensor([b]))
                    continue
                if step >= max_step_per_sample[b]:                                                  # :528-537
                    finished[b] = True
                    reach[b] = True
                    if audio_streamer is not None:
                        deliver()
                        ours[b] = True
                        audio_streamer.end(torch.tensor([b]))
                    continue
                if tok == SE:                                                                       # :540-544
                    with torch.cuda.stream(lanes[b].stream):
                        lanes[b].reset_speech_caches()
                if tok == SD:
                    diffusing.append(b)
                else:
                    lanes[b].step_embed()                                                           # :567
            need = [b for b in diffusing if b not in speculated and (nz[b] is None or frame[b] >= len(nz[b]))]
            drawn = draw(len(need)) if need else None
            for b in diffusing:                                                                     # :571-670
                if b not in speculated:
                    if b in need:
                        i = need.index(b)
                        n_row, s_row = drawn[0][i], (drawn[1][i] if sde else None)
                    else:
                        n_row, s_row = nz[b][frame[b]], (snz[b][frame[b]] if sde else None)
                    lanes[b].step_speech(n_row, s_row)
                with torch.cuda.stream(lanes[b].stream):
                    chunks[b].append(lanes[b].wav.clone())
                if audio_streamer is not None:
                    pending.append((b, lanes[b].stage_chunk()))
                frame[b] += 1
        deliver()
        if pool is not None:
            pool.shutdown()
        for e in lanes[:B]:
            e.stream.synchronize()
        if audio_streamer is not None:
            audio_streamer.end()
        pad_id = getattr(tokenizer, "pad_id", None)
        if pad_id is None:
            pad_id = special["eos"]
        rows = []
        for b in range(B):
            rows.append(torch.cat([input_ids[b][~keep[b]], torch.tensor(seq[b], dtype=torch.long)]))
        mx = max(r.shape[0] for r in rows)
        seq_t = torch.full((B, mx), int(pad_id), dtype=torch.long)
        for b, r in enumerate(rows):
            seq_t[b, : r.shape[0]] = r
        audios = [(torch.cat(c)[None] if c else None) for c in chunks]
        return VibeVoiceGenerationOutput(sequences=seq_t.to(in_dev), speech_outputs=audios if return_speech else None,
                                         reach_max_step_sample=torch.tensor(reach, dtype=torch.bool))

    def _generate_rowbatch(self, input_ids, attention_mask, speech_input_mask, conn_all, special, cfg_scale, max_new_tokens, max_length_times,
                           forced_tokens, noise, sde_noise, audio_streamer, stop_check_fn, verbose, sample_fn, tokenizer, return_speech, in_dev):
        """The lock-step loop of `_generate_lockstep` with the B dialogues batched into the ROW dimension of the weight-heavy half of a frame
        (rowbatch.RowBatch: one Qwen2 decode step with 2 B rows, one diffusion sampling with 2 B rows; the conv tokenizers stay per dialogue on
        their lanes' streams).  5..16 dialogues run as ceil(B / 4) row batches inside the same loop, all on the main stream: each step enqueues A and
        H of every batch, then the conv tails - a batch's tails overlap the other batches' A and H.  Token handling, the draws' order, speculation and rollback
        are those of the lock-step loop; results agree with the lanes to the rounding of the matrix-core GEMV (activations as bf16 hi + lo,
        ~2e-6 relative per product)."""
        from .rowbatch import RowBatch
        cfg = self.config
        B, Lp = input_ids.shape
        ST, SE, SD, EOS = special["speech_start"], special["speech_end"], special["speech_diffusion"], special["eos"]
        valid = [ST, SE, SD, EOS] + ([special["bos"]] if special.get("bos") is not None else [])
        if B > 4:
            # two row batches: no dialogue's conv tail on the main stream (lanes 0, 4, 8, ... live there) - the main stream then runs A and H of
            # the two batches back to back while all tails run beside it on the three side streams (8 dialogues: 108 -> 119 audio-sec/s, 6: 88 -> 99)
            idx = [i for i in range(3 * B) if i % LANES_IN_FLIGHT][:B]
            lanes = [self._lane(i) for i in idx]
        else:
            lanes = [self._lane(b) for b in range(B)]
        keep = attention_mask.bool()
        L0 = keep.sum(-1).tolist()
        max_length = cfg.max_pos if max_new_tokens is None else Lp + int(max_new_tokens)            # :370-371 (padded length, as the reference)
        max_steps = min(max_length - Lp, int(max_length_times * Lp))                                # :420
        max_step_per_sample = [min(max_length - l, int(max_length_times * l)) for l in L0]          # :421
        groups, rb_of, loc, off = [], {}, {}, 0
        n_groups = -(-B // 4)                                                                       # row batches of <= 4 dialogues, sizes balanced
        for n in [B // n_groups + (1 if g < B % n_groups else 0) for g in range(n_groups)]:
            idxs = list(range(off, off + n))
            key = (n, off) if lanes[0] is self._lanes[0] else (n, off, "side")
            rb = self._rowbatch.get(key)
            if rb is None:
                rb = self._rowbatch[key] = RowBatch([lanes[b] for b in idxs], stream=self.engine.stream)
            rb.begin(max(L0[b] for b in idxs) + max(max_steps, 1) + 8, valid, cfg_scale)
            groups.append((rb, idxs))
            for b in idxs:
                rb_of[b], loc[b] = rb, b - off
            off += n
        per_list = forced_tokens is not None and len(forced_tokens) > 0 and isinstance(forced_tokens[0], (list, tuple))
        ftok = [(forced_tokens[b] if per_list else forced_tokens) for b in range(B)]
        nz = [(noise[b] if (noise is not None and noise.dim() == 3) else noise) for b in range(B)]
        x0s, off = [], 0
        with torch.cuda.stream(self.engine.stream):
            for b in range(B):
                x0 = self.engine.embed_ids(input_ids[b][keep[b]])
                if speech_input_mask is not None and conn_all is not None:
                    sp_b = speech_input_mask[b][keep[b]].bool()
                    n_b = int(sp_b.sum())
                    if n_b:
                        x0[sp_b.to(self.device)] = conn_all[off: off + n_b]                         # :221-224
                        off += n_b
                x0s.append(x0)
        seq = [input_ids[b][keep[b]].tolist() for b in range(B)]
        chunks = [[] for _ in range(B)]
        frame = [0] * B
        finished = [False] * B
        reach = [False] * B
        prev_tok = [None] * B
        pending = []
        ours = [False] * max(B, getattr(audio_streamer, "batch_size", B) if audio_streamer is not None else B)
        speculate = self.speculative_frames and self._use_graphs

        def deliver():
            if audio_streamer is None or not pending:
                pending.clear()
                return
            idx = [b for b, _ in pending]
            audio_streamer.put(torch.stack([lanes[b].take_chunk(k)[None] for b, k in pending]), torch.tensor(idx))   # one put per step, all samples (:644-653)
            pending.clear()

        def finish(b):
            finished[b] = True
            rb_of[b].set_active(loc[b], False)
            if audio_streamer is not None:
                deliver()
                ours[b] = True
                audio_streamer.end(torch.tensor([b]))

        for step in range(max_steps):
            if stop_check_fn is not None and stop_check_fn():                                       # :432-438
                if verbose:
                    print(f"Generation stopped externally at step {step + 1}")
                deliver()
                if audio_streamer is not None:
                    audio_streamer.end()
                break
            if audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and \
                    any(f and not ours[i] for i, f in enumerate(audio_streamer.finished_flags)):
                break                                                                               # :441-445, see _generate_lockstep
            if all(finished):
                break
            if Lp + step >= max_length:                                                             # :452-457
                for b in range(B):
                    reach[b] = reach[b] or not finished[b]
                break
            live = [b for b in range(B) if not finished[b]]
            forced = {b: (ftok[b][step] if (ftok[b] is not None and step < len(ftok[b])) else None) for b in live}
            toks = {}
            speculated = set()
            if step == 0:
                st_embed = self.engine.embed_ids(torch.tensor([ST]))
                for b in live:
                    rb_of[b].prefill(loc[b], x0s[b], chunk=getattr(self, "_prefill_chunk", 1024), neg_embed=st_embed)
                for b in live:
                    toks[b] = rb_of[b].first_token(loc[b], forced[b])
                    if toks[b] == SD:
                        rb_of[b].commit_negative(loc[b])
            else:
                # graph A of every row batch; a batch in its steady state (every live dialogue diffusing, noise injected) gets its diffusion
                # sampling enqueued speculatively behind it.  The conv tails follow once all A / H are queued: each batch's tails are enqueued
                # when ITS sampler has finished (RowBatch.speech_tails) and run while the main stream works on the next batch
                plan = []
                for rb, idxs in groups:
                    lv = [b for b in idxs if b in forced]
                    if not lv:
                        continue
                    spec = speculate and all(prev_tok[b] == SD and nz[b] is not None and frame[b] < len(nz[b]) for b in lv)
                    rb.decode_begin(ST, SD, {loc[b]: forced[b] for b in lv})
                    if spec:
                        rb.speech_begin([loc[b] for b in lv], {loc[b]: nz[b][frame[b]] for b in lv})
                    plan.append((rb, lv, spec))
                deliver()                  # the previous step's chunks (their copies completed long ago), before the host waits for a sampler
                for rb, lv, spec in plan:
                    if spec:
                        rb.speech_tails([loc[b] for b in lv])
                        speculated.update(lv)
                for rb, lv, spec in plan:
                    tk = rb.decode_end()
                    toks.update({b: tk[loc[b]] for b in lv})
            diffusing = []
            for b in live:
                tok = toks[b]
                rb = rb_of[b]
                if b in speculated and tok != SD:
                    rb.rollback(loc[b])
                prev_tok[b] = tok
                seq[b].append(tok)
                if tok == EOS:                                                                      # :517-526
                    if verbose:
                        print(f"Samples [{b}] reached EOS token at step {step + 1}.", flush=True)
                    finish(b)
                    continue
                if step >= max_step_per_sample[b]:                                                  # :528-537
                    reach[b] = True
                    finish(b)
                    continue
                if tok == SE:                                                                       # :540-544
                    rb.reset_speech(loc[b])
                if tok == SD:
                    diffusing.append(b)
                else:
                    rb.embed(loc[b])                                                                # :567
            todo = [b for b in diffusing if b not in speculated]
            if todo:
                need = [b for b in todo if nz[b] is None or frame[b] >= len(nz[b])]
                drawn = torch.randn(2 * len(need), cfg.latent)[: len(need)] if need else None     # the reference's draw for n diffusing samples (:699)
                rows = {b: (drawn[need.index(b)] if b in need else nz[b][frame[b]]) for b in todo}
                for rb, idxs in groups:
                    mine = [b for b in todo if rb_of[b] is rb]
                    if mine:
                        rb.speech([loc[b] for b in mine], {loc[b]: rows[b] for b in mine})
            for rb, _ in groups:
                rb.flush()                 # the conv tails are enqueued from worker threads: the chunk copies below must queue behind them
            for b in diffusing:                                                                     # :571-670
                with torch.cuda.stream(lanes[b].stream):
                    chunks[b].append(lanes[b].wav.clone())
                if audio_streamer is not None:
                    pending.append((b, lanes[b].stage_chunk()))
                frame[b] += 1
        deliver()
        for rb, _ in groups:
            rb.synchronize()
        if audio_streamer is not None:
            audio_streamer.end()
        pad_id = getattr(tokenizer, "pad_id", None)
        if pad_id is None:
            pad_id = special["eos"]
        rows = []
        for b in range(B):
            rows.append(torch.cat([input_ids[b][~keep[b]], torch.tensor(seq[b], dtype=torch.long)]))
        mx = max(r.shape[0] for r in rows)
        seq_t = torch.full((B, mx), int(pad_id), dtype=torch.long)
        for b, r in enumerate(rows):
            seq_t[b, : r.shape[0]] = r
        audios = [(torch.cat(c)[None] if c else None) for c in chunks]
        return VibeVoiceGenerationOutput(sequences=seq_t.to(in_dev), speech_outputs=audios if return_speech else None,
                                         reach_max_step_sample=torch.tensor(reach, dtype=torch.bool))

    def _generate_one(self, ids: torch.Tensor, sp_mask, conn, special, cfg_scale, max_new_tokens, max_length_times, forced_tokens,
                      noise, audio_streamer, stop_check_fn, sample_idx, verbose, sample_fn=None, sde_noise=None, refresh_negative=True):
        eng, cfg = self.engine, self.config

        def draw(frame):
            """The frame's random draws, in the reference's order: randn(2, latent) for the initial latent (:699), then - SDE solver
            only - one randn(2, latent) per solver step (dpm_solver.py:993-998); rows [1:] never reach the result."""
            nz = noise[frame] if noise is not None else torch.randn(2, cfg.latent)[0]
            sz = None
            if eng.sde:
                sz = sde_noise[frame] if sde_noise is not None else torch.randn(eng.n_steps, 2, cfg.latent)[:, 0]
            return nz, sz

        ST, SE, SD, EOS = special["speech_start"], special["speech_end"], special["speech_diffusion"], special["eos"]
        valid = [ST, SE, SD, EOS] + ([special["bos"]] if special.get("bos") is not None else [])
        # device-side position bookkeeping (vv_advance_lens): a negative "speech_start" id selects refresh_negative=False - the negative
        # row consumes every step's embedding and is never reset (:501-515); the batch-2 step computes that row anyway
        ST_dev = ST if refresh_negative else -1
        L0 = int(ids.shape[0])
        max_length = cfg.max_pos if max_new_tokens is None else L0 + int(max_new_tokens)          # :370-371
        max_steps = min(max_length - L0, int(max_length_times * L0))                            # :420
        eng.cfg_scale = float(cfg_scale)
        eng.begin_sequence(L0 + max(max_steps, 1) + 8, valid)
        x0 = eng.embed_ids(ids)
        if conn is not None:
            with torch.cuda.stream(eng.stream):
                x0[sp_mask.to(self.device)] = conn                                              # :221-224
        seq = ids.tolist()
        chunks: List[torch.Tensor] = []
        reach_max = False
        frame = 0
        stream_idx = torch.tensor([sample_idx])
        # speculative frame launch needs the greedy / forced token path (a sampled token needs the logits on the host first)
        speculate = self.speculative_frames and sample_fn is None and eng.use_graphs
        prev_tok, pending_nz = None, None
        staged: List[int] = []          # ring slots whose audio has not been handed to the streamer yet (at most 2)

        def deliver():
            """Hand finished frames to the streamer.  Called right after the next step's launches are enqueued: the frame's copy
            completes before that step's token does, so waiting for it here delays nothing on the GPU."""
            while staged:
                audio_streamer.put(eng.take_chunk(staged.pop(0))[None, None], stream_idx)

        hook = deliver if audio_streamer is not None else None
        for step in range(max_steps):
            if stop_check_fn is not None and stop_check_fn():                                  # :432-438
                if verbose:
                    print(f"Generation stopped externally at step {step + 1}")
                if audio_streamer is not None:
                    deliver()
                    audio_streamer.end()
                break
            if audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and audio_streamer.finished_flags[sample_idx]:
                break                                                                           # :441-445 (this sample's stream was ended externally)
            if len(seq) >= max_length:                                                          # :452-457
                reach_max = True
                break
            forced = forced_tokens[step] if (forced_tokens is not None and step < len(forced_tokens)) else None
            speculated = False
            if step == 0:
                # the negative branch's prompt, a single speech_start (:377-381), is one more row of the prompt prefill (cache row 1, position 0)
                eng.prefill(x0, row=0, pos0=0, chunk=getattr(self, "_prefill_chunk", 1024), neg_embed=eng.embed_ids(torch.tensor([ST])))
                tok = eng.first_token(ST_dev, SD, forced, sample_fn)
                if tok == SD or not refresh_negative:
                    eng.commit_negative_prompt()         # the branch is in use from step 0 on (otherwise the row is overwritten by the next speech_start)
            elif speculate and prev_tok == SD and (pending_nz is not None or ((noise is None or frame < len(noise)) and
                                                                            (sde_noise is None or frame < len(sde_noise)))):
                # steady state of a dialogue: the frame's diffusion tail is enqueued right behind the LLM step, the host waits
                # for the token only.  The noise row is the draw the reference makes when the token IS speech_diffusion; a draw
                # made for a mis-speculated frame is kept for the next real one (same RNG sequence).
                if pending_nz is None:
                    pending_nz = draw(frame)
                tok = eng.step_decode_speculative(ST_dev, SD, forced, *pending_nz, on_enqueued=hook, stage=audio_streamer is not None)
                speculated = True
                if tok != SD:
                    eng.rollback_speech_state()
            else:
                tok = eng.step_decode(ST_dev, SD, forced, sample_fn, on_enqueued=hook)          # :478-496 (+ speculative :581-583)
            prev_tok = tok
            seq.append(tok)
            if tok == EOS:                                                                      # :517-526
                if verbose:
                    print(f"Samples [{sample_idx}] reached EOS token at step {step + 1}.", flush=True)
                if audio_streamer is not None:
                    deliver()
                    audio_streamer.end(stream_idx)
                break
            if tok == SE:                                                                       # :540-544
                with torch.cuda.stream(eng.stream):
                    eng.reset_speech_caches()
            if tok == SD:                                                                       # :571-670
                slot = eng.spec_slot if speculated else None
                if not speculated:
                    if pending_nz is None:
                        pending_nz = draw(frame)
                    slot = eng.step_speech(*pending_nz, stage=audio_streamer is not None)
                pending_nz = None
                with torch.cuda.stream(eng.stream):
                    chunk = eng.wav.clone()
                chunks.append(chunk)
                if audio_streamer is not None:
                    staged.append(slot if slot is not None else eng.stage_chunk())     # the copy went out right behind the acoustic decoder
                    if len(staged) > 2:
                        deliver()
                frame += 1
            else:
                eng.step_embed()                                                                # :567
        if audio_streamer is not None:
            deliver()
        eng.stream.synchronize()
        audio = torch.cat(chunks)[None] if chunks else None
        return dict(sequence=torch.tensor(seq, dtype=torch.long), audio=audio, reach_max=reach_max)
