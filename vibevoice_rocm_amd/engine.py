"""Per-GPU VibeVoice engine: device state + the launch sequences of the per-frame loop on the HIP C ABI.

One Engine = one utterance stream on one MI355X (dialogues shard across GPUs, batch 1 per GPU).
Everything numerical goes through libvv_hip.so; torch only owns device memory, the stream and pinned staging.

Per generated token (reference loop: modeling_vibevoice_inference.py:430-673):
  graph A  positive AND negative Qwen2 decode as ONE batch-2 weight pass (the negative row is speculative: it is
           committed only if the chosen token is speech_diffusion, exactly when the reference runs it, :575-587)
           -> 4/5 constrained logits -> argmax -> device-side position bookkeeping
  host     reads the 4-byte token (the only sync of the frame) and runs the reference's token state machine
  graph B  (speech_diffusion) CFG diffusion sampling -> acoustic decode -> semantic encode -> connectors -> next embedding
  graph C  (otherwise) next embedding = embed_tokens[token]
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from .config import VVConfig
from .schedule import DPMSolverMultistepScheduler, timestep_sinusoid
from .weights import DeviceWeights


# HIP streams an Engine gave back, per device.  Streams are recycled, never just dropped: every stream that has run work keeps a hardware
# queue, and on MI355X a process with more than ~5 queues that have ever been active runs concurrent lanes 2-3x slower (measured:
# batch-of-4 generate 69 -> 30 audio-s/s once three more streams had been used, tools/dbg_lanes.py), so the number of streams has to
# stay at the number of engines alive at once.
_IDLE_STREAMS: Dict[str, list] = {}


class Engine:
    def __init__(self, cfg: VVConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0", dtype=torch.bfloat16,
                 kv_dtype: Optional[torch.dtype] = None, use_graphs: bool = True, bf16_timestep_quirk: Optional[bool] = None,
                 weight_quant: Optional[str] = None, stream: Optional[torch.cuda.Stream] = None, weights_from: Optional["Engine"] = None):
        self.lib = L.load()
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.VVError("the VibeVoice MI355X engine needs a GPU device (there is no CPU path)")
        self.dtype = dtype
        self.kv_dtype = kv_dtype or dtype
        self.use_graphs = use_graphs
        self.bf16_t_quirk = (dtype == torch.bfloat16) if bf16_timestep_quirk is None else bf16_timestep_quirk
        torch.cuda.set_device(self.device)
        self._own_stream = stream is None
        if stream is None:                               # `stream`: run on another engine's stream (lanes of a large lock-step batch)
            idle = _IDLE_STREAMS.setdefault(str(self.device), [])
            stream = idle.pop() if idle else torch.cuda.Stream(self.device)
        self.stream = stream
        torch.zeros(1, device=self.device)              # make sure the HIP context exists before the library touches it
        L.check(self.lib.vv_init(), "vv_init")          # one-time kernel attributes, before any graph capture
        self.sync_in()
        with torch.cuda.stream(self.stream):
            # weights_from: another engine of the same model on this device - its weights are used as they are (DeviceWeights.fork), only
            # the streaming state is this engine's own
            self.w = weights_from.w.fork() if weights_from is not None else DeviceWeights(cfg, state_dict, self.device, dtype, quant=weight_quant)
            H = cfg.hidden
            f32 = dict(dtype=torch.float32, device=self.device)
            self.x2 = torch.zeros(2, H, **f32)            # input embedding of the step, rows {positive, negative}
            self.hidden2 = torch.zeros(2, H, **f32)       # final-normed hidden states = {condition, negative condition}
            self.lens = torch.zeros(2, dtype=torch.int32, device=self.device)     # positions {positive, negative}
            self.frame_ctr = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.token_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.forced_dev = torch.full((1,), -1, dtype=torch.int32, device=self.device)
            self.noise_dev = torch.zeros(cfg.latent, **f32)
            self.latent = torch.zeros(cfg.latent, **f32)
            self.wav = torch.zeros(cfg.hop, **f32)
            self.sem = torch.zeros(cfg.sem_dim, **f32)
            self.conn_ws = torch.zeros(8 * H, **f32)
            self.logits = torch.zeros(8, **f32)
        self.token_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.logits_host = torch.zeros(8, dtype=torch.float32).pin_memory()
        self.forced_host = torch.full((1,), -1, dtype=torch.int32).pin_memory()
        self._forced_dev_val = -1                         # what forced_dev holds (see _set_forced)
        self.noise_host = torch.zeros(2, cfg.latent, dtype=torch.float32).pin_memory()   # double-buffered: the host runs a frame ahead
        self._noise_k = 0
        self._tok_event = torch.cuda.Event()
        # streaming delivery: a frame's 3200 samples go device -> pinned ring asynchronously, an event per
        # slot says when the host may hand them to the AudioStreamer; generate() never blocks the launch queue on a D2H copy
        self._ring = [torch.zeros(cfg.hop, dtype=torch.float32).pin_memory() for _ in range(8)]     # 8 slots: <= 3 chunks wait for delivery, a
        self._ring_ev = [torch.cuda.Event() for _ in range(8)]                                       # mis-speculated frame burns one
        self._ring_n = 0
        self.spec_slot = None
        with torch.cuda.stream(self.stream):
            self._state_snap = torch.empty_like(self.w.state_blob())
        self.scheduler = DPMSolverMultistepScheduler(num_train_timesteps=cfg.ddpm_steps, beta_schedule=cfg.beta_schedule,
                                                     prediction_type=cfg.prediction_type)
        self.n_steps = 0
        self.cfg_scale = 1.3
        self.kv = None
        self._kv_t = None
        self._llm_ws = None
        self._llm_ws_rows = 0
        self._graphs: Dict[str, int] = {}
        self._graph_key = None
        self.valid_ids: List[int] = []
        self._w_valid = None
        self._ids_dev = None
        with torch.cuda.stream(self.stream):
            self._alloc_ws()
        self.set_steps(cfg.ddpm_infer)
        self.stream.synchronize()

    # ---------------------------------------------------------------------------------------------------------
    @property
    def sp(self) -> int:
        return self.stream.cuda_stream

    def _ck(self, rc, what):
        L.check(rc, what)

    def sync_in(self):
        """Order the engine stream after whatever the caller enqueued on torch's current stream (inputs, weights)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _alloc_ws(self):
        lib = self.lib
        u8 = dict(dtype=torch.uint8, device=self.device)
        self._dec_ws = torch.empty(lib.vv_convnet_ws_bytes(C.byref(self.w.dec), 1, 1), **u8)
        self._sem_ws = torch.empty(lib.vv_convnet_ws_bytes(C.byref(self.w.sem), self.cfg.hop, 0), **u8)
        self._ensure_llm_ws(2)

    def _ensure_llm_ws(self, rows: int):
        if rows > self._llm_ws_rows:
            n = self.lib.vv_llm_ws_bytes(C.byref(self.w.llm), rows)
            self._llm_ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._llm_ws_rows = rows
            self._drop_graphs()

    def _drop_graphs(self):
        for g in self._graphs.values():
            self.lib.vv_graph_destroy(g)
        self._graphs = {}

    # ---------------------------------------------------------------------------------------------------------
    def set_steps(self, n_steps: int):
        """set_ddpm_inference_steps: schedule tables + the step-invariant t_embedder(t_i) table [n, D]."""
        algo = self.scheduler.config["algorithm_type"]
        if n_steps == self.n_steps and algo == getattr(self, "_algo", None):
            return
        self.n_steps, self._algo = n_steps, algo
        self.sde = algo.startswith("sde")            # variance noise per solver step (dpm_solver.py:993-998)
        self.scheduler.set_timesteps(n_steps)
        coefs = (L.DpmCoef * n_steps)()
        for i, c in enumerate(self.scheduler.coefs):
            coefs[i].alpha_s, coefs[i].sigma_s, coefs[i].cx, coefs[i].cd = c["alpha_s"], c["sigma_s"], c["cx"], c["cd"]
            coefs[i].rinv, coefs[i].order, coefs[i].cn = c["rinv"], c["order"], c["cn"]
        self._coefs = coefs
        D = self.cfg.head_hidden
        with torch.cuda.stream(self.stream):
            sin = timestep_sinusoid(self.scheduler.timesteps.numpy(), self.w.t_mlp0.shape[1], bf16_quirk=self.bf16_t_quirk).to(self.device)
            t1 = torch.empty(n_steps, D, dtype=torch.float32, device=self.device)
            self.temb = torch.empty(n_steps, D, dtype=torch.float32, device=self.device)
            self.linear(sin, self.w.t_mlp0, t1)
            self.linear(t1, self.w.t_mlp2, self.temb, pro=L.PRO_SILU)
            self._head_ws = torch.empty(self.lib.vv_head_ws_bytes(C.byref(self.w.head), n_steps), dtype=torch.uint8, device=self.device)
            self.sde_noise_dev = torch.zeros(n_steps, self.cfg.latent, dtype=torch.float32, device=self.device)
        self.sde_noise_host = torch.zeros(2, n_steps, self.cfg.latent, dtype=torch.float32).pin_memory()
        self._drop_graphs()

    def linear(self, x, w, out, pro=L.PRO_NONE, bias=None):
        """thin vv_linear wrapper for [m,k] x [n,k]^T (host-side table building and tests)."""
        a = L.LinArgs()
        a.x, a.ldx, a.m = x.data_ptr(), x.stride(0), x.shape[0]
        a.pro = pro
        a.w, a.n, a.k = w.data_ptr(), w.shape[0], w.shape[1]
        a.wdt = L.VV_F32 if w.dtype == torch.float32 else L.VV_BF16
        a.bias = L.ptr(bias)
        a.out, a.ldo = out.data_ptr(), out.stride(0)
        self._ck(self.lib.vv_linear(C.byref(a), self.sp), "vv_linear")

    # ---------------------------------------------------------------------------------------------------------
    def begin_sequence(self, s_max: int, valid_ids: List[int]):
        """Fresh utterance: KV cache sized for s_max tokens, conv states zeroed, constrained-vocabulary rows gathered."""
        cfg = self.cfg
        s_max = (int(s_max) + 63) // 64 * 64
        with torch.cuda.stream(self.stream):
            if self.kv is None or self.kv.s_max < s_max:
                shape = (cfg.layers, 2, cfg.kv_heads, s_max, cfg.head_dim)
                self._kv_t = (torch.zeros(shape, dtype=self.kv_dtype, device=self.device),
                              torch.zeros(shape, dtype=self.kv_dtype, device=self.device))
                kv = L.KV()
                kv.k, kv.v = self._kv_t[0].data_ptr(), self._kv_t[1].data_ptr()
                if self.kv_dtype == torch.bfloat16 and cfg.head_dim == 128:
                    # transposed value cache in 32-key tiles [.., s_max / 32, head_dim, 32] for the matrix-core attention kernels (kept in step
                    # with v by vv_rope_store for prompt rows and by vv_attn_decode for decode steps)
                    self._kv_vt = torch.empty((cfg.layers, 2, cfg.kv_heads, s_max // 32, cfg.head_dim, 32), dtype=self.kv_dtype, device=self.device)
                    kv.vt = self._kv_vt.data_ptr()
                kv.kvdt = L.VV_F32 if self.kv_dtype == torch.float32 else L.VV_BF16
                kv.layers, kv.rows, kv.kv_heads, kv.s_max, kv.head_dim = cfg.layers, 2, cfg.kv_heads, s_max, cfg.head_dim
                self.kv = kv
                self._drop_graphs()
            ids = sorted(set(int(i) for i in valid_ids))
            if ids != self.valid_ids:
                self.valid_ids = ids
                arr = (C.c_int * len(ids))(*ids)
                self._w_valid = torch.empty(len(ids), cfg.hidden, dtype=self.dtype, device=self.device)
                self._ck(self.lib.vv_gather_rows(self.w.lm_head.data_ptr(), self.w.wdt, cfg.hidden, arr, len(ids),
                                                 self._w_valid.data_ptr(), self.sp), "vv_gather_rows")
                self._ids_dev = torch.tensor(ids, dtype=torch.int32, device=self.device)
                self._drop_graphs()
            self.lens.zero_()
            self.frame_ctr.zero_()
            self.reset_speech_caches()

    def reset_speech_caches(self):
        """acoustic_cache.set_to_zero / semantic_cache.set_to_zero (modeling_vibevoice_inference.py:540-544)."""
        self._ck(self.lib.vv_convnet_reset(C.byref(self.w.dec), self.sp), "vv_convnet_reset")
        self._ck(self.lib.vv_convnet_reset(C.byref(self.w.sem), self.sp), "vv_convnet_reset")

    # ---------------------------------------------------------------------------------------------------------
    # component launch sequences (all asynchronous on self.stream)
    # ---------------------------------------------------------------------------------------------------------
    def llm_forward(self, x: torch.Tensor, lens: torch.Tensor, cache_rows: Optional[torch.Tensor], out: torch.Tensor):
        R = x.shape[0]
        self._ensure_llm_ws(R)
        self._ck(self.lib.vv_llm_forward(C.byref(self.w.llm), C.byref(self.kv), x.data_ptr(), x.stride(0), R, lens.data_ptr(),
                                         L.ptr(cache_rows), out.data_ptr(), out.stride(0), self._llm_ws.data_ptr(), self.sp),
                 "vv_llm_forward")

    def prefill(self, embeds: torch.Tensor, row: int = 0, pos0: int = 0, chunk: int = 1024, neg_embed: Optional[torch.Tensor] = None) -> None:
        """Prompt prefill on cache row `row`: embeds [L0, H] fp32 -> self.hidden2[row] = last hidden state; lens[row] = pos0+L0.
        Prompts longer than `chunk` rows run as ceil(L0 / chunk) EQUAL chunks (rounded up to 32 rows): a short trailing chunk would stream
        every weight matrix once more for a handful of rows (1 040 tokens as 1 024 + 16 cost 18.5 ms, as 2 x 520 they cost 12).
        neg_embed [1, H]: the negative branch's one-token prompt (a single speech_start, modeling_vibevoice_inference.py:377-381) rides along as
        one more row of the last chunk - cache row 1, position 0 - instead of a weight pass of its own (1.1 ms of the first chunk's latency);
        hidden2[1] then holds its state and the CALLER sets lens[1] = 1 if the branch is in use (`commit_negative_prompt`)."""
        L0 = embeds.shape[0]
        n_chunks = max(1, -(-L0 // max(1, chunk)))
        size = -(-L0 // n_chunks)
        size = min(chunk, (size + 31) // 32 * 32) if n_chunks > 1 else L0
        with torch.cuda.stream(self.stream):
            for c0 in range(0, L0, size):
                c1 = min(L0, c0 + size)
                n = c1 - c0
                last = c1 == L0 and neg_embed is not None
                lens = torch.arange(pos0 + c0, pos0 + c1 + (1 if last else 0), dtype=torch.int32, device=self.device)
                rows = torch.full((n + (1 if last else 0),), row, dtype=torch.int32, device=self.device)
                x = embeds[c0:c1].contiguous()
                if last:
                    lens[-1] = 0
                    rows[-1] = 1
                    x = torch.cat([x, neg_embed.to(x.dtype).reshape(1, -1)])
                out = torch.empty(x.shape[0], self.cfg.hidden, dtype=torch.float32, device=self.device)
                self.llm_forward(x, lens, rows, out)
            if neg_embed is not None:
                self.hidden2[row].copy_(out[-2])
                self.hidden2[1].copy_(out[-1])
            else:
                self.hidden2[row].copy_(out[-1])
            self.lens[row] = pos0 + L0

    def commit_negative_prompt(self):
        """the negative branch consumed its one-token prompt (see prefill, neg_embed)"""
        with torch.cuda.stream(self.stream):
            self.lens[1] = 1

    def _logits(self):
        a = L.LinArgs()
        nv = len(self.valid_ids)
        a.x, a.ldx, a.m = self.hidden2.data_ptr(), self.cfg.hidden, 1
        a.w, a.n, a.k, a.wdt = self._w_valid.data_ptr(), nv, self.cfg.hidden, self.w.wdt
        a.out, a.ldo = self.logits.data_ptr(), nv
        self._ck(self.lib.vv_linear(C.byref(a), self.sp), "lm_head")

    def _select_token(self):
        self._logits()
        self._pick()

    def _pick(self):
        nv = len(self.valid_ids)
        self._ck(self.lib.vv_argmax_ids(self.logits.data_ptr(), nv, self._ids_dev.data_ptr(), self.token_dev.data_ptr(),
                                        self.forced_dev.data_ptr(), self.sp), "vv_argmax_ids")

    def _seq_A(self, tok_start, tok_diff):
        """batch-2 decode step + token selection + device-side position bookkeeping: the Qwen2 stack leaves its un-normalised last
        hidden rows at the start of the workspace, vv_llm_tail does final norm + constrained logits + argmax / forced token +
        position update in ONE launch."""
        nv = len(self.valid_ids)
        self._ck(self.lib.vv_llm_forward(C.byref(self.w.llm), C.byref(self.kv), self.x2.data_ptr(), self.cfg.hidden, 2,
                                         self.lens.data_ptr(), None, None, 0, self._llm_ws.data_ptr(), self.sp), "vv_llm_forward")
        self._ck(self.lib.vv_llm_tail(C.byref(self.w.llm), self._llm_ws.data_ptr(), self.cfg.hidden, 2, self.hidden2.data_ptr(), self.cfg.hidden,
                                      self._w_valid.data_ptr(), nv, self._ids_dev.data_ptr(), self.logits.data_ptr(), self.token_dev.data_ptr(),
                                      self.forced_dev.data_ptr(), self.lens.data_ptr(), tok_start, tok_diff, self.frame_ctr.data_ptr(), self.sp),
                 "vv_llm_tail")

    def _seq_A1(self):
        self._ck(self.lib.vv_llm_forward(C.byref(self.w.llm), C.byref(self.kv), self.x2.data_ptr(), self.cfg.hidden, 2,
                                         self.lens.data_ptr(), None, self.hidden2.data_ptr(), self.cfg.hidden,
                                         self._llm_ws.data_ptr(), self.sp), "vv_llm_forward")
        self._logits()

    def _seq_A2(self, tok_start, tok_diff):
        self._pick()
        self._ck(self.lib.vv_advance_lens(self.lens.data_ptr(), self.token_dev.data_ptr(), tok_start, tok_diff,
                                          self.frame_ctr.data_ptr(), self.sp), "vv_advance_lens")

    def _seq_B(self, cfg_scale):
        """speech_diffusion tail: sample latent, decode audio, re-encode semantics, build the next embedding."""
        lib, w, cfg = self.lib, self.w, self.cfg
        # snapshot of the streaming state (one ~1.4 MB copy): lets a frame that was launched speculatively be rolled back
        blob = w.state_blob()
        self._ck(lib.vv_copy_rows(blob.data_ptr(), blob.numel(), self._state_snap.data_ptr(), blob.numel(), 1, blob.numel(), self.sp), "snapshot")
        self._ck(lib.vv_head_sample(C.byref(w.head), self.hidden2.data_ptr(), cfg.hidden, self.noise_dev.data_ptr(),
                                    self.temb.data_ptr(), self._coefs, self.n_steps, cfg_scale, self.latent.data_ptr(),
                                    self._head_ws.data_ptr(), self.sde_noise_dev.data_ptr() if self.sde else None, self.sp), "vv_head_sample")
        self._ck(lib.vv_decoder_forward(C.byref(w.dec), self.latent.data_ptr(), 1, 1.0 / w.speech_scale, -w.speech_bias,
                                        self.wav.data_ptr(), self._dec_ws.data_ptr(), self.sp), "vv_decoder_forward")
        self._ck(lib.vv_encoder_forward(C.byref(w.sem), self.wav.data_ptr(), cfg.hop, self.sem.data_ptr(),
                                        self._sem_ws.data_ptr(), self.sp), "vv_encoder_forward")
        self._ck(lib.vv_connector_pair(C.byref(w.ac_conn), C.byref(w.sem_conn), self.latent.data_ptr(), self.sem.data_ptr(), self.x2.data_ptr(), cfg.hidden, 2,
                                       self.conn_ws.data_ptr(), self.sp), "connectors")

    def _seq_B1(self, cfg_scale):
        """first half of _seq_B, up to the frame's audio (streaming delivery: the chunk's D2H copy goes out between the halves)"""
        lib, w, cfg = self.lib, self.w, self.cfg
        blob = w.state_blob()
        self._ck(lib.vv_copy_rows(blob.data_ptr(), blob.numel(), self._state_snap.data_ptr(), blob.numel(), 1, blob.numel(), self.sp), "snapshot")
        self._ck(lib.vv_head_sample(C.byref(w.head), self.hidden2.data_ptr(), cfg.hidden, self.noise_dev.data_ptr(),
                                    self.temb.data_ptr(), self._coefs, self.n_steps, cfg_scale, self.latent.data_ptr(),
                                    self._head_ws.data_ptr(), self.sde_noise_dev.data_ptr() if self.sde else None, self.sp), "vv_head_sample")
        self._ck(lib.vv_decoder_forward(C.byref(w.dec), self.latent.data_ptr(), 1, 1.0 / w.speech_scale, -w.speech_bias,
                                        self.wav.data_ptr(), self._dec_ws.data_ptr(), self.sp), "vv_decoder_forward")

    def _seq_B2(self):
        """second half of _seq_B: semantic re-encode of the frame and the next step's input embedding"""
        lib, w, cfg = self.lib, self.w, self.cfg
        self._ck(lib.vv_encoder_forward(C.byref(w.sem), self.wav.data_ptr(), cfg.hop, self.sem.data_ptr(),
                                        self._sem_ws.data_ptr(), self.sp), "vv_encoder_forward")
        self._ck(lib.vv_connector_pair(C.byref(w.ac_conn), C.byref(w.sem_conn), self.latent.data_ptr(), self.sem.data_ptr(), self.x2.data_ptr(), cfg.hidden, 2,
                                       self.conn_ws.data_ptr(), self.sp), "connectors")

    def _speech(self, cfg_scale: float, stage: bool) -> Optional[int]:
        """Phase B on the current stream.  stage: a consumer is waiting for the audio (AudioStreamer) - the frame runs as two graphs and the
        chunk's device -> pinned copy is enqueued BETWEEN them, as soon as the acoustic decoder has produced it: the chunk reaches the host
        ~0.5 ms earlier than behind the semantic encoder and the connectors (which only the NEXT step needs).  Returns the ring slot."""
        if not stage:
            self._run("B", self._seq_B, cfg_scale)
            return None
        self._run("B1", self._seq_B1, cfg_scale)
        k = self._ring_n % len(self._ring)
        self._ring_n += 1
        self._ring[k].copy_(self.wav, non_blocking=True)
        self._ring_ev[k].record(self.stream)
        self._run("B2", self._seq_B2)
        return k

    def _seq_C(self):
        """next embedding = embed_tokens[token] for both rows (modeling_vibevoice_inference.py:567)."""
        cfg = self.cfg
        self._ck(self.lib.vv_embed_row(self.w.embed.data_ptr(), self.w.wdt, cfg.hidden, self.token_dev.data_ptr(), self.x2.data_ptr(), self.sp), "embed")
        self._ck(self.lib.vv_copy_rows(self.x2.data_ptr(), 0, self.x2.data_ptr() + 4 * cfg.hidden, cfg.hidden, 1, cfg.hidden, self.sp), "copy")

    def _run(self, name: str, fn, *args):
        """Run a launch sequence, through a cached hipGraph when enabled."""
        if not self.use_graphs:
            fn(*args)
            return
        key = (name,) + tuple(args)
        g = self._graphs.get(key)
        if g is None:
            self.stream.synchronize()
            self._ck(self.lib.vv_graph_begin(self.sp), "graph begin")
            try:
                fn(*args)
            finally:
                ge = C.c_void_p()
                rc = self.lib.vv_graph_end(self.sp, C.byref(ge))
            self._ck(rc, "graph end")
            g = ge.value
            self._graphs[key] = g
        self._ck(self.lib.vv_graph_launch(g, self.sp), "graph launch")

    # ---------------------------------------------------------------------------------------------------------
    # the three per-token phases used by generate()
    # ---------------------------------------------------------------------------------------------------------
    def _set_forced(self, forced: Optional[int]):
        """Device-side forced token (-1: none).  Uploaded only when it changes: in the steady state of a dialogue (greedy decoding, or a forced
        schedule of speech_diffusion frames) the value repeats and the 4-byte H2D blit (~5 us on the frame's critical path) is skipped.  The
        pinned word is rewritten only after the previous step's token has been awaited, i.e. after any earlier copy from it has executed."""
        v = -1 if forced is None else int(forced)
        if v == self._forced_dev_val:
            return
        self.forced_host[0] = v
        self.forced_dev.copy_(self.forced_host, non_blocking=True)
        self._forced_dev_val = v

    def _host_logits(self) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            self.logits_host.copy_(self.logits, non_blocking=True)
        self.stream.synchronize()
        return self.logits_host[: len(self.valid_ids)].clone()

    def step_decode(self, tok_start: int, tok_diff: int, forced: Optional[int] = None, sample_fn=None, on_enqueued=None) -> int:
        """Phase A + the frame's only host sync: returns the chosen token.  With `sample_fn(logits, ids) -> token` (do_sample)
        the constrained logits are read back first and the sampled token is fed to the device-side bookkeeping."""
        if sample_fn is not None and forced is None:
            with torch.cuda.stream(self.stream):
                self._run("A1", self._seq_A1)
            tok = int(sample_fn(self._host_logits(), self.valid_ids))
            with torch.cuda.stream(self.stream):
                self._set_forced(tok)
                self._run("A2", self._seq_A2, int(tok_start), int(tok_diff))
            self.stream.synchronize()
            return tok
        with torch.cuda.stream(self.stream):
            self._set_forced(forced)
            self._run("A", self._seq_A, int(tok_start), int(tok_diff))
            self.token_host.copy_(self.token_dev, non_blocking=True)
        if on_enqueued is not None:
            on_enqueued()
        self.stream.synchronize()
        return int(self.token_host[0])

    def first_token(self, tok_start: int, tok_diff: int, forced: Optional[int] = None, sample_fn=None) -> int:
        """Token selection right after prefill (hidden2[0] already holds the last prompt state)."""
        if sample_fn is not None and forced is None:
            with torch.cuda.stream(self.stream):
                self._logits()
            forced = int(sample_fn(self._host_logits(), self.valid_ids))
        with torch.cuda.stream(self.stream):
            self._set_forced(forced)
            self._select_token()
            self.token_host.copy_(self.token_dev, non_blocking=True)
        self.stream.synchronize()
        return int(self.token_host[0])

    def _upload_noise(self, noise: torch.Tensor, sde_noise: Optional[torch.Tensor]):
        """Host staging (double-buffered pinned rows: the host runs up to a frame ahead) + async copies on the engine stream."""
        self._noise_k ^= 1
        nh = self.noise_host[self._noise_k]
        nh.copy_(noise.reshape(-1)[: self.cfg.latent])
        self.noise_dev.copy_(nh, non_blocking=True)
        if self.sde:
            if sde_noise is None:
                raise L.VVError("the SDE solver needs the per-step variance noise [n_steps, latent]")
            sh = self.sde_noise_host[self._noise_k]
            sh.copy_(sde_noise.reshape(self.n_steps, -1)[:, : self.cfg.latent])
            self.sde_noise_dev.copy_(sh, non_blocking=True)

    def step_speech(self, noise: torch.Tensor, sde_noise: Optional[torch.Tensor] = None, stage: bool = False) -> Optional[int]:
        """Phase B.  `noise` is the CPU fp32 [latent] row the reference would have drawn (modeling_vibevoice_inference.py:699),
        `sde_noise` [n_steps, latent] the variance noise of the SDE solver's steps (dpm_solver.py:993-998), if that solver is set.
        stage: also enqueue the chunk's copy to the pinned ring as soon as the decoder is done (see _speech); returns the slot."""
        with torch.cuda.stream(self.stream):
            self._upload_noise(noise, sde_noise)
            return self._speech(float(self.cfg_scale), stage)

    def step_decode_speculative(self, tok_start: int, tok_diff: int, forced: Optional[int], noise: torch.Tensor,
                                sde_noise: Optional[torch.Tensor] = None, on_enqueued=None, stage: bool = False) -> int:
        """Phase A, then phase B enqueued right behind it ON THE ASSUMPTION that the token is speech_diffusion (the steady state
        of a dialogue), then one host wait on the token alone.  The GPU therefore never idles between A and B while the host
        wakes up and decides; if the token turns out to be something else the caller rolls the speech state back
        (`rollback_speech_state`) - phase B touches nothing else that survives (x2 is rewritten by the embed phase)."""
        with torch.cuda.stream(self.stream):
            self._set_forced(forced)
            self._run("A", self._seq_A, int(tok_start), int(tok_diff))
            self.token_host.copy_(self.token_dev, non_blocking=True)
            self._tok_event.record(self.stream)
            self._upload_noise(noise, sde_noise)
            self.spec_slot = self._speech(float(self.cfg_scale), stage)     # ring slot of the speculative frame's chunk (stage), or None
        if on_enqueued is not None:
            on_enqueued()          # e.g. hand the previous frame's audio to the streamer: it completes before this step's token
        self._tok_event.synchronize()
        return int(self.token_host[0])

    def __del__(self):
        try:
            if self._own_stream:
                self.stream.synchronize()
                _IDLE_STREAMS.setdefault(str(self.device), []).append(self.stream)
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    def decode_begin(self, tok_start: int, tok_diff: int, forced: Optional[int], spec_noise=None):
        """Lock-step batches (one Engine per sample, one host loop): enqueue phase A - and, with `spec_noise` = (noise, sde_noise), the
        speculative phase B behind it - without waiting; `decode_end` returns the token."""
        with torch.cuda.stream(self.stream):
            self._set_forced(forced)
            self._run("A", self._seq_A, int(tok_start), int(tok_diff))
            self.token_host.copy_(self.token_dev, non_blocking=True)
            self._tok_event.record(self.stream)
            if spec_noise is not None:
                self._upload_noise(*spec_noise)
                self._run("B", self._seq_B, float(self.cfg_scale))

    def decode_end(self) -> int:
        self._tok_event.synchronize()
        return int(self.token_host[0])

    def stage_chunk(self) -> int:
        """Enqueue the async D2H copy of the frame just generated into the next ring slot; returns the slot."""
        k = self._ring_n % len(self._ring)
        self._ring_n += 1
        with torch.cuda.stream(self.stream):
            self._ring[k].copy_(self.wav, non_blocking=True)
            self._ring_ev[k].record(self.stream)
        return k

    def take_chunk(self, k: int) -> torch.Tensor:
        """Wait for slot k's copy and return its samples (a fresh CPU tensor: the slot is reused seven frames later)."""
        self._ring_ev[k].synchronize()
        return self._ring[k].clone()

    def rollback_speech_state(self):
        """Undo the streaming-state updates of the last phase B (a mis-speculated frame)."""
        blob = self.w.state_blob()
        with torch.cuda.stream(self.stream):
            self._ck(self.lib.vv_copy_rows(self._state_snap.data_ptr(), blob.numel(), blob.data_ptr(), blob.numel(), 1, blob.numel(), self.sp), "rollback")

    def step_embed(self):
        with torch.cuda.stream(self.stream):
            self._run("C", self._seq_C)

    # ---------------------------------------------------------------------------------------------------------
    # voice-prompt path (runs once per utterance)
    # ---------------------------------------------------------------------------------------------------------
    def acoustic_encode(self, wav: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """Whole-utterance (non-streaming) acoustic encoder: wav [T] -> mean latents [ceil(T/hop), vae_dim].  `stream`: run on another HIP
        stream (the encoder of a voice prompt is stateless: the voices of a dialogue encode concurrently, see acoustic_encode_many)."""
        T = wav.shape[0]
        F = (T + self.cfg.hop - 1) // self.cfg.hop
        st = stream or self.stream
        with torch.cuda.stream(st):
            x = wav.to(device=self.device, dtype=torch.float32).contiguous()
            ws = torch.empty(self.lib.vv_convnet_ws_bytes(C.byref(self.w.ac_enc), T, 0), dtype=torch.uint8, device=self.device)
            out = torch.empty(F, self.cfg.ac_dim, dtype=torch.float32, device=self.device)
            self._ck(self.lib.vv_encoder_forward(C.byref(self.w.ac_enc), x.data_ptr(), T, out.data_ptr(), ws.data_ptr(), st.cuda_stream),
                     "vv_encoder_forward")
        return out

    def acoustic_encode_many(self, wavs: List[torch.Tensor], max_streams: int = 4, side_streams: Optional[list] = None) -> List[torch.Tensor]:
        """The S voice prompts of a dialogue (modeling_vibevoice_inference.py:149-163 encodes them as one padded batch).  Each voice is an
        independent causal sequence with its own zero left context and its own ragged tail, so they are not stacked into the row
        dimension of one launch sequence (every conv / mixer would need per-segment halos); instead they run CONCURRENTLY on up to
        `max_streams` HIP streams over the shared weights: a single voice's late stages (T <= 200 rows) leave most CUs idle and its ~300
        launches are latency chains, which the other voices' chains fill.  Streams come from the recycle pool (engine.py, _IDLE_STREAMS)."""
        if len(wavs) <= 1:
            return [self.acoustic_encode(w) for w in wavs]
        pool = _IDLE_STREAMS.setdefault(str(self.device), [])
        n_side = min(max_streams, len(wavs)) - 1
        # side_streams: streams the caller already owns (the lanes of a batch) are used before any is taken from the pool or created: with the
        # lanes holding the pooled streams, every batched call would otherwise create three more (8 hardware queues: a batch of 5 on the lanes
        # dropped from 62 to 35 audio-sec/s after a batch of 2 had run first)
        lent, seen = [], {self.stream.cuda_stream}
        for st in (side_streams or []):
            if st.cuda_stream not in seen and len(lent) < n_side:
                lent.append(st)
                seen.add(st.cuda_stream)
        own = [pool.pop() if pool else torch.cuda.Stream(self.device) for _ in range(n_side - len(lent))]
        side = lent + own
        streams = [self.stream] + side
        outs = []
        try:
            for st in side:
                st.wait_stream(self.stream)                 # inputs were produced on / ordered into the engine stream
            for i, w in enumerate(wavs):
                st = streams[i % len(streams)]
                o = self.acoustic_encode(w, stream=st)
                if st is not self.stream:
                    o.record_stream(self.stream)
                outs.append(o)
        finally:                                            # streams go back to the pool whatever happened (they are never just dropped)
            for st in side:
                self.stream.wait_stream(st)
            pool.extend(own)
        return outs

    def connector(self, which: str, x: torch.Tensor) -> torch.Tensor:
        c = self.w.ac_conn if which == "acoustic" else self.w.sem_conn
        R = x.shape[0]
        with torch.cuda.stream(self.stream):
            x = x.to(device=self.device, dtype=torch.float32).contiguous()
            out = torch.empty(R, self.cfg.hidden, dtype=torch.float32, device=self.device)
            ws = torch.empty(R, self.cfg.hidden, dtype=torch.float32, device=self.device)
            self._ck(self.lib.vv_connector_forward(C.byref(c), x.data_ptr(), R, out.data_ptr(), 0, ws.data_ptr(), self.sp), "connector")
        return out

    def embed_ids(self, ids: torch.Tensor) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            return self.w.embed[ids.to(self.device)].float()

    def close(self):
        self._drop_graphs()
