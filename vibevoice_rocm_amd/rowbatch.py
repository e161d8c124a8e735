"""Row-batched decode for the dialogues of ONE generate() call (reference: the batched loop of modeling_vibevoice_inference.py:430-673).

`Engine` lanes run B dialogues as B independent launch chains that each stream every LLM and diffusion-head weight once per frame.  Here the
B dialogues (2 <= B <= 4 per RowBatch; generate() runs 5..8 as two of them in one loop) share ONE chain for the two weight-heavy parts of a frame:

  graph A   Qwen2 decode step with R = 2 B rows (dialogue b = rows {2 b: positive, 2 b + 1: negative} of x, lens and one KV cache with 2 B
            rows): every weight matrix is read once for all dialogues (4..8 rows: the matrix-core GEMV of csrc/vv_gemv_rows.hip on
            fragment-major weight copies), then vv_llm_tail_batch = final norm, constrained logits, argmax / forced token and position
            bookkeeping per dialogue;
  graph H   vv_head_sample_batch: the CFG diffusion sampler for all B utterances, 2 B rows through every head matrix per solver step.

The conv tokenizers (acoustic decode, semantic encode) and the connectors stay per dialogue - each has its own streaming state - and run as
B concurrent hipGraphs on the lanes' streams between H and the next A (events fork / join them), exactly the launch sequences the lanes use.
They are enqueued only once H has finished: a lane queue that sits on a cross-stream wait while the main queue runs A and H slows every dependent
launch there.  The host loop, token state machine, speculation and rollback are those of the lock-step loop (modeling.py)."""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .engine import Engine

_UID = itertools.count(1)
# One worker thread per HIP stream, shared by every RowBatch of the process: two row batches of one generate() call (5..8 dialogues) put their
# lanes on the same four streams, and a stream must never be fed by two host threads at once (the first use of a graph CAPTURES on the stream).
# _JOBS: every launch handed to a worker and not yet known to be in its stream's queue.
_STREAM_WORKERS: Dict[int, "object"] = {}
_JOBS: List["object"] = []


def _submit(stream: torch.cuda.Stream, fn, *args):
    from concurrent.futures import ThreadPoolExecutor
    w = _STREAM_WORKERS.get(stream.cuda_stream)
    if w is None:
        w = _STREAM_WORKERS[stream.cuda_stream] = ThreadPoolExecutor(max_workers=1)
    _JOBS.append(w.submit(fn, *args))


def _wait_all_jobs():
    while _JOBS:
        _JOBS.pop(0).result()


class RowBatch:
    def __init__(self, lanes: List[Engine], stream: Optional[torch.cuda.Stream] = None):
        """lanes: one Engine per dialogue (conv state, conv graphs, its stream).  stream: the stream graphs A and H run on (default: lanes[0]'s);
        a lane whose stream IS that stream runs its conv tail inline behind H, the others fork / join through events."""
        self.lanes = lanes
        self.B = B = len(lanes)
        if not 2 <= B <= 4:
            raise L.VVError("row batching serves 2..4 dialogues per call")
        eng = self.main = lanes[0]
        self.lib, self.cfg, self.device, self.stream = eng.lib, eng.cfg, eng.device, (stream or eng.stream)
        self._on_main = [e.stream.cuda_stream == self.stream.cuda_stream for e in lanes]
        if eng.dtype != torch.bfloat16 or eng.kv_dtype != torch.bfloat16 or eng.w.quant is not None:
            raise L.VVError("row batching needs bf16 weights and a bf16 KV cache")
        self.uid = next(_UID)
        cfg, H = self.cfg, self.cfg.hidden
        f32 = dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        with torch.cuda.stream(self.stream):
            eng.w.ensure_frag()
            self.x = torch.zeros(2 * B, H, **f32)
            self.hidden = torch.zeros(2 * B, H, **f32)
            self.lens = torch.zeros(2 * B, **i32)
            self.frame_ctr = torch.zeros(B, **i32)
            self.token_dev = torch.zeros(B, **i32)
            self.forced_dev = torch.full((B,), -1, **i32)
            self.active_dev = torch.ones(B, **i32)
            self.logits = torch.zeros(B, 8, **f32)
            self.noise_dev = torch.zeros(B, cfg.latent, **f32)
            self.latent = torch.zeros(B, cfg.latent, **f32)
            self._llm_ws = torch.empty(self.lib.vv_llm_ws_bytes(C.byref(eng.w.llm), 2 * B), dtype=torch.uint8, device=self.device)
        self._pf_ws = None
        self._pf_rows = 0
        self.token_host = torch.zeros(B, dtype=torch.int32).pin_memory()
        self.forced_host = torch.full((B,), -1, dtype=torch.int32).pin_memory()
        self.active_host = torch.ones(B, dtype=torch.int32).pin_memory()
        self.noise_host = torch.zeros(2, B, cfg.latent, dtype=torch.float32).pin_memory()
        self._noise_k = 0
        self._forced_val = [-1] * B
        self._active_val = [1] * B
        self._tok_event = torch.cuda.Event()
        self._head_event = torch.cuda.Event()
        self._lane_event = [torch.cuda.Event() for _ in range(B)]
        self._lane_dirty = [False] * B      # lane b has work in flight that the next graph A must wait for
        # the conv tails go out from one worker thread per lane stream (a hipGraph launch costs the host ~0.08 ms: four in a row would delay the
        # last tail by a quarter of a millisecond; the HIP calls release the GIL), see _STREAM_WORKERS
        self.late_tails = os.environ.get("VV_RB_LATE_TAILS", "1") != "0"
        self._timing = [] if os.environ.get("VV_RB_TIMING") else None      # debug: per-step HIP events (A start / A end / H end / tails end)
        self._tcur = None
        self.kv = None
        self._kv_t = None
        self._graphs: Dict[tuple, int] = {}
        self._head_ws = None
        self._head_steps = 0
        self.valid_ids: List[int] = []
        self._w_valid = None
        self._ids_dev = None

    # -----------------------------------------------------------------------------------------------------------------
    @property
    def sp(self) -> int:
        return self.stream.cuda_stream

    def _ck(self, rc, what):
        L.check(rc, what)

    def _drop_graphs(self):
        for g in self._graphs.values():
            self.lib.vv_graph_destroy(g)
        self._graphs = {}

    def close(self):
        self._wait_jobs()
        self._drop_graphs()

    def flush(self):
        """every launch handed to the worker threads is in its stream's queue"""
        self._wait_jobs()

    def _wait_jobs(self):
        _wait_all_jobs()

    def _run(self, name: str, fn, *args):
        if not self.main.use_graphs:
            fn(*args)
            return
        key = (name,) + tuple(args)
        g = self._graphs.get(key)
        if g is None:
            _wait_all_jobs()          # no worker may be waiting on an event of this stream while it captures (another row batch's tails)
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

    # -----------------------------------------------------------------------------------------------------------------
    def begin(self, s_max: int, valid_ids: List[int], cfg_scale: float):
        """Fresh batch: one KV cache with 2 B rows sized for s_max tokens, every lane's conv state zeroed."""
        cfg, eng, B = self.cfg, self.main, self.B
        s_max = (int(s_max) + 63) // 64 * 64
        self.cfg_scale = float(cfg_scale)
        for b, e in enumerate(self.lanes):
            if not self._on_main[b]:
                e.sync_in()
                e.stream.wait_stream(self.stream)
        with torch.cuda.stream(self.stream):
            if self.kv is None or self.kv.s_max < s_max:
                shape = (cfg.layers, 2 * B, cfg.kv_heads, s_max, cfg.head_dim)
                self._kv_t = (torch.zeros(shape, dtype=torch.bfloat16, device=self.device), torch.zeros(shape, dtype=torch.bfloat16, device=self.device))
                kv = L.KV()
                kv.k, kv.v = self._kv_t[0].data_ptr(), self._kv_t[1].data_ptr()
                if cfg.head_dim == 128:
                    self._kv_vt = torch.empty((cfg.layers, 2 * B, cfg.kv_heads, s_max // 32, cfg.head_dim, 32), dtype=torch.bfloat16, device=self.device)
                    kv.vt = self._kv_vt.data_ptr()
                kv.kvdt = L.VV_BF16
                kv.layers, kv.rows, kv.kv_heads, kv.s_max, kv.head_dim = cfg.layers, 2 * B, cfg.kv_heads, s_max, cfg.head_dim
                self.kv = kv
                self._drop_graphs()
            ids = sorted(set(int(i) for i in valid_ids))
            if ids != self.valid_ids:
                if len(ids) > 8:
                    raise L.VVError("row batching serves at most 8 constrained vocabulary ids")
                self.valid_ids = ids
                arr = (C.c_int * len(ids))(*ids)
                self._w_valid = torch.empty(len(ids), cfg.hidden, dtype=torch.bfloat16, device=self.device)
                self._ck(self.lib.vv_gather_rows(eng.w.lm_head.data_ptr(), eng.w.wdt, cfg.hidden, arr, len(ids), self._w_valid.data_ptr(), self.sp), "vv_gather_rows")
                self._ids_dev = torch.tensor(ids, dtype=torch.int32, device=self.device)
                self._drop_graphs()
            if self._head_steps != eng.n_steps:
                self._head_ws = torch.empty(self.lib.vv_head_ws_bytes_batch(C.byref(eng.w.head), eng.n_steps, B), dtype=torch.uint8, device=self.device)
                self._head_steps = eng.n_steps
                self._drop_graphs()
            self.lens.zero_()
            self.frame_ctr.zero_()
            self.forced_dev.fill_(-1)
            self.active_dev.fill_(1)
            self.forced_host.fill_(-1)      # the pinned mirrors are uploaded WHOLE whenever one entry changes: they must not carry the last call's values
            self.active_host.fill_(1)
            self._forced_val = [-1] * B
            self._active_val = [1] * B
        for e in self.lanes:
            with torch.cuda.stream(e.stream):
                e.reset_speech_caches()
        self._lane_dirty = [False] * B

    def prefill(self, b: int, embeds: torch.Tensor, neg: bool = False, chunk: int = 1024, neg_embed: Optional[torch.Tensor] = None):
        """Prompt prefill of dialogue b on cache row 2 b (neg: 2 b + 1), on the main stream; hidden[row] = last hidden state.  neg_embed [1, H]:
        the negative branch's one-token prompt rides along as one more row of the last chunk (cache row 2 b + 1, position 0; Engine.prefill);
        `commit_negative` then puts the branch in use."""
        row = 2 * b + (1 if neg else 0)
        L0 = embeds.shape[0]
        n_chunks = max(1, -(-L0 // max(1, chunk)))
        size = -(-L0 // n_chunks)
        size = min(chunk, (size + 31) // 32 * 32) if n_chunks > 1 else L0
        with torch.cuda.stream(self.stream):
            if max(size, 1) + 1 > self._pf_rows:
                self._pf_rows = max(size + 1, 64)
                self._pf_ws = torch.empty(self.lib.vv_llm_ws_bytes(C.byref(self.main.w.llm), self._pf_rows), dtype=torch.uint8, device=self.device)
            for c0 in range(0, L0, size):
                c1 = min(L0, c0 + size)
                last = c1 == L0 and neg_embed is not None
                n = c1 - c0 + (1 if last else 0)
                lens = torch.arange(c0, c0 + n, dtype=torch.int32, device=self.device)
                rows = torch.full((n,), row, dtype=torch.int32, device=self.device)
                xe = embeds[c0:c1].contiguous()
                if last:
                    lens[-1] = 0
                    rows[-1] = 2 * b + 1
                    xe = torch.cat([xe, neg_embed.to(xe.dtype).reshape(1, -1)])
                out = torch.empty(n, self.cfg.hidden, dtype=torch.float32, device=self.device)
                self._ck(self.lib.vv_llm_forward(C.byref(self.main.w.llm), C.byref(self.kv), xe.data_ptr(), xe.stride(0), n, lens.data_ptr(), rows.data_ptr(),
                                                 out.data_ptr(), out.stride(0), self._pf_ws.data_ptr(), self.sp), "vv_llm_forward")
            if neg_embed is not None:
                self.hidden[row].copy_(out[-2])
                self.hidden[2 * b + 1].copy_(out[-1])
            else:
                self.hidden[row].copy_(out[-1])
            self.lens[row] = L0

    def commit_negative(self, b: int):
        with torch.cuda.stream(self.stream):
            self.lens[2 * b + 1] = 1

    def first_token(self, b: int, forced: Optional[int]) -> int:
        """Token selection right after the prefill of dialogue b (hidden[2 b] holds its last prompt state)."""
        nv = len(self.valid_ids)
        with torch.cuda.stream(self.stream):
            self._set_forced({b: forced})
            a = L.LinArgs()
            a.x, a.ldx, a.m = self.hidden[2 * b].data_ptr(), self.cfg.hidden, 1
            a.w, a.n, a.k, a.wdt = self._w_valid.data_ptr(), nv, self.cfg.hidden, L.VV_BF16
            a.out, a.ldo = self.logits[b].data_ptr(), nv
            self._ck(self.lib.vv_linear(C.byref(a), self.sp), "lm_head")
            self._ck(self.lib.vv_argmax_ids(self.logits[b].data_ptr(), nv, self._ids_dev.data_ptr(), self.token_dev[b:].data_ptr(),
                                            self.forced_dev[b:].data_ptr(), self.sp), "vv_argmax_ids")
            self.token_host.copy_(self.token_dev, non_blocking=True)
        self.stream.synchronize()
        return int(self.token_host[b])

    # -----------------------------------------------------------------------------------------------------------------
    def _set_forced(self, forced: Dict[int, Optional[int]]):
        """Device-side forced tokens (-1: none), uploaded only when one changes (see Engine._set_forced)."""
        changed = False
        for b, f in forced.items():
            v = -1 if f is None else int(f)
            if v != self._forced_val[b]:
                self._forced_val[b] = v
                self.forced_host[b] = v
                changed = True
        if changed:
            self.forced_dev.copy_(self.forced_host, non_blocking=True)

    def set_active(self, b: int, on: bool):
        """A finished dialogue stays in the row batch (its rows are computed and dropped) but no longer advances its positions."""
        v = 1 if on else 0
        if v != self._active_val[b]:
            self._active_val[b] = v
            self.active_host[b] = v
            with torch.cuda.stream(self.stream):
                self.active_dev.copy_(self.active_host, non_blocking=True)

    def _seq_A(self, tok_start, tok_diff):
        eng, B, H = self.main, self.B, self.cfg.hidden
        nv = len(self.valid_ids)
        self._ck(self.lib.vv_llm_forward(C.byref(eng.w.llm), C.byref(self.kv), self.x.data_ptr(), H, 2 * B, self.lens.data_ptr(), None, None, 0,
                                         self._llm_ws.data_ptr(), self.sp), "vv_llm_forward")
        self._ck(self.lib.vv_llm_tail_batch(C.byref(eng.w.llm), self._llm_ws.data_ptr(), H, B, self.hidden.data_ptr(), H, self._w_valid.data_ptr(), nv,
                                            self._ids_dev.data_ptr(), self.logits.data_ptr(), self.token_dev.data_ptr(), self.forced_dev.data_ptr(),
                                            self.lens.data_ptr(), tok_start, tok_diff, self.frame_ctr.data_ptr(), self.active_dev.data_ptr(), self.sp),
                 "vv_llm_tail_batch")

    def _seq_H(self, cfg_scale):
        eng, cfg = self.main, self.cfg
        self._ck(self.lib.vv_head_sample_batch(C.byref(eng.w.head), self.hidden.data_ptr(), cfg.hidden, self.noise_dev.data_ptr(), cfg.latent,
                                               eng.temb.data_ptr(), eng._coefs, eng.n_steps, cfg_scale, self.latent.data_ptr(), cfg.latent, self.B,
                                               self._head_ws.data_ptr(), self.sp), "vv_head_sample_batch")

    def _seq_conv(self, b, uid):
        """acoustic decode -> semantic encode -> connectors of dialogue b on ITS stream, into rows 2 b, 2 b + 1 of the next step's input"""
        e, cfg = self.lanes[b], self.cfg
        lib, w = self.lib, e.w
        blob = w.state_blob()
        self._ck(lib.vv_copy_rows(blob.data_ptr(), blob.numel(), e._state_snap.data_ptr(), blob.numel(), 1, blob.numel(), e.sp), "snapshot")
        lat = self.latent[b].data_ptr()
        self._ck(lib.vv_decoder_forward(C.byref(w.dec), lat, 1, 1.0 / w.speech_scale, -w.speech_bias, e.wav.data_ptr(), e._dec_ws.data_ptr(), e.sp),
                 "vv_decoder_forward")
        self._ck(lib.vv_encoder_forward(C.byref(w.sem), e.wav.data_ptr(), cfg.hop, e.sem.data_ptr(), e._sem_ws.data_ptr(), e.sp), "vv_encoder_forward")
        self._ck(lib.vv_connector_pair(C.byref(w.ac_conn), C.byref(w.sem_conn), lat, e.sem.data_ptr(), self.x[2 * b].data_ptr(), cfg.hidden, 2,
                                       e.conn_ws.data_ptr(), e.sp), "connectors")

    def _seq_embed(self, b, uid):
        e, cfg = self.lanes[b], self.cfg
        self._ck(self.lib.vv_embed_row(e.w.embed.data_ptr(), e.w.wdt, cfg.hidden, self.token_dev[b:].data_ptr(), self.x[2 * b].data_ptr(), e.sp), "embed")
        self._ck(self.lib.vv_copy_rows(self.x[2 * b].data_ptr(), 0, self.x[2 * b + 1].data_ptr(), cfg.hidden, 1, cfg.hidden, e.sp), "copy")

    def _join_lanes(self):
        """the next graph A reads every lane's rows of x"""
        self._wait_jobs()
        for b in range(self.B):
            if self._lane_dirty[b]:
                self.stream.wait_event(self._lane_event[b])
                self._lane_dirty[b] = False

    def _lane_done(self, b):
        if not self._on_main[b]:
            self._lane_event[b].record(self.lanes[b].stream)
            self._lane_dirty[b] = True

    def decode_begin(self, tok_start: int, tok_diff: int, forced: Dict[int, Optional[int]]):
        """graph A for all dialogues + the asynchronous token read-back; `decode_end` returns the tokens."""
        with torch.cuda.stream(self.stream):
            self._join_lanes()
            self._set_forced(forced)
            if self._timing is not None:
                self._tcur = dict(a0=torch.cuda.Event(enable_timing=True), a1=torch.cuda.Event(enable_timing=True), h1=None, t=[])
                self._timing.append(self._tcur)
                self._tcur["a0"].record(self.stream)
            self._run("A", self._seq_A, int(tok_start), int(tok_diff))
            if self._timing is not None:
                self._tcur["a1"].record(self.stream)
            self.token_host.copy_(self.token_dev, non_blocking=True)
            self._tok_event.record(self.stream)

    def decode_end(self) -> List[int]:
        self._tok_event.synchronize()
        return [int(t) for t in self.token_host]

    def speech(self, which: List[int], noise: Dict[int, torch.Tensor]):
        """Diffusion sampling for the whole batch (graph H), then the conv tail of the dialogues in `which`, each on its own stream."""
        self.speech_begin(which, noise)
        self.speech_tails(which)

    def speech_begin(self, which: List[int], noise: Dict[int, torch.Tensor]):
        """noise upload + graph H on the main stream"""
        cfg = self.cfg
        with torch.cuda.stream(self.stream):
            self._noise_k ^= 1
            nh = self.noise_host[self._noise_k]
            for b in which:
                nh[b].copy_(noise[b].reshape(-1)[: cfg.latent])
            self.noise_dev.copy_(nh, non_blocking=True)
            self._run("H", self._seq_H, float(self.cfg_scale))
            self._head_event.record(self.stream)
            if self._timing is not None and self._tcur is not None:
                self._tcur["h1"] = torch.cuda.Event(enable_timing=True)
                self._tcur["h1"].record(self.stream)

    def speech_tails(self, which: List[int]):
        """The conv tails of the dialogues in `which`, enqueued once H HAS FINISHED (late_tails): a lane queue that sits on a cross-stream wait
        while the main queue runs A and H slows every dependent launch there by ~6 us (graph A 1.35 -> 2.28 ms, H 1.72 -> 2.9 ms, measured with
        VV_RB_TIMING=1), so the lanes' queues stay EMPTY until their work can run.  With several row batches in one loop (more than 4 dialogues)
        the caller enqueues every batch's A and H first and the tails afterwards: a batch's tails then overlap the next batch's A and H."""
        late = self.late_tails

        def tail(b):
            e = self.lanes[b]
            torch.cuda.set_device(self.device)
            if late and not self._on_main[b]:
                self._head_event.synchronize()      # the worker itself waits for H: its launch goes out the moment the sampler is done, and needs
            with torch.cuda.stream(e.stream):       # no stream-side wait on the event any more
                e._run("RBconv", self._seq_conv, b, self.uid)
                self._lane_done(b)
                if self._timing is not None and self._tcur is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(e.stream)
                    self._tcur["t"].append(ev)

        self._wait_jobs()       # another row batch's worker may still be feeding (first use: capturing on) one of these streams
        inline = []
        for b in which:
            if self._on_main[b]:
                inline.append(b)
            else:
                if not late:
                    # a cross-stream wait is issued HERE, by the thread that owns the main stream: a wait on an event of a stream that another
                    # thread is capturing (first use of a graph) is a capture-isolation error, and only this thread captures on the main stream
                    self.lanes[b].stream.wait_event(self._head_event)
                _submit(self.lanes[b].stream, tail, b)
        if late and inline:
            self._head_event.synchronize()
        for b in inline:
            tail(b)             # this dialogue's tail rides the main stream, behind H

    def embed(self, b: int):
        """next input of dialogue b = embed_tokens[its token], on its stream (after a rollback of that stream, if any)"""
        self._wait_jobs()
        e = self.lanes[b]
        with torch.cuda.stream(e.stream):
            if not self._on_main[b]:
                e.stream.wait_event(self._tok_event)
            e._run("RBembed", self._seq_embed, b, self.uid)
            self._lane_done(b)

    def rollback(self, b: int):
        self._wait_jobs()
        self.lanes[b].rollback_speech_state()

    def reset_speech(self, b: int):
        self._wait_jobs()
        e = self.lanes[b]
        with torch.cuda.stream(e.stream):
            e.reset_speech_caches()

    def synchronize(self):
        self._wait_jobs()
        for e in self.lanes:
            e.stream.synchronize()
        if self._timing:
            import statistics
            T = [t for t in self._timing if t["h1"] is not None and len(t["t"]) == self.B]
            rows = []
            for i in range(5, len(T) - 1):
                t, nx = T[i], T[i + 1]
                tails = max(t["h1"].elapsed_time(ev) for ev in t["t"])
                rows.append((t["a0"].elapsed_time(t["a1"]), t["a1"].elapsed_time(t["h1"]), tails, t["a0"].elapsed_time(nx["a0"])))
            if rows:
                med = [statistics.median(r[k] for r in rows) for k in range(4)]
                print(f"[rowbatch timing] {len(rows)} steps, medians: A {med[0]:.3f} ms, token copy + noise + H {med[1]:.3f} ms, tails after H {med[2]:.3f} ms, "
                      f"step period {med[3]:.3f} ms (gap {med[3] - med[0] - med[1] - med[2]:.3f})", flush=True)
            self._timing.clear()
