"""Device-resident weights in the layout the MI355X kernels stream, plus the ctypes descriptors that point at them.

Input is a flat state dict with the reference's tensor names (`synth.state_dict_shapes`).
Re-layouts done once at load time (HBM is 288 GB: we spend bytes to make every per-frame read a straight stream):
  * q/k/v projection rows concatenated into one [q+2kv, H] matrix (+ one bias vector) -> one GEMV per layer;
  * dense SConv1d weights [C_out, C_in, k] -> [C_out, k*C_in] (tap-major) so a conv is a GEMM over the
    channels-last activation buffer read in place with overlapping rows;
  * SConvTranspose1d weights [C_in, C_out, 2s] -> [(r, C_out), (j, C_in)] (j = 0 previous input, 1 current), bias
    tiled over r: a transposed conv with k = 2s is one GEMM producing s output rows per input row;
  * depthwise taps [C, 1, 7] -> [C, 7] fp32; every 1-D tensor fp32.
Matrix weights are stored in `wdtype` (fp32 for graded parity, bf16 for the benchmark configs).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .config import VVConfig


def _wdt(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return L.VV_F32
    if dtype == torch.bfloat16:
        return L.VV_BF16
    raise ValueError(f"unsupported weight dtype {dtype}")


FP8_MAX = 448.0      # largest finite e4m3fn magnitude


def quantize_e4m3_pow2(w: torch.Tensor):
    """Weight-only fp8 of a [N, K] matrix: one POWER-OF-TWO scale per output row (2^ceil(log2(amax / 448))) and round-to-nearest
    e4m3fn codes.  With a power-of-two scale the dequantised weight code * scale is exactly representable in bf16, so the bf16
    copy the GEMM-shaped uses keep (prefill, T > 4) and the fp8 copy the decode GEMVs stream are the SAME effective matrix.
    Returns (codes uint8 [N, K], scale fp32 [N], effective weights fp32 [N, K])."""
    wf = w.detach().float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / FP8_MAX)))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous(), q.float() * scale[:, None]


def fp8_matrix_names(cfg: VVConfig):
    """State-dict keys of the matrices that run as weight-streaming GEMVs every frame and get an fp8 companion: the LLM's linears,
    the diffusion head's SwiGLU matrices, the FFN linears of the 1-row conv stage (decoder stage 0 / semantic-encoder last stage)."""
    names = []
    for l in range(cfg.layers):
        q = f"model.language_model.layers.{l}."
        names += [q + f"self_attn.{n}_proj.weight" for n in "qkvo"] + [q + f"mlp.{n}_proj.weight" for n in ("gate", "up", "down")]
    for l in range(cfg.head_layers):
        q = f"model.prediction_head.layers.{l}."
        names += [q + f"ffn.{n}_proj.weight" for n in ("gate", "up", "down")]
    n_st = len(cfg.ac_ratios) + 1
    for j in range(cfg.ac_depths[-1]):
        names += [f"model.acoustic_tokenizer.decoder.stages.0.{j}.ffn.linear{i}.weight" for i in (1, 2)]
    for j in range(cfg.sem_depths[-1]):
        names += [f"model.semantic_tokenizer.encoder.stages.{n_st - 1}.{j}.ffn.linear{i}.weight" for i in (1, 2)]
    return names


def fp8_effective_state_dict(cfg: VVConfig, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The state dict a weight_quant="fp8" engine computes with (for the checker side of parity tests): the listed matrices replaced
    by code * scale (q/k/v are quantised as the fused [q|k|v] matrix, row scales make that identical to quantising them apart)."""
    out = dict(sd)
    for name in fp8_matrix_names(cfg):
        out[name] = quantize_e4m3_pow2(sd[name])[2].to(sd[name].dtype if sd[name].dtype == torch.float32 else torch.float32)
    return out


class DeviceWeights:
    """Owns every device tensor of the model and the C descriptors (vv_llm, vv_head, vv_convnet x3, vv_connector x2)."""

    def __init__(self, cfg: VVConfig, sd: Dict[str, torch.Tensor], device, wdtype=torch.bfloat16, streaming_state=True, quant=None):
        self.cfg, self.device, self.wdtype = cfg, torch.device(device), wdtype
        if quant not in (None, "fp8"):
            raise ValueError(f"weight_quant {quant!r}: only None or 'fp8' (weight-only e4m3, decode GEMVs) is built")
        if quant == "fp8" and wdtype != torch.bfloat16:
            raise ValueError("weight_quant='fp8' keeps bf16 copies for the GEMM-shaped uses: torch_dtype must be bfloat16")
        self.quant = quant
        self._fp8_names = set(fp8_matrix_names(cfg)) if quant == "fp8" else set()
        self.wdt = _wdt(wdtype)
        self._keep: List[object] = []     # tensors / ctypes arrays that must outlive the descriptors
        self._sd = sd
        self.state_tensors: Dict[str, List[torch.Tensor]] = {"acoustic_dec": [], "semantic_enc": []}
        # every streaming-state tensor (conv left contexts, mixer histories) is a 256-byte aligned view into ONE arena, so the
        # engine can snapshot / roll back the whole speech state with a single copy (speculative frame launch)
        self._state_arena = torch.zeros(1 << 21, dtype=torch.float32, device=self.device)
        self._state_used = 0
        self._build_llm()
        self._build_head()
        self.dec = self._build_convnet("model.acoustic_tokenizer.decoder.", decoder=True, filters=cfg.ac_dec_filters,
                                       ratios=cfg.ac_ratios, depths_enc=cfg.ac_depths, dim=cfg.ac_dim, eps=cfg.ac_eps,
                                       state_key="acoustic_dec" if streaming_state else None)
        self.sem = self._build_convnet("model.semantic_tokenizer.encoder.", decoder=False, filters=cfg.sem_filters,
                                       ratios=cfg.sem_ratios, depths_enc=cfg.sem_depths, dim=cfg.sem_dim, eps=cfg.sem_eps,
                                       state_key="semantic_enc" if streaming_state else None)
        self.ac_enc = self._build_convnet("model.acoustic_tokenizer.encoder.", decoder=False, filters=cfg.ac_filters,
                                          ratios=cfg.ac_ratios, depths_enc=cfg.ac_depths, dim=cfg.ac_dim, eps=cfg.ac_eps,
                                          state_key=None)     # voice prompts are encoded whole (non-streaming)
        self.ac_conn = self._build_connector("model.acoustic_connector.", cfg.ac_dim)
        self.sem_conn = self._build_connector("model.semantic_connector.", cfg.sem_dim)
        self.speech_scale = float(sd["model.speech_scaling_factor"].float().item())
        self.speech_bias = float(sd["model.speech_bias_factor"].float().item())
        self._sd = None

    # ---- helpers -------------------------------------------------------------------------------------------------
    @staticmethod
    def _aligned(t: torch.Tensor) -> torch.Tensor:
        # the streaming / matrix-core kernels need 16-byte aligned rows; a view into a packed blob may not be
        return t if t.data_ptr() % 256 == 0 else t.clone()

    def _mat(self, t: torch.Tensor) -> torch.Tensor:
        t = self._aligned(t.detach().to(device=self.device, dtype=self.wdtype).contiguous())
        self._keep.append(t)
        return t

    def _mat_q(self, t: torch.Tensor, quantise: bool):
        """(bf16 matrix, vv_w8 companion): in fp8 mode the bf16 copy holds the dequantised values (exact, power-of-two scales)."""
        w8 = L.W8()
        if not (quantise and self.quant == "fp8"):
            return self._mat(t), w8
        codes, scale, eff = quantize_e4m3_pow2(t.to(self.device))
        codes = self._aligned(codes.to(self.device))
        self._keep.append(codes)
        w8.q, w8.scale = L.ptr(codes), L.ptr(self._vec(scale))
        return self._mat(eff), w8

    def _vec(self, t: torch.Tensor) -> torch.Tensor:
        t = self._aligned(t.detach().to(device=self.device, dtype=torch.float32).contiguous())
        self._keep.append(t)
        return t

    def _state_zeros(self, rows: int, ch: int) -> torch.Tensor:
        n = rows * ch
        if self._state_used + n > self._state_arena.numel():
            raise ValueError("streaming-state arena too small for this tokenizer configuration")
        t = self._state_arena[self._state_used: self._state_used + n].view(rows, ch)
        self._state_used += (n + 63) // 64 * 64
        return t

    def fork(self) -> "DeviceWeights":
        """A second set of descriptors over the SAME device weights with a streaming-state arena of its own (a lane of a lock-step batch,
        a concurrent dialogue): nothing is re-laid-out or copied except the few KB of ctypes descriptors that carry state pointers."""
        w = object.__new__(DeviceWeights)
        w.__dict__.update(self.__dict__)
        w._keep = list(self._keep)
        w._state_arena = torch.zeros_like(self._state_arena)
        old0, new0 = self._state_arena.data_ptr(), w._state_arena.data_ptr()

        def reloc(p):
            return None if not p else new0 + (int(p) - old0)

        def fork_net(net: "L.ConvNet") -> "L.ConvNet":
            n2 = L.ConvNet.from_buffer_copy(net)
            for i in range(net.n_stages):
                n2.sample[i].state = reloc(net.sample[i].state)
                nb = net.n_blocks[i]
                if nb > 0:
                    arr = (L.Block * nb)()
                    C.memmove(arr, net.blocks[i], C.sizeof(L.Block) * nb)
                    for j in range(nb):
                        arr[j].hist = reloc(arr[j].hist)
                        arr[j].hs = reloc(arr[j].hs)
                    w._keep.append(arr)
                    n2.blocks[i] = C.cast(arr, C.POINTER(L.Block))
            n2.head.state = reloc(net.head.state)
            return n2

        w.dec, w.sem = fork_net(self.dec), fork_net(self.sem)
        f32 = 4
        w.state_tensors = {k: [w._state_arena[(t.data_ptr() - old0) // f32: (t.data_ptr() - old0) // f32 + t.numel()].view_as(t) for t in v]
                           for k, v in self.state_tensors.items()}
        return w

    @staticmethod
    def frag_major(w: torch.Tensor) -> Optional[torch.Tensor]:
        """[N, K] bf16 -> the fragment-major copy [N / 16][K / 32][4][16][8] (include/vv_hip.h, vv_llm_layer.f_*): element (16 g + n, 32 j + 8 c + e)
        at ((g * K/32 + j) * 64 + 16 c + n) * 8 + e, one matrix-core B fragment per KB of contiguous memory."""
        n, k = w.shape
        if n % 16 or k % 32 or w.dtype != torch.bfloat16:
            return None
        return w.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()

    def ensure_frag(self) -> None:
        """Fragment-major copies of the LLM's and the diffusion head's per-frame matrices for the row-batched decode step (rowbatch.py): built
        once, on first use, next to the row-major matrices (which the prefill GEMMs and the 1..4-row GEMVs keep using); +2.9 GB at 1.5B."""
        if getattr(self, "_frag_done", False) or self.wdtype != torch.bfloat16:
            return
        by_ptr = {t.data_ptr(): t for t in self._keep if isinstance(t, torch.Tensor)}

        def frag_of(p, n, k):
            t = by_ptr.get(int(p)) if p else None
            if t is None or tuple(t.shape) != (n, k):
                return None
            f = self.frag_major(t)
            if f is None:
                return None
            f = self._aligned(f)
            self._keep.append(f)
            return f.data_ptr()

        cfg = self.cfg
        qkvd = (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim
        for l in range(cfg.layers):
            lay = self.llm.layer[l]
            lay.f_qkv = frag_of(lay.wqkv, qkvd, cfg.hidden)
            lay.f_o = frag_of(lay.wo, cfg.hidden, cfg.heads * cfg.head_dim)
            lay.f_gate = frag_of(lay.wgate, cfg.inter, cfg.hidden)
            lay.f_up = frag_of(lay.wup, cfg.inter, cfg.hidden)
            lay.f_down = frag_of(lay.wdown, cfg.hidden, cfg.inter)
        for l in range(cfg.head_layers):
            lay = self.head.layer[l]
            lay.f_gate = frag_of(lay.wgate, cfg.head_ffn, cfg.head_hidden)
            lay.f_up = frag_of(lay.wup, cfg.head_ffn, cfg.head_hidden)
            lay.f_down = frag_of(lay.wdown, cfg.head_hidden, cfg.head_ffn)
        self._frag_done = True

    def state_blob(self) -> torch.Tensor:
        """All streaming state of the speech path as one flat fp32 tensor."""
        return self._state_arena[: max(self._state_used, 64)]

    def _zeros(self, *shape) -> torch.Tensor:
        t = torch.zeros(*shape, dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep if isinstance(t, torch.Tensor))

    # ---- Qwen2 ---------------------------------------------------------------------------------------------------
    def _build_llm(self):
        cfg, sd = self.cfg, self._sd
        p = "model.language_model."
        self.embed = self._mat(sd[p + "embed_tokens.weight"])
        self.lm_head = self.embed if (cfg.tie or "lm_head.weight" not in sd) else self._mat(sd["lm_head.weight"])
        layers = (L.LlmLayer * cfg.layers)()
        for l in range(cfg.layers):
            q = f"{p}layers.{l}."
            qz = self.quant == "fp8"
            lay = layers[l]
            wqkv, lay.q_qkv = self._mat_q(torch.cat([sd[q + "self_attn.q_proj.weight"], sd[q + "self_attn.k_proj.weight"],
                                                    sd[q + "self_attn.v_proj.weight"]], dim=0), qz)
            bqkv = self._vec(torch.cat([sd[q + "self_attn.q_proj.bias"], sd[q + "self_attn.k_proj.bias"],
                                        sd[q + "self_attn.v_proj.bias"]], dim=0))
            lay.ln1 = L.ptr(self._vec(sd[q + "input_layernorm.weight"]))
            lay.ln2 = L.ptr(self._vec(sd[q + "post_attention_layernorm.weight"]))
            lay.wqkv, lay.bqkv = L.ptr(wqkv), L.ptr(bqkv)
            for field, qf, key in (("wo", "q_o", "self_attn.o_proj.weight"), ("wgate", "q_gate", "mlp.gate_proj.weight"),
                                   ("wup", "q_up", "mlp.up_proj.weight"), ("wdown", "q_down", "mlp.down_proj.weight")):
                wm, w8 = self._mat_q(sd[q + key], qz)
                setattr(lay, field, L.ptr(wm))
                setattr(lay, qf, w8)
        # Qwen2RotaryEmbedding.compute_default_rope_parameters, same fp32 ops as the reference stack
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float) / cfg.head_dim))
        self.inv_freq = self._vec(inv_freq)
        m = L.Llm()
        m.wdt, m.hidden, m.inter, m.layers = self.wdt, cfg.hidden, cfg.inter, cfg.layers
        m.heads, m.kv_heads, m.head_dim, m.rms_eps = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.rms_eps
        m.inv_freq = L.ptr(self.inv_freq)
        m.final_norm = L.ptr(self._vec(sd[p + "norm.weight"]))
        m.layer = C.cast(layers, C.POINTER(L.LlmLayer))
        self._keep.append(layers)
        self.llm = m

    # ---- diffusion head ------------------------------------------------------------------------------------------
    def _build_head(self):
        cfg, sd = self.cfg, self._sd
        p = "model.prediction_head."
        layers = (L.HeadLayer * cfg.head_layers)()
        for l in range(cfg.head_layers):
            q = f"{p}layers.{l}."
            lay = layers[l]
            lay.norm_w = L.ptr(self._vec(sd[q + "norm.weight"]))
            for field, qf, key in (("wgate", "q_gate", "ffn.gate_proj.weight"), ("wup", "q_up", "ffn.up_proj.weight"),
                                   ("wdown", "q_down", "ffn.down_proj.weight")):
                wm, w8 = self._mat_q(sd[q + key], self.quant == "fp8")
                setattr(lay, field, L.ptr(wm))
                setattr(lay, qf, w8)
            lay.adaln = L.ptr(self._mat(sd[q + "adaLN_modulation.1.weight"]))
        h = L.Head()
        h.wdt, h.D, h.ffn, h.layers, h.latent, h.cond_dim = self.wdt, cfg.head_hidden, cfg.head_ffn, cfg.head_layers, cfg.latent, cfg.hidden
        h.eps = cfg.head_eps
        h.noisy_proj = L.ptr(self._mat(sd[p + "noisy_images_proj.weight"]))
        h.cond_proj = L.ptr(self._mat(sd[p + "cond_proj.weight"]))
        h.final_adaln = L.ptr(self._mat(sd[p + "final_layer.adaLN_modulation.1.weight"]))
        h.final_linear = L.ptr(self._mat(sd[p + "final_layer.linear.weight"]))
        h.layer = C.cast(layers, C.POINTER(L.HeadLayer))
        self._keep.append(layers)
        # G = [P F ; F] (fp32, built once in fp64 from the matrices the kernels stream: bf16-rounded in bf16 mode): every solver-step
        # boundary - final linear, CFG, DPM-Solver++ update, next noisy_images_proj, all linear in the modulated hidden state - is
        # one GEMV over it (csrc/vv_fused.hip); P (F y) == (P F) y up to fp32 rounding
        P = sd[p + "noisy_images_proj.weight"].to(device=self.device, dtype=self.wdtype).double()
        F = sd[p + "final_layer.linear.weight"].to(device=self.device, dtype=self.wdtype).double()
        self.head_g = self._vec(torch.cat([P @ F, F], dim=0).float())
        h.fused_g = L.ptr(self.head_g)
        self.head = h
        self.t_mlp0 = self._mat(sd[p + "t_embedder.mlp.0.weight"])
        self.t_mlp2 = self._mat(sd[p + "t_embedder.mlp.2.weight"])

    # ---- conv tokenizers -----------------------------------------------------------------------------------------
    def _conv(self, w: torch.Tensor, b: torch.Tensor, stride: int, transposed: bool, state_key) -> L.Conv:
        c = L.Conv()
        if transposed:
            cin, cout, k = w.shape
            s = stride
            assert k == 2 * s, "SConvTranspose1d with k != 2*stride is not on the shipped path"
            w4 = w.reshape(cin, cout, 2, s).flip(2)                     # [ci, co, j, r]: j=0 -> taps r+s (previous input)
            wl = w4.permute(3, 1, 2, 0).reshape(s * cout, 2 * cin)      # [(r, co), (j, ci)]
            bl = b.repeat(s)
            ctx = 1
        else:
            cout, cin, k = w.shape
            wl = w.permute(0, 2, 1).reshape(cout, k * cin)              # [co, (tap, ci)]
            bl = b
            ctx = k - stride
        c.w, c.b = L.ptr(self._mat(wl)), L.ptr(self._vec(bl))
        c.cin, c.cout, c.kk, c.stride, c.transposed = cin, cout, k, stride, int(transposed)
        if state_key is not None and ctx > 0:
            st = self._state_zeros(ctx, cin)
            self.state_tensors[state_key].append(st)
            c.state = L.ptr(st)
        else:
            c.state = None
        return c

    def _blocks(self, prefix: str, n: int, ch: int, state_key):
        sd = self._sd
        arr = (L.Block * max(n, 1))()
        for j in range(n):
            q = f"{prefix}{j}."
            b = arr[j]
            b.gamma = L.ptr(self._vec(sd[q + "gamma"]))
            b.ffn_gamma = L.ptr(self._vec(sd[q + "ffn_gamma"]))
            b.norm_w = L.ptr(self._vec(sd[q + "norm.weight"]))
            b.ffn_norm_w = L.ptr(self._vec(sd[q + "ffn_norm.weight"]))
            b.dw_w = L.ptr(self._vec(sd[q + "mixer.conv.conv.conv.weight"].reshape(ch, 7)))
            b.dw_b = L.ptr(self._vec(sd[q + "mixer.conv.conv.conv.bias"]))
            qz = state_key is not None and (q + "ffn.linear1.weight") in self._fp8_names
            w1m, b.q_w1 = self._mat_q(sd[q + "ffn.linear1.weight"], qz)
            w2m, b.q_w2 = self._mat_q(sd[q + "ffn.linear2.weight"], qz)
            b.w1, b.b1 = L.ptr(w1m), L.ptr(self._vec(sd[q + "ffn.linear1.bias"]))
            b.w2, b.b2 = L.ptr(w2m), L.ptr(self._vec(sd[q + "ffn.linear2.bias"]))
            if state_key is not None:
                h = self._state_zeros(6, ch)
                self.state_tensors[state_key].append(h)
                b.hist = L.ptr(h)
                if ch == 2048 and not qz:
                    # the one-row stage: the history part of the depthwise conv is carried as state (hs = sum_k<6 tap_k * hist_k, zero with
                    # hist, inside the same arena so reset / snapshot / rollback cover it) and the newest row's tap is packed
                    b.dw_last = L.ptr(self._vec(sd[q + "mixer.conv.conv.conv.weight"].reshape(ch, 7)[:, 6].contiguous()))
                    b.hs = L.ptr(self._state_zeros(1, ch))
            else:
                b.hist = None
        self._keep.append(arr)
        return arr

    def _build_convnet(self, prefix, decoder, filters, ratios, depths_enc, dim, eps, state_key) -> L.ConvNet:
        sd = self._sd
        n = len(depths_enc)
        if n > L.VV_MAX_STAGES:
            raise ValueError("too many tokenizer stages")
        net = L.ConvNet()
        net.wdt, net.n_stages, net.eps = self.wdt, n, eps
        if decoder:
            depths = list(reversed(depths_enc))
            for i in range(n):
                ch = filters * 2 ** (n - 1 - i)
                if i == 0:
                    q = prefix + "upsample_layers.0.0.conv.conv."
                    net.sample[i] = self._conv(sd[q + "weight"], sd[q + "bias"], 1, False, state_key)
                else:
                    q = prefix + f"upsample_layers.{i}.0.convtr.convtr."
                    net.sample[i] = self._conv(sd[q + "weight"], sd[q + "bias"], ratios[i - 1], True, state_key)
                net.n_blocks[i] = depths[i]
                arr = self._blocks(prefix + f"stages.{i}.", depths[i], ch, state_key)
                net.blocks[i] = C.cast(arr, C.POINTER(L.Block))
        else:
            rr = list(reversed(ratios))
            for i in range(n):
                ch = filters * 2 ** i
                q = prefix + f"downsample_layers.{i}.0.conv.conv."
                net.sample[i] = self._conv(sd[q + "weight"], sd[q + "bias"], 1 if i == 0 else rr[i - 1], False, state_key)
                net.n_blocks[i] = depths_enc[i]
                arr = self._blocks(prefix + f"stages.{i}.", depths_enc[i], ch, state_key)
                net.blocks[i] = C.cast(arr, C.POINTER(L.Block))
        q = prefix + "head.conv.conv."
        net.head = self._conv(sd[q + "weight"], sd[q + "bias"], 1, False, state_key)
        return net

    def _build_connector(self, prefix, din) -> L.Connector:
        sd = self._sd
        c = L.Connector()
        c.wdt, c.din, c.hidden = self.wdt, din, self.cfg.hidden
        c.fc1, c.b1 = L.ptr(self._mat(sd[prefix + "fc1.weight"])), L.ptr(self._vec(sd[prefix + "fc1.bias"]))
        c.norm_w = L.ptr(self._vec(sd[prefix + "norm.weight"]))
        c.fc2, c.b2 = L.ptr(self._mat(sd[prefix + "fc2.weight"])), L.ptr(self._vec(sd[prefix + "fc2.bias"]))
        return c
