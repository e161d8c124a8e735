"""CPU oracle for the VibeVoice per-frame hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain torch-CPU fp32 tensor ops, no nn.Module, no GPU) of the
arithmetic of the reference's inference loop.  It exists so that `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg have something to check / time the HIP path against on a box
where `/root/reference` does not exist.  The product (`vibevoice_rocm_amd/`) never imports it.

Pinning: the reference ships no tests and no golden vectors, so this oracle is
pinned by fixtures generated in the build container from the reference's OWN modules
(`oracle/gen/make_golden.py` -> `tests/golden/*.npz`; checked by `tests/test_oracle_vs_golden.py`).
The Qwen2 arithmetic is third-party (`transformers`, reference pin 4.51.3; fixtures generated with
the installed 5.15.0 whose Qwen2 math is unchanged) and `generate()` itself cannot run under the
installed transformers, so the loop is pinned by a hand-driven trace over the reference's
`forward`/`sample_speech_tokens`/tokenizers that performs the reference's own negative-branch
mask/KV surgery (`oracle/gen/make_golden.py::loop_trace`).

Every function cites the reference file:line (relative to /root/reference) it restates.
Weights are a flat dict {state-dict name: torch.Tensor} using the reference's names
(SURVEY.md Appendix D).  Layout here is the reference's channels-first [C, T] for convs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------------------

def cfg_from_json(j: dict) -> dict:
    """Flatten the reference's config JSON (vibevoice/configs/*.json) into the few numbers the
    arithmetic needs.  configuration_vibevoice.py:14-247."""
    d, h = j["decoder_config"], j["diffusion_head_config"]
    a, s = j["acoustic_tokenizer_config"], j["semantic_tokenizer_config"]

    def depths(x):
        return [int(v) for v in x.split("-")] if isinstance(x, str) else list(x)

    heads = d["num_attention_heads"]
    return dict(
        hidden=d["hidden_size"], inter=d["intermediate_size"], layers=d["num_hidden_layers"],
        heads=heads, kv_heads=d["num_key_value_heads"],
        head_dim=d.get("head_dim") or d["hidden_size"] // heads,
        vocab=d["vocab_size"], rope_theta=float(d.get("rope_theta", 10000.0)),
        rms_eps=float(d.get("rms_norm_eps", 1e-6)), max_pos=d.get("max_position_embeddings", 32768),
        tie=bool(d.get("tie_word_embeddings", False)),
        head_hidden=h["hidden_size"], head_ffn=int(h["hidden_size"] * h.get("head_ffn_ratio", 3.0)),
        head_layers=h.get("head_layers", 4), latent=h.get("latent_size", 64),
        head_eps=float(h.get("rms_norm_eps", 1e-5)),
        ddpm_steps=h.get("ddpm_num_steps", 1000), ddpm_infer=h.get("ddpm_num_inference_steps", 20),
        beta_schedule=h.get("ddpm_beta_schedule", "cosine"), prediction_type=h.get("prediction_type", "v_prediction"),
        ac_filters=a["encoder_n_filters"], ac_dec_filters=a.get("decoder_n_filters", a["encoder_n_filters"]),
        ac_ratios=list(a["encoder_ratios"]), ac_depths=depths(a["encoder_depths"]),
        ac_dim=a["vae_dim"], ac_eps=float(a.get("layernorm_eps", 1e-5)), ac_fix_std=float(a.get("fix_std", 0.5)),
        ac_std_dist=a.get("std_dist_type", "gaussian"),
        sem_filters=s["encoder_n_filters"], sem_ratios=list(s["encoder_ratios"]),
        sem_depths=depths(s["encoder_depths"]), sem_dim=s["vae_dim"], sem_eps=float(s.get("layernorm_eps", 1e-5)),
    )


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------

def rmsnorm(x: Tensor, w: Optional[Tensor], eps: float) -> Tensor:
    """x * rsqrt(mean(x^2) + eps) [* w], fp32.  modular_vibevoice_diffusion_head.py:31-38,
    modular_vibevoice_tokenizer.py:65-72, transformers Qwen2RMSNorm."""
    xf = x.float()
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return out * w if w is not None else out


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return F.linear(x, w, b)


# --------------------------------------------------------------------------------------
# DPM-Solver++ (2M, midpoint, v-prediction)            vibevoice/schedule/dpm_solver.py
# --------------------------------------------------------------------------------------

def cosine_alphas_cumprod(num_train: int = 1000, max_beta: float = 0.999) -> Tensor:
    """betas_for_alpha_bar(cosine) + cumprod.  dpm_solver.py:53-54,78-83,251-252."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    betas = [min(1 - alpha_bar((i + 1) / num_train) / alpha_bar(i / num_train), max_beta) for i in range(num_train)]
    betas = torch.tensor(betas, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def dpm_set_timesteps(alphas_cumprod: Tensor, n: int):
    """timesteps (linspace spacing) and sigmas with final sigma 0.  dpm_solver.py:358-364,385-411."""
    num_train = alphas_cumprod.shape[0]
    last_timestep = num_train  # lambda_min_clipped = -inf -> clipped_idx = 0 (dpm_solver.py:353-354)
    timesteps = np.linspace(0, last_timestep - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    sigmas = (((1 - alphas_cumprod) / alphas_cumprod) ** 0.5).numpy()
    sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
    sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
    return timesteps, torch.from_numpy(sigmas)


def _alpha_sigma(sigma: Tensor):
    """dpm_solver.py:483-487."""
    alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
    return alpha_t, sigma * alpha_t


def dpm_coefficients(sigmas: Tensor, algorithm: str = "dpmsolver++") -> List[dict]:
    """Per-step scalar coefficients, computed with fp32 0-dim tensors exactly as the reference does.
    convert_model_output dpm_solver.py:581-584; first order :669-677; second order midpoint :738-764;
    order selection :976-1003 (solver_order 2, final_sigmas_type zero => last step first order).
    algorithm "sde-dpmsolver++" (main.py:543-548): same update shape with cx = sigma_t/sigma_s e^-h, cd = -alpha_t (1 - e^-2h) and
    a noise term cn = sigma_t sqrt(1 - e^-2h) (dpm_solver.py:680-686 first order, :785-793 second order midpoint)."""
    n = sigmas.shape[0] - 1
    out = []
    for i in range(n):
        a_s0, s_s0 = _alpha_sigma(sigmas[i])
        a_t, s_t = _alpha_sigma(sigmas[i + 1])
        lam_t = torch.log(a_t) - torch.log(s_t)
        lam_s0 = torch.log(a_s0) - torch.log(s_s0)
        h = lam_t - lam_s0
        c = dict(alpha_s=float(a_s0), sigma_s=float(s_s0),          # x0 = alpha_s*x - sigma_s*v
                 cx=float(s_t / s_s0), cd=float(a_t * (torch.exp(-h) - 1.0)),
                 order=1, rinv=0.0, cn=0.0)
        if algorithm == "sde-dpmsolver++":
            c["cx"] = float(s_t / s_s0 * torch.exp(-h))
            c["cd"] = float(-(a_t * (1 - torch.exp(-2.0 * h))))
            c["cn"] = float(s_t * torch.sqrt(1.0 - torch.exp(-2 * h)))
        elif algorithm != "dpmsolver++":
            raise ValueError(algorithm)
        first = i == 0
        last = i == n - 1
        if not (first or last):
            a_s1, s_s1 = _alpha_sigma(sigmas[i - 1])
            lam_s1 = torch.log(a_s1) - torch.log(s_s1)
            h0 = lam_s0 - lam_s1
            r0 = h0 / h
            c["order"] = 2
            c["rinv"] = float(1.0 / r0)
        out.append(c)
    return out


def dpm_step(c: dict, x: Tensor, v: Tensor, m_prev: Optional[Tensor], noise: Optional[Tensor] = None):
    """One scheduler.step in fp32: returns (x_next, x0_pred).  dpm_solver.py:935-1022.  `noise`: the variance noise of the SDE variant."""
    x0 = c["alpha_s"] * x - c["sigma_s"] * v
    if c["order"] == 1:
        x_t = c["cx"] * x - c["cd"] * x0
    else:
        d1 = c["rinv"] * (x0 - m_prev)
        x_t = c["cx"] * x - c["cd"] * x0 - 0.5 * c["cd"] * d1
    if c.get("cn", 0.0) != 0.0:
        x_t = x_t + c["cn"] * noise
    return x_t, x0


# --------------------------------------------------------------------------------------
# diffusion head                        vibevoice/modular/modular_vibevoice_diffusion_head.py
# --------------------------------------------------------------------------------------
HEAD = "model.prediction_head."


def timestep_embedding(t: Tensor, dim: int = 256, max_period: float = 10000.0, bf16_t: bool = False) -> Tensor:
    """[cos | sin] sinusoid.  modular_vibevoice_diffusion_head.py:66-88.
    bf16_t: the two roundings the reference's bf16 run applies around it - `t` arrives cast to the activation dtype
    (modeling_vibevoice_inference.py:703: 949 -> 948, 999 -> 1000, ...) and the embedding is cast back to it (:88)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    t = t.float()
    if bf16_t:
        t = t.to(torch.bfloat16).float()
    args = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return emb.to(torch.bfloat16).float() if bf16_t else emb


def head_forward(W: Dict[str, Tensor], cfg: dict, x: Tensor, t: Tensor, cond: Tensor, bf16_t: bool = False) -> Tensor:
    """VibeVoiceDiffusionHead.forward.  modular_vibevoice_diffusion_head.py:254-280 (HeadLayer :158-161,
    FinalLayer :184-188, TimestepEmbedder :90-93)."""
    p = HEAD
    h = linear(x, W[p + "noisy_images_proj.weight"])
    te = timestep_embedding(t, W[p + "t_embedder.mlp.0.weight"].shape[1], bf16_t=bf16_t)
    te = linear(silu(linear(te, W[p + "t_embedder.mlp.0.weight"])), W[p + "t_embedder.mlp.2.weight"])
    c = linear(cond, W[p + "cond_proj.weight"]) + te
    for l in range(cfg["head_layers"]):
        q = f"{p}layers.{l}."
        shift, scale, gate = linear(silu(c), W[q + "adaLN_modulation.1.weight"]).chunk(3, dim=-1)
        y = rmsnorm(h, W[q + "norm.weight"], cfg["head_eps"]) * (1 + scale) + shift
        y = linear(silu(linear(y, W[q + "ffn.gate_proj.weight"])) * linear(y, W[q + "ffn.up_proj.weight"]),
                   W[q + "ffn.down_proj.weight"])
        h = h + gate * y
    shift, scale = linear(silu(c), W[p + "final_layer.adaLN_modulation.1.weight"]).chunk(2, dim=-1)
    y = rmsnorm(h, None, cfg["head_eps"]) * (1 + scale) + shift
    return linear(y, W[p + "final_layer.linear.weight"])


def sample_speech_tokens(W, cfg, cond: Tensor, ncond: Tensor, noise: Tensor, cfg_scale: float, n_steps: int,
                         tables=None, algorithm: str = "dpmsolver++", sde_noise: Optional[Tensor] = None, bf16_t: bool = False) -> Tensor:
    """CFG DPM-Solver++ sampling loop with INJECTED noise [n, latent] (the reference draws
    randn(2n, latent) on the CPU and only ever uses rows [:n]).  modeling_vibevoice_inference.py:695-708.
    SDE variant: `sde_noise` [n_steps, n, latent] = rows [:n] of the [2n, latent] variance noise scheduler.step draws per step
    (dpm_solver.py:993-998); rows [n:] only ever perturb the half of the batch that the next iteration throws away."""
    if tables is None:
        tables = make_dpm_tables(cfg, n_steps, algorithm)
    timesteps, coefs = tables
    n = cond.shape[0]
    condition = torch.cat([cond, ncond], dim=0).float()
    x = noise.float().clone()
    m_prev = None
    for i, t in enumerate(timesteps):
        combined = torch.cat([x, x], dim=0)
        tt = torch.full((2 * n,), float(t), dtype=torch.float32)
        eps = head_forward(W, cfg, combined, tt, condition, bf16_t=bf16_t)
        ce, ue = eps[:n], eps[n:]
        half = ue + cfg_scale * (ce - ue)
        x, m_prev = dpm_step(coefs[i], x, half, m_prev, None if sde_noise is None else sde_noise[i].float())
    return x


def make_dpm_tables(cfg: dict, n_steps: int, algorithm: str = "dpmsolver++"):
    ac = cosine_alphas_cumprod(cfg["ddpm_steps"])
    timesteps, sigmas = dpm_set_timesteps(ac, n_steps)
    return timesteps, dpm_coefficients(sigmas, algorithm)


# --------------------------------------------------------------------------------------
# connectors                                    vibevoice/modular/modeling_vibevoice.py:58-69
# --------------------------------------------------------------------------------------

def connector(W, prefix: str, x: Tensor) -> Tensor:
    """SpeechConnector: fc1 -> LlamaRMSNorm(eps 1e-6) -> fc2."""
    y = linear(x, W[prefix + "fc1.weight"], W[prefix + "fc1.bias"])
    y = rmsnorm(y, W[prefix + "norm.weight"], 1e-6)
    return linear(y, W[prefix + "fc2.weight"], W[prefix + "fc2.bias"])


# --------------------------------------------------------------------------------------
# streaming causal conv tokenizer          vibevoice/modular/modular_vibevoice_tokenizer.py
# --------------------------------------------------------------------------------------

class ConvState:
    """Explicit per-layer streaming state (replaces VibeVoiceTokenizerStreamingCache :193-256).
    state[name] = tensor [C, ctx].  `zero()` == set_to_zero (:234-241): shapes kept, values 0."""

    def __init__(self):
        self.s: Dict[str, Tensor] = {}

    def zero(self):
        for k in self.s:
            self.s[k] = torch.zeros_like(self.s[k])


def sconv1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, groups: int,
            st: Optional[ConvState], name: str) -> Tensor:
    """SConv1d on x [C, T].  Streaming (_forward_streaming :327-382): ctx = (k-1) - (s-1) previous
    input columns are kept.  Non-streaming (st None; _forward_non_streaming :384-418): left zero pad
    ctx, right zero pad `extra` so the last partial window is covered (:127-133)."""
    k = w.shape[-1]
    ctx = (k - 1) - (stride - 1)
    C, T = x.shape
    if st is not None:
        prev = st.s.get(name)
        if prev is None:
            prev = torch.zeros(C, ctx, dtype=x.dtype)
        full = torch.cat([prev, x], dim=1)
        if ctx > 0:
            st.s[name] = full[:, -ctx:].clone() if full.shape[1] >= ctx else full.clone()
    else:
        n_frames = (T - k + ctx) / stride + 1
        ideal = (math.ceil(n_frames) - 1) * stride + (k - ctx)
        full = F.pad(x, (ctx, ideal - T))
    return F.conv1d(full[None], w, b, stride=stride, groups=groups)[0]


def sconvtr1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, st: Optional[ConvState], name: str) -> Tensor:
    """SConvTranspose1d (causal, trim_right_ratio 1) on x [C, T] -> [C_out, T*stride].
    Streaming (_forward_streaming :478-549): up to k-1 previous inputs cached, conv over
    cat[cache, x], trim k-s on the right, keep the last T*s samples."""
    k = w.shape[-1]
    C, T = x.shape
    if st is not None:
        prev = st.s.get(name)
        if prev is None:
            prev = torch.zeros(C, 0, dtype=x.dtype)
        full = torch.cat([prev, x], dim=1)
        st.s[name] = full[:, -(k - 1):].clone() if full.shape[1] > k - 1 else full.clone()
    else:
        full = x
    y = F.conv_transpose1d(full[None], w, b, stride=stride)[0]
    trim = k - stride
    if trim > 0:
        y = y[:, : y.shape[1] - trim]
    if st is not None and full.shape[1] > T:
        y = y[:, -T * stride:]
    return y


def block1d(W, p: str, x: Tensor, eps: float, st: Optional[ConvState]) -> Tensor:
    """Block1D body as TokenizerEncoder/Decoder.forward_features runs it (:786-804 / :924-942):
    x += gamma * dwconv7(RMSNorm_c(x));  x += ffn_gamma * W2 gelu(W1 RMSNorm_c(x) + b1) + b2."""
    C = x.shape[0]
    n = rmsnorm(x.t(), W[p + "norm.weight"], eps).t()
    m = sconv1d(n, W[p + "mixer.conv.conv.conv.weight"], W[p + "mixer.conv.conv.conv.bias"], 1, C, st, p + "mixer")
    x = x + m * W[p + "gamma"][:, None]
    n = rmsnorm(x.t(), W[p + "ffn_norm.weight"], eps)                       # [T, C]
    y = linear(F.gelu(linear(n, W[p + "ffn.linear1.weight"], W[p + "ffn.linear1.bias"])),
               W[p + "ffn.linear2.weight"], W[p + "ffn.linear2.bias"])
    return x + (y * W[p + "ffn_gamma"]).t()


def tokenizer_decoder(W, cfg: dict, latent: Tensor, st: Optional[ConvState],
                      prefix: str = "model.acoustic_tokenizer.decoder.") -> Tensor:
    """TokenizerDecoder.forward on latent [vae_dim, T] -> wav [1, T*hop].  :914-951.
    Decoder depths are the reversed encoder depths (:1024-1028); ratios as given (:830)."""
    depths = list(reversed(cfg["ac_depths"]))
    ratios = cfg["ac_ratios"]
    x = latent
    for i in range(len(depths)):
        if i == 0:
            q = prefix + "upsample_layers.0.0.conv.conv."
            x = sconv1d(x, W[q + "weight"], W[q + "bias"], 1, 1, st, q)
        else:
            q = prefix + f"upsample_layers.{i}.0.convtr.convtr."
            x = sconvtr1d(x, W[q + "weight"], W[q + "bias"], ratios[i - 1], st, q)
        for j in range(depths[i]):
            x = block1d(W, prefix + f"stages.{i}.{j}.", x, cfg["ac_eps"], st)
    q = prefix + "head.conv.conv."
    return sconv1d(x, W[q + "weight"], W[q + "bias"], 1, 1, st, q)


def tokenizer_encoder(W, ratios_cfg, depths, eps, wav: Tensor, st: Optional[ConvState], prefix: str) -> Tensor:
    """TokenizerEncoder.forward on wav [1, T] -> [vae_dim, T/hop].  :776-813.  Ratios are the
    reversed config list (:701)."""
    ratios = list(reversed(ratios_cfg))
    x = wav
    for i in range(len(depths)):
        q = prefix + f"downsample_layers.{i}.0.conv.conv."
        x = sconv1d(x, W[q + "weight"], W[q + "bias"], 1 if i == 0 else ratios[i - 1], 1, st, q)
        for j in range(depths[i]):
            x = block1d(W, prefix + f"stages.{i}.{j}.", x, eps, st)
    q = prefix + "head.conv.conv."
    return sconv1d(x, W[q + "weight"], W[q + "bias"], 1, 1, st, q)


def semantic_encode(W, cfg, wav, st):
    """VibeVoiceSemanticTokenizerModel.encode(...).mean -> [T/hop, 128].  :1171-1175."""
    return tokenizer_encoder(W, cfg["sem_ratios"], cfg["sem_depths"], cfg["sem_eps"], wav, st,
                             "model.semantic_tokenizer.encoder.").t()


def acoustic_encode(W, cfg, wav, st=None):
    """VibeVoiceAcousticTokenizerModel.encode(...).mean -> [T/hop, 64].  :1081-1085."""
    return tokenizer_encoder(W, cfg["ac_ratios"], cfg["ac_depths"], cfg["ac_eps"], wav, st,
                             "model.acoustic_tokenizer.encoder.").t()


# --------------------------------------------------------------------------------------
# Qwen2 decoder (third-party math: transformers.models.qwen2.modeling_qwen2)
# --------------------------------------------------------------------------------------
LLM = "model.language_model."


class KVCache:
    """Per-layer K/V [n_kv, S, D] (DynamicCache equivalent)."""

    def __init__(self, layers: int):
        self.k: List[Optional[Tensor]] = [None] * layers
        self.v: List[Optional[Tensor]] = [None] * layers

    @property
    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[1]

    def truncate(self, n: int):
        for i in range(len(self.k)):
            if self.k[i] is not None:
                self.k[i] = self.k[i][:, :n]
                self.v[i] = self.v[i][:, :n]


def rope_cos_sin(cfg: dict, positions: Tensor):
    """Qwen2RotaryEmbedding: inv_freq = 1/theta^(2i/d) in fp32, emb = cat(freqs, freqs)."""
    d = cfg["head_dim"]
    inv_freq = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llm_forward(W, cfg: dict, x: Tensor, kv: KVCache, pos_start: int) -> Tensor:
    """Causal Qwen2Model forward of T new embeddings x [T, H] at positions pos_start..; appends to kv;
    returns the final-normed hidden states [T, H].  Reference call sites modeling_vibevoice.py:187-199,
    modeling_vibevoice_inference.py:226-237."""
    T, H = x.shape
    nh, nkv, d = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    pos = torch.arange(pos_start, pos_start + T)
    cos, sin = rope_cos_sin(cfg, pos)                                  # [T, d]
    h = x.float()
    for l in range(cfg["layers"]):
        p = f"{LLM}layers.{l}."
        n = rmsnorm(h, W[p + "input_layernorm.weight"], cfg["rms_eps"])
        q = linear(n, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]).view(T, nh, d).transpose(0, 1)
        k = linear(n, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"]).view(T, nkv, d).transpose(0, 1)
        v = linear(n, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]).view(T, nkv, d).transpose(0, 1)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        kv.k[l] = k if kv.k[l] is None else torch.cat([kv.k[l], k], dim=1)
        kv.v[l] = v if kv.v[l] is None else torch.cat([kv.v[l], v], dim=1)
        S = kv.k[l].shape[1]
        kk = kv.k[l].repeat_interleave(nh // nkv, dim=0)               # [nh, S, d]
        vv = kv.v[l].repeat_interleave(nh // nkv, dim=0)
        att = torch.matmul(q, kk.transpose(1, 2)) * (d ** -0.5)          # [nh, T, S]
        kpos = torch.arange(S)[None, :]
        qpos = (S - T + torch.arange(T))[:, None]
        att = att.masked_fill(kpos > qpos, float("-inf"))
        att = torch.softmax(att, dim=-1, dtype=torch.float32)
        o = torch.matmul(att, vv).transpose(0, 1).reshape(T, nh * d)
        h = h + linear(o, W[p + "self_attn.o_proj.weight"])
        n = rmsnorm(h, W[p + "post_attention_layernorm.weight"], cfg["rms_eps"])
        h = h + linear(silu(linear(n, W[p + "mlp.gate_proj.weight"])) * linear(n, W[p + "mlp.up_proj.weight"]),
                       W[p + "mlp.down_proj.weight"])
    return rmsnorm(h, W[LLM + "norm.weight"], cfg["rms_eps"])


def lm_head_weight(W, cfg):
    """tie_weights: lm_head.weight is embed_tokens.weight when tie_word_embeddings.
    modeling_vibevoice_inference.py:119-128."""
    if cfg["tie"] or "lm_head.weight" not in W:
        return W[LLM + "embed_tokens.weight"]
    return W["lm_head.weight"]


def constrained_argmax(hidden: Tensor, W, cfg, valid_ids: List[int]) -> int:
    """lm_head on the last position, -inf mask outside valid ids, argmax (first max wins).
    modeling_vibevoice_inference.py:53-66,241-242,486-496."""
    ids = sorted(set(valid_ids))
    logits = linear(hidden.float(), lm_head_weight(W, cfg)[ids])
    return ids[int(torch.argmax(logits))]


# --------------------------------------------------------------------------------------
# voice-prompt prefill                    modeling_vibevoice_inference.py:149-163, :221-224
# --------------------------------------------------------------------------------------

def process_speech_inputs(W, cfg, speech_tensors: Tensor, speech_masks: Tensor,
                          std_noise: Optional[Tensor], eps_noise: Optional[Tensor]):
    """acoustic encoder (non-streaming) -> gaussian sample with INJECTED noise -> (x+bias)*scale ->
    acoustic_connector -> rows selected by speech_masks.  speech_tensors [S, Tmax] fp32.
    std_noise [S] and eps_noise [S, F, 64] replace the two randn draws of
    VibeVoiceTokenizerEncoderOutput.sample('gaussian') (modular_vibevoice_tokenizer.py:980-989)."""
    means = torch.stack([acoustic_encode(W, cfg, speech_tensors[i][None].float()) for i in range(speech_tensors.shape[0])])
    if cfg["ac_std_dist"] == "gaussian" and std_noise is not None:
        std = std_noise.float() * (cfg["ac_fix_std"] / 0.8)
        lat = means + std[:, None, None] * eps_noise.float()
    else:
        lat = means
    feats = (lat + W["model.speech_bias_factor"].float()) * W["model.speech_scaling_factor"].float()
    conn = connector(W, "model.acoustic_connector.", feats)
    return feats, conn[speech_masks]


# --------------------------------------------------------------------------------------
# the per-token loop                        modeling_vibevoice_inference.py:364-693 (batch 1)
# --------------------------------------------------------------------------------------

class GenerateResult:
    def __init__(self):
        self.sequences: List[int] = []
        self.audio: List[Tensor] = []
        self.latents: List[Tensor] = []
        self.trace: List[dict] = []
        self.reach_max_step = False


def generate(W, cfg: dict, input_ids: List[int], speech_input_mask: Optional[Tensor], speech_embeds: Optional[Tensor],
             special: dict, noise: Tensor, cfg_scale: float = 1.3, n_steps: int = 10, max_length_times: float = 2.0,
             forced_tokens: Optional[List[int]] = None, max_new_tokens: Optional[int] = None,
             keep_trace: bool = False, bf16_t: bool = False, refresh_negative: bool = True) -> GenerateResult:
    """Batch-1 restatement of generate().  `special` = dict(speech_start, speech_end, speech_diffusion, eos[, bos]).
    `noise` [F, latent] is consumed one row per diffusion frame (replaces the CPU randn at :699).
    `forced_tokens` overrides the argmax (bench / random-weight runs) but the logits
    are still computed.  Negative branch (:377-384, :547-563, :575-587): its context is every
    embedding the positive branch consumed since the last speech_start; the reference's reset
    (mask all slots, unmask only the next one => position 0, attends to nothing else) is a truncation
    to length 0 — pinned against the reference's own mask surgery by the loop_trace fixture.
    `bf16_t`: the timestep roundings of the reference's bf16 run (see timestep_embedding).
    `refresh_negative=False` (:501-515): the negative branch instead consumes EVERY step's input embedding right after the token
    choice (a single speech_start at step 0, where inputs_embeds is still None), is never reset on speech_start, and the
    diffusion branch uses that step's negative hidden state (batch 1: the correction of :588-622 touches no sample)."""
    res = GenerateResult()
    emb = W[LLM + "embed_tokens.weight"]
    ids = list(input_ids)
    L0 = len(ids)
    x0 = emb[torch.tensor(ids)].float().clone()
    if speech_embeds is not None:
        x0[speech_input_mask] = speech_embeds.float()                              # :221-224
    pos_kv, neg_kv = KVCache(cfg["layers"]), KVCache(cfg["layers"])
    ac_state, sem_state = ConvState(), ConvState()
    tables = make_dpm_tables(cfg, n_steps)
    valid = [special["speech_start"], special["speech_end"], special["speech_diffusion"], special["eos"]]
    if special.get("bos") is not None:
        valid.append(special["bos"])
    max_length = cfg["max_pos"] if max_new_tokens is None else L0 + max_new_tokens    # :370-371
    max_steps = min(max_length - L0, int(max_length_times * L0))                    # :420
    inputs_embeds = None
    frame = 0
    for step in range(max_steps):
        if len(ids) >= max_length:                                                 # :452-457
            res.reach_max_step = True
            break
        x_in = x0 if step == 0 else inputs_embeds                                  # [T, H]
        hidden = llm_forward(W, cfg, x_in, pos_kv, pos_kv.length)[-1]              # :478-480
        tok = constrained_argmax(hidden, W, cfg, valid)                            # :486-496
        if forced_tokens is not None and step < len(forced_tokens):
            tok = forced_tokens[step]
        ids.append(tok)
        rec = dict(step=step, token=tok) if keep_trace else None
        finished = tok == special["eos"]                                           # :517-526
        if tok == special["speech_end"]:                                           # :540-544
            ac_state.zero()
            sem_state.zero()
        if not refresh_negative:                                                   # :501-515
            neg_in = x_in[-1:] if step > 0 else emb[special["speech_start"]].float()[None]
            nhidden = llm_forward(W, cfg, neg_in, neg_kv, neg_kv.length)[-1]
        if refresh_negative and not finished and tok == special["speech_start"]:   # :547-563
            neg_kv.truncate(0)
        next_embeds = emb[tok].float()[None]                                       # :567
        if not finished and tok == special["speech_diffusion"]:                    # :571-670
            if refresh_negative:
                neg_in = x_in[-1:] if step > 0 else emb[special["speech_start"]].float()[None]
                nhidden = llm_forward(W, cfg, neg_in, neg_kv, neg_kv.length)[-1]   # :575-587
            latent = sample_speech_tokens(W, cfg, hidden[None], nhidden[None], noise[frame][None], cfg_scale,
                                          n_steps, tables, bf16_t=bf16_t)          # :627-631
            scaled = latent / W["model.speech_scaling_factor"].float() - W["model.speech_bias_factor"].float()
            wav = tokenizer_decoder(W, cfg, scaled.t(), ac_state)                  # [1, 3200]   :634-641
            sem = semantic_encode(W, cfg, wav, sem_state)                          # [1, 128]    :656-662
            next_embeds = connector(W, "model.acoustic_connector.", latent) + \
                connector(W, "model.semantic_connector.", sem)                     # :665-670
            res.audio.append(wav[0])
            res.latents.append(latent[0])
            if keep_trace:
                rec.update(cond=hidden, ncond=nhidden, latent=latent[0], wav=wav[0], sem=sem[0])
            frame += 1
        if keep_trace:
            rec["next_embeds"] = next_embeds[0]
            res.trace.append(rec)
        inputs_embeds = next_embeds                                                # :673
        if finished:
            break
    res.sequences = ids
    return res
