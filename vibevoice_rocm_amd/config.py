"""Model shape/hyper-parameter container for the VibeVoice hot path.

Mirrors what the reference's config classes carry (vibevoice/modular/configuration_vibevoice.py:14-247,
JSON layout of vibevoice/configs/qwen2.5_1.5b_64k.json / qwen2.5_7b_32k.json) reduced to the numbers the
kernels need.  `VVConfig.from_json_dict` accepts the reference's `config.json` schema unchanged, so a
checkpoint directory written by the reference's converter loads as-is.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field, asdict
from typing import List


def _depths(x) -> List[int]:
    return [int(v) for v in x.split("-")] if isinstance(x, str) else list(x)


@dataclass
class VVConfig:
    # Qwen2 decoder (decoder_config)
    hidden: int = 1536
    inter: int = 8960
    layers: int = 28
    heads: int = 12
    kv_heads: int = 2
    head_dim: int = 128
    vocab: int = 151936
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    max_pos: int = 65536
    tie: bool = True
    # diffusion head (diffusion_head_config)
    head_hidden: int = 1536
    head_ffn: int = 4608
    head_layers: int = 4
    latent: int = 64
    head_eps: float = 1e-5
    ddpm_steps: int = 1000
    ddpm_infer: int = 20
    beta_schedule: str = "cosine"
    prediction_type: str = "v_prediction"
    # acoustic tokenizer
    ac_filters: int = 32
    ac_dec_filters: int = 32
    ac_ratios: List[int] = field(default_factory=lambda: [8, 5, 5, 4, 2, 2])
    ac_depths: List[int] = field(default_factory=lambda: [3, 3, 3, 3, 3, 3, 8])
    ac_dim: int = 64
    ac_eps: float = 1e-5
    ac_fix_std: float = 0.5
    ac_std_dist: str = "gaussian"
    # semantic tokenizer
    sem_filters: int = 32
    sem_ratios: List[int] = field(default_factory=lambda: [8, 5, 5, 4, 2, 2])
    sem_depths: List[int] = field(default_factory=lambda: [3, 3, 3, 3, 3, 3, 8])
    sem_dim: int = 128
    sem_eps: float = 1e-5

    # ---- derived ----
    @property
    def hop(self) -> int:
        h = 1
        for r in self.ac_ratios:
            h *= r
        return h

    @property
    def q_dim(self) -> int:
        return self.heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.kv_heads * self.head_dim

    def as_dict(self) -> dict:
        return asdict(self)

    # ---- constructors ----
    @classmethod
    def from_json_dict(cls, j: dict) -> "VVConfig":
        d, h = j["decoder_config"], j["diffusion_head_config"]
        a, s = j["acoustic_tokenizer_config"], j["semantic_tokenizer_config"]
        heads = d["num_attention_heads"]
        if a.get("mixer_layer", "depthwise_conv") != "depthwise_conv" or a.get("layernorm", "RMSNorm") != "RMSNorm" \
                or a.get("conv_norm", "none") != "none" or not a.get("causal", True):
            raise NotImplementedError("only the shipped tokenizer variant (causal depthwise conv + RMSNorm, conv_norm none) is built")
        if h.get("prediction_type", "v_prediction") != "v_prediction" or h.get("ddpm_beta_schedule", "cosine") != "cosine":
            raise NotImplementedError("only cosine / v_prediction DPM-Solver++ is built")
        return cls(
            hidden=d["hidden_size"], inter=d["intermediate_size"], layers=d["num_hidden_layers"], heads=heads,
            kv_heads=d["num_key_value_heads"], head_dim=d.get("head_dim") or d["hidden_size"] // heads,
            vocab=d["vocab_size"], rope_theta=float(d.get("rope_theta", 10000.0)),
            rms_eps=float(d.get("rms_norm_eps", 1e-6)), max_pos=d.get("max_position_embeddings", 32768),
            # the inference class ties on the TOP-LEVEL key (modeling_vibevoice_inference.py:119-128), which the converter writes from
            # decoder_config's (scripts/convert_nnscaler_checkpoint_to_transformers.py:46-50); the in-repo init JSONs carry only the latter
            tie=bool(j["tie_word_embeddings"] if "tie_word_embeddings" in j else d.get("tie_word_embeddings", False)),
            head_hidden=h["hidden_size"], head_ffn=int(h["hidden_size"] * h.get("head_ffn_ratio", 3.0)),
            head_layers=h.get("head_layers", 4), latent=h.get("latent_size", 64),
            head_eps=float(h.get("rms_norm_eps", 1e-5)), ddpm_steps=h.get("ddpm_num_steps", 1000),
            ddpm_infer=h.get("ddpm_num_inference_steps", 20),
            ac_filters=a["encoder_n_filters"], ac_dec_filters=a.get("decoder_n_filters", a["encoder_n_filters"]),
            ac_ratios=list(a["encoder_ratios"]), ac_depths=_depths(a["encoder_depths"]), ac_dim=a["vae_dim"],
            ac_eps=float(a.get("layernorm_eps", 1e-5)), ac_fix_std=float(a.get("fix_std", 0.5)),
            ac_std_dist=a.get("std_dist_type", "gaussian"),
            sem_filters=s["encoder_n_filters"], sem_ratios=list(s["encoder_ratios"]),
            sem_depths=_depths(s["encoder_depths"]), sem_dim=s["vae_dim"], sem_eps=float(s.get("layernorm_eps", 1e-5)),
        )

    @classmethod
    def from_pretrained(cls, path: str) -> "VVConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_json_dict(json.load(f))

    @classmethod
    def preset(cls, name: str) -> "VVConfig":
        """Shapes of the two shipped models (vibevoice/configs/qwen2.5_1.5b_64k.json, qwen2.5_7b_32k.json)
        and a tiny shape used by the parity fixtures."""
        name = name.lower()
        if name in ("1.5b", "vibevoice-1.5b"):
            return cls()
        if name in ("7b", "vibevoice-7b", "vibevoice-7b-preview"):
            return cls(hidden=3584, inter=18944, layers=28, heads=28, kv_heads=4, head_dim=128, vocab=152064,
                       max_pos=32768, tie=False, head_hidden=3584, head_ffn=10752)
        if name == "tiny":
            return cls(hidden=64, inter=128, layers=2, heads=4, kv_heads=2, head_dim=16, vocab=160, max_pos=512,
                       tie=True, head_hidden=64, head_ffn=192, head_layers=2, ac_filters=2, ac_dec_filters=2,
                       ac_depths=[1, 1, 1, 1, 1, 1, 2], sem_filters=2, sem_depths=[1, 1, 1, 1, 1, 1, 2])
        if name == "mid":   # real head_dim / GQA ratio, small widths: GPU parity at non-trivial sizes
            return cls(hidden=512, inter=1536, layers=3, heads=4, kv_heads=2, head_dim=128, vocab=1024, max_pos=4096,
                       tie=True, head_hidden=512, head_ffn=1536, head_layers=2, ac_filters=8, ac_dec_filters=8,
                       ac_depths=[1, 1, 1, 1, 1, 1, 2], sem_filters=8, sem_depths=[1, 1, 1, 1, 1, 1, 2])
        raise KeyError(name)

    def to_reference_json(self) -> dict:
        """The reference's config.json schema for these shapes (used by the fixture generator to build the
        reference's own modules at the same shape, and by save_pretrained)."""
        tok = lambda filt, dfilt, ratios, depths, dim, eps, fix, dist: dict(  # noqa: E731
            causal=True, channels=1, conv_bias=True, conv_norm="none", corpus_normalize=0.0,
            encoder_depths="-".join(str(x) for x in depths), encoder_n_filters=filt, encoder_ratios=list(ratios),
            fix_std=fix, layer_scale_init_value=1e-6, layernorm="RMSNorm", layernorm_elementwise_affine=True,
            layernorm_eps=eps, mixer_layer="depthwise_conv", pad_mode="constant", std_dist_type=dist, vae_dim=dim,
            weight_init_value=0.01, disable_last_norm=True, **({} if dfilt is None else dict(
                decoder_n_filters=dfilt, decoder_ratios=list(ratios), decoder_depths=None)))
        return dict(
            acoustic_vae_dim=self.ac_dim, semantic_vae_dim=self.sem_dim,
            acoustic_tokenizer_config=tok(self.ac_filters, self.ac_dec_filters, self.ac_ratios, self.ac_depths,
                                          self.ac_dim, self.ac_eps, self.ac_fix_std, self.ac_std_dist),
            semantic_tokenizer_config=tok(self.sem_filters, None, self.sem_ratios, self.sem_depths, self.sem_dim,
                                          self.sem_eps, 0, "none"),
            decoder_config=dict(
                model_type="qwen2", attention_dropout=0.0, hidden_act="silu", hidden_size=self.hidden,
                initializer_range=0.02, intermediate_size=self.inter, max_position_embeddings=self.max_pos,
                max_window_layers=self.layers, num_attention_heads=self.heads, num_hidden_layers=self.layers,
                num_key_value_heads=self.kv_heads, rms_norm_eps=self.rms_eps, rope_scaling=None,
                rope_theta=self.rope_theta, sliding_window=None, tie_word_embeddings=self.tie, use_cache=True,
                use_sliding_window=False, vocab_size=self.vocab,
                **({"head_dim": self.head_dim} if self.head_dim * self.heads != self.hidden else {})),
            diffusion_head_config=dict(
                ddpm_batch_mul=4, ddpm_beta_schedule=self.beta_schedule, ddpm_num_inference_steps=self.ddpm_infer,
                ddpm_num_steps=self.ddpm_steps, diffusion_type="ddpm", head_ffn_ratio=self.head_ffn / self.head_hidden,
                head_layers=self.head_layers, hidden_size=self.head_hidden, latent_size=self.latent,
                prediction_type=self.prediction_type, rms_norm_eps=self.head_eps, speech_vae_dim=self.latent),
        )
