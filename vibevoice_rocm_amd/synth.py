"""State-dict layout (names + shapes) of a VibeVoice checkpoint and a deterministic synthetic-weight
generator for it.

There are no weights offline, so parity fixtures, smoke and bench run on seeded random
weights at the real shapes.  Names follow the reference's module paths (produced by
vibevoice/scripts/convert_nnscaler_checkpoint_to_transformers.py:116-123) so a real checkpoint drops in.

Two generators:
  * `synth_state_dict(cfg, seed)`            numpy PCG64 keyed per tensor name: bit-stable across boxes,
                                             used for the committed fixtures (tiny/mid shapes).
  * `synth_state_dict_torch(cfg, seed, dev)` torch generator on `dev`: fast at 1.5B/7B shapes (bench).
Initialisation differs from the reference's on purpose: the zero-initialised adaLN / final
layers and the 1e-6 layer scales would make most of the network a no-op, so they get O(0.1) values.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import numpy as np

from .config import VVConfig

Shape = Tuple[int, ...]


def _block_shapes(p: str, c: int) -> Iterator[Tuple[str, Shape]]:
    yield p + "gamma", (c,)
    yield p + "ffn_gamma", (c,)
    yield p + "norm.weight", (c,)
    yield p + "mixer.conv.conv.conv.weight", (c, 1, 7)
    yield p + "mixer.conv.conv.conv.bias", (c,)
    yield p + "ffn_norm.weight", (c,)
    yield p + "ffn.linear1.weight", (4 * c, c)
    yield p + "ffn.linear1.bias", (4 * c,)
    yield p + "ffn.linear2.weight", (c, 4 * c)
    yield p + "ffn.linear2.bias", (c,)


def _encoder_shapes(p: str, filt: int, ratios, depths, dim: int) -> Iterator[Tuple[str, Shape]]:
    rr = list(reversed(ratios))                      # modular_vibevoice_tokenizer.py:701
    for i in range(len(depths)):
        c = filt * 2 ** i
        if i == 0:
            yield p + "downsample_layers.0.0.conv.conv.weight", (c, 1, 7)
        else:
            yield p + f"downsample_layers.{i}.0.conv.conv.weight", (c, c // 2, 2 * rr[i - 1])
        yield p + f"downsample_layers.{i}.0.conv.conv.bias", (c,)
        for j in range(depths[i]):
            yield from _block_shapes(p + f"stages.{i}.{j}.", c)
    c = filt * 2 ** (len(depths) - 1)
    yield p + "head.conv.conv.weight", (dim, c, 7)
    yield p + "head.conv.conv.bias", (dim,)


def _decoder_shapes(p: str, filt: int, ratios, depths_enc, dim: int) -> Iterator[Tuple[str, Shape]]:
    depths = list(reversed(depths_enc))              # modular_vibevoice_tokenizer.py:1024-1028
    n = len(depths)
    for i in range(n):
        c = filt * 2 ** (n - 1 - i)
        if i == 0:
            yield p + "upsample_layers.0.0.conv.conv.weight", (c, dim, 7)
            yield p + "upsample_layers.0.0.conv.conv.bias", (c,)
        else:
            yield p + f"upsample_layers.{i}.0.convtr.convtr.weight", (2 * c, c, 2 * ratios[i - 1])
            yield p + f"upsample_layers.{i}.0.convtr.convtr.bias", (c,)
        for j in range(depths[i]):
            yield from _block_shapes(p + f"stages.{i}.{j}.", c)
    yield p + "head.conv.conv.weight", (1, filt, 7)
    yield p + "head.conv.conv.bias", (1,)


def state_dict_shapes(cfg: VVConfig) -> Dict[str, Shape]:
    """Every tensor of a VibeVoiceForConditionalGenerationInference state dict (lm_head.weight only when untied)."""
    s: Dict[str, Shape] = {}
    H, I = cfg.hidden, cfg.inter
    lm = "model.language_model."
    s[lm + "embed_tokens.weight"] = (cfg.vocab, H)
    for l in range(cfg.layers):
        p = f"{lm}layers.{l}."
        s[p + "self_attn.q_proj.weight"] = (cfg.q_dim, H)
        s[p + "self_attn.q_proj.bias"] = (cfg.q_dim,)
        s[p + "self_attn.k_proj.weight"] = (cfg.kv_dim, H)
        s[p + "self_attn.k_proj.bias"] = (cfg.kv_dim,)
        s[p + "self_attn.v_proj.weight"] = (cfg.kv_dim, H)
        s[p + "self_attn.v_proj.bias"] = (cfg.kv_dim,)
        s[p + "self_attn.o_proj.weight"] = (H, cfg.q_dim)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        s[p + "input_layernorm.weight"] = (H,)
        s[p + "post_attention_layernorm.weight"] = (H,)
    s[lm + "norm.weight"] = (H,)
    if not cfg.tie:
        s["lm_head.weight"] = (cfg.vocab, H)
    # diffusion head (no biases anywhere; final_layer.norm_final has no weight)
    hd, D, Fh = "model.prediction_head.", cfg.head_hidden, cfg.head_ffn
    s[hd + "noisy_images_proj.weight"] = (D, cfg.latent)
    s[hd + "cond_proj.weight"] = (D, H)
    s[hd + "t_embedder.mlp.0.weight"] = (D, 256)
    s[hd + "t_embedder.mlp.2.weight"] = (D, D)
    for l in range(cfg.head_layers):
        p = f"{hd}layers.{l}."
        s[p + "ffn.gate_proj.weight"] = (Fh, D)
        s[p + "ffn.up_proj.weight"] = (Fh, D)
        s[p + "ffn.down_proj.weight"] = (D, Fh)
        s[p + "norm.weight"] = (D,)
        s[p + "adaLN_modulation.1.weight"] = (3 * D, D)
    s[hd + "final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
    s[hd + "final_layer.linear.weight"] = (cfg.latent, D)
    # tokenizers
    s.update(_encoder_shapes("model.acoustic_tokenizer.encoder.", cfg.ac_filters, cfg.ac_ratios, cfg.ac_depths, cfg.ac_dim))
    s.update(_decoder_shapes("model.acoustic_tokenizer.decoder.", cfg.ac_dec_filters, cfg.ac_ratios, cfg.ac_depths, cfg.ac_dim))
    s.update(_encoder_shapes("model.semantic_tokenizer.encoder.", cfg.sem_filters, cfg.sem_ratios, cfg.sem_depths, cfg.sem_dim))
    # connectors
    for name, din in (("acoustic", cfg.ac_dim), ("semantic", cfg.sem_dim)):
        p = f"model.{name}_connector."
        s[p + "fc1.weight"] = (H, din)
        s[p + "fc1.bias"] = (H,)
        s[p + "norm.weight"] = (H,)
        s[p + "fc2.weight"] = (H, H)
        s[p + "fc2.bias"] = (H,)
    s["model.speech_scaling_factor"] = ()
    s["model.speech_bias_factor"] = ()
    return s


def _rule(name: str, shape: Shape):
    """(kind, scale) for a tensor: how synthetic values are drawn."""
    if name.endswith("speech_scaling_factor"):
        return "const", 0.2
    if name.endswith("speech_bias_factor"):
        return "const", 0.05
    if name.endswith("gamma"):                                   # gamma / ffn_gamma layer scales
        return "around", (0.1, 0.02)
    if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
        return "around", (1.0, 0.1)
    if name.endswith(".bias"):
        return "normal", 0.02
    if name.endswith("embed_tokens.weight") or name == "lm_head.weight":
        return "normal", 0.02
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if "convtr" in name:                                          # ConvTranspose1d weight is [C_in, C_out, k]
        fan_in = shape[0] * 2                                     # two taps reach each output (k = 2 s)
    return "normal", 0.8 / float(np.sqrt(max(fan_in, 1)))


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def synth_tensor(name: str, shape: Shape, seed: int) -> np.ndarray:
    kind, sc = _rule(name, shape)
    if kind == "const":
        return np.full(shape, sc, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(_seed_for(name, seed)))
    z = rng.standard_normal(shape, dtype=np.float32)
    if kind == "around":
        return (sc[0] + sc[1] * z).astype(np.float32)
    return (sc * z).astype(np.float32)


def synth_state_dict(cfg: VVConfig, seed: int = 1234) -> Dict[str, np.ndarray]:
    return {n: synth_tensor(n, shp, seed) for n, shp in state_dict_shapes(cfg).items()}


def synth_state_dict_torch(cfg: VVConfig, seed: int = 1234, device="cuda", dtype=None):
    """Same rules, drawn with a torch generator on `device` (values differ from the numpy generator)."""
    import torch
    out = {}
    g = torch.Generator(device=device)
    for n, shp in state_dict_shapes(cfg).items():
        kind, sc = _rule(n, shp)
        if kind == "const":
            t = torch.full(shp, sc, dtype=torch.float32, device=device)
        else:
            g.manual_seed(_seed_for(n, seed))
            z = torch.randn(shp, generator=g, device=device, dtype=torch.float32)
            t = sc[0] + sc[1] * z if kind == "around" else sc * z
        keep_fp32 = len(shp) <= 1
        out[n] = t if (dtype is None or keep_fp32) else t.to(dtype)
    return out
