"""ctypes binding of libvv_hip.so (C ABI in include/vv_hip.h).

The HIP extension IS the product's compute path: importing this module without a built library raises, and no
CPU fallback exists anywhere in the package.  Build with `python -m vibevoice_rocm_amd.build` (or
`__graft_entry__.build()`); the .so lives in-tree next to this file.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvv_hip.so")

VV_F32, VV_BF16, VV_FP8 = 0, 1, 2
PRO_NONE, PRO_RMSNORM, PRO_SILU = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_SWIGLU = 0, 1, 2
LIN_X_BF16, LIN_OUT_BF16, LIN_W_REUSED, LIN_W_FRAG = 1, 2, 4, 8
VV_MAX_STAGES = 8

vp = C.c_void_p
i64 = C.c_int64


class W8(C.Structure):      # vv_w8: optional fp8 (e4m3fn) companion of a matrix + per-output-row scale
    _fields_ = [("q", vp), ("scale", vp)]


class LinArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("m", C.c_int), ("pro", C.c_int), ("norm_w", vp), ("eps", C.c_float),
                ("mod_shift", vp), ("mod_scale", vp), ("ld_mod", i64), ("w", vp), ("w2", vp), ("bias", vp),
                ("n", C.c_int), ("k", C.c_int), ("wdt", C.c_int), ("act", C.c_int), ("gate", vp), ("gate_ld", i64),
                ("res", vp), ("ldres", i64), ("out", vp), ("ldo", i64), ("flags", C.c_int), ("wscale", vp), ("w2scale", vp)]


class KV(C.Structure):
    _fields_ = [("k", vp), ("v", vp), ("kvdt", C.c_int), ("layers", C.c_int), ("rows", C.c_int),
                ("kv_heads", C.c_int), ("s_max", C.c_int), ("head_dim", C.c_int), ("vt", vp)]


class LlmLayer(C.Structure):
    _fields_ = [("ln1", vp), ("ln2", vp), ("wqkv", vp), ("bqkv", vp), ("wo", vp), ("wgate", vp), ("wup", vp), ("wdown", vp),
                ("q_qkv", W8), ("q_o", W8), ("q_gate", W8), ("q_up", W8), ("q_down", W8),
                ("f_qkv", vp), ("f_o", vp), ("f_gate", vp), ("f_up", vp), ("f_down", vp)]


class Llm(C.Structure):
    _fields_ = [("wdt", C.c_int), ("hidden", C.c_int), ("inter", C.c_int), ("layers", C.c_int), ("heads", C.c_int),
                ("kv_heads", C.c_int), ("head_dim", C.c_int), ("rms_eps", C.c_float), ("inv_freq", vp),
                ("final_norm", vp), ("layer", C.POINTER(LlmLayer))]


class HeadLayer(C.Structure):
    _fields_ = [("norm_w", vp), ("wgate", vp), ("wup", vp), ("wdown", vp), ("adaln", vp), ("q_gate", W8), ("q_up", W8), ("q_down", W8),
                ("f_gate", vp), ("f_up", vp), ("f_down", vp)]


class Head(C.Structure):
    _fields_ = [("wdt", C.c_int), ("D", C.c_int), ("ffn", C.c_int), ("layers", C.c_int), ("latent", C.c_int),
                ("cond_dim", C.c_int), ("eps", C.c_float), ("flags", C.c_int), ("noisy_proj", vp), ("cond_proj", vp), ("final_adaln", vp),
                ("final_linear", vp), ("layer", C.POINTER(HeadLayer)), ("fused_g", vp)]


class DpmCoef(C.Structure):
    _fields_ = [("alpha_s", C.c_float), ("sigma_s", C.c_float), ("cx", C.c_float), ("cd", C.c_float),
                ("rinv", C.c_float), ("order", C.c_int), ("cn", C.c_float)]


class Block(C.Structure):
    _fields_ = [("gamma", vp), ("ffn_gamma", vp), ("norm_w", vp), ("ffn_norm_w", vp), ("dw_w", vp), ("dw_b", vp),
                ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("hist", vp), ("q_w1", W8), ("q_w2", W8), ("dw_last", vp), ("hs", vp)]


class Conv(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("cin", C.c_int), ("cout", C.c_int), ("kk", C.c_int), ("stride", C.c_int),
                ("transposed", C.c_int), ("state", vp)]


class ConvNet(C.Structure):
    _fields_ = [("wdt", C.c_int), ("n_stages", C.c_int), ("eps", C.c_float), ("sample", Conv * VV_MAX_STAGES),
                ("n_blocks", C.c_int * VV_MAX_STAGES), ("blocks", C.POINTER(Block) * VV_MAX_STAGES), ("head", Conv)]


class Connector(C.Structure):
    _fields_ = [("wdt", C.c_int), ("din", C.c_int), ("hidden", C.c_int), ("fc1", vp), ("b1", vp), ("norm_w", vp),
                ("fc2", vp), ("b2", vp)]


class ProfEntry(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("dual", C.c_int), ("wdt", C.c_int), ("count", C.c_int),
                ("total_ms", C.c_double)]


_STRUCTS = dict(vv_w8=W8, vv_prof_entry=ProfEntry, vv_lin_args=LinArgs, vv_kv=KV, vv_llm_layer=LlmLayer, vv_llm=Llm, vv_head_layer=HeadLayer, vv_head=Head,
                vv_dpm_coef=DpmCoef, vv_block=Block, vv_conv=Conv, vv_convnet=ConvNet, vv_connector=Connector)

# name -> (restype, argtypes); every symbol include/vv_hip.h declares
PROTOTYPES = {
    "vv_last_error": (C.c_char_p, []),
    "vv_abi_version": (C.c_int, []),
    "vv_init": (C.c_int, []),
    "vv_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "vv_linear": (C.c_int, [C.POINTER(LinArgs), vp]),
    "vv_rope_table": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "vv_rope_store": (C.c_int, [vp, i64, C.c_int, C.c_int, C.POINTER(KV), C.c_int, vp, vp, vp, vp]),
    "vv_attn": (C.c_int, [vp, i64, C.c_int, C.c_int, C.POINTER(KV), C.c_int, vp, vp, vp, i64, vp]),
    "vv_attn_decode": (C.c_int, [vp, i64, C.c_int, C.c_int, C.POINTER(KV), C.c_int, vp, vp, vp, i64, vp]),
    "vv_block_mixer": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_float, vp, vp, vp, vp, vp]),
    "vv_block1d": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "vv_block_mid_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vv_block_mid": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "vv_conv_ctx": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "vv_affine": (C.c_int, [vp, C.c_float, C.c_float, vp, i64, vp]),
    "vv_add_rows": (C.c_int, [vp, i64, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp]),
    "vv_add_rows_silu": (C.c_int, [vp, i64, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp]),
    "vv_add_rows_silu_bf16": (C.c_int, [vp, i64, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp]),
    "vv_dpm_proj": (C.c_int, [vp, i64, C.c_float, C.POINTER(DpmCoef), vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, vp, vp]),
    "vv_embed_row": (C.c_int, [vp, C.c_int, i64, vp, vp, vp]),
    "vv_gather_rows": (C.c_int, [vp, C.c_int, i64, C.POINTER(C.c_int), C.c_int, vp, vp]),
    "vv_argmax_ids": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    "vv_cast_rows_bf16": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, i64, vp]),
    "vv_copy_rows": (C.c_int, [vp, i64, vp, i64, C.c_int, C.c_int, vp]),
    "vv_dpm_step": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                              C.c_float, C.c_int, vp, vp, vp]),
    "vv_advance_lens": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "vv_llm_ws_bytes": (C.c_size_t, [C.POINTER(Llm), C.c_int]),
    "vv_llm_forward": (C.c_int, [C.POINTER(Llm), C.POINTER(KV), vp, i64, C.c_int, vp, vp, vp, i64, vp, vp]),
    "vv_llm_tail": (C.c_int, [C.POINTER(Llm), vp, i64, C.c_int, vp, i64, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "vv_llm_tail_batch": (C.c_int, [C.POINTER(Llm), vp, i64, C.c_int, vp, i64, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "vv_head_ws_bytes": (C.c_size_t, [C.POINTER(Head), C.c_int]),
    "vv_head_ws_bytes_batch": (C.c_size_t, [C.POINTER(Head), C.c_int, C.c_int]),
    "vv_head_sample_batch": (C.c_int, [C.POINTER(Head), vp, i64, vp, i64, vp, C.POINTER(DpmCoef), C.c_int, C.c_float, vp, i64, C.c_int, vp, vp]),
    "vv_head_sample": (C.c_int, [C.POINTER(Head), vp, i64, vp, vp, C.POINTER(DpmCoef), C.c_int, C.c_float, vp, vp, vp, vp]),
    "vv_head_forward": (C.c_int, [C.POINTER(Head), vp, vp, vp, C.c_int, vp, vp, vp]),
    "vv_convnet_ws_bytes": (C.c_size_t, [C.POINTER(ConvNet), i64, C.c_int]),
    "vv_decoder_forward": (C.c_int, [C.POINTER(ConvNet), vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]),
    "vv_encoder_forward": (C.c_int, [C.POINTER(ConvNet), vp, i64, vp, vp, vp]),
    "vv_convnet_reset": (C.c_int, [C.POINTER(ConvNet), vp]),
    "vv_connector_forward": (C.c_int, [C.POINTER(Connector), vp, C.c_int, vp, C.c_int, vp, vp]),
    "vv_connector_pair": (C.c_int, [C.POINTER(Connector), C.POINTER(Connector), vp, vp, vp, i64, C.c_int, vp, vp]),
    "vv_graph_begin": (C.c_int, [vp]),
    "vv_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "vv_graph_launch": (C.c_int, [vp, vp]),
    "vv_graph_destroy": (C.c_int, [vp]),
    "vv_sizeof": (C.c_size_t, [C.c_char_p]),
    "vv_prof_begin": (C.c_int, [C.c_int]),
    "vv_prof_end": (C.c_int, [C.POINTER(ProfEntry), C.c_int, C.POINTER(C.c_int)]),
}


class VVError(RuntimeError):
    pass


_lib = None


def load():
    """Load libvv_hip.so (once).  Raises VVError when it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64): it must be in the process BEFORE our library is loaded so both
    # share one runtime instance (loading ours first would bind it to /opt/rocm's copy and split the process in two)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise VVError(f"{LIB_PATH} is missing: build the HIP extension first (python -m vibevoice_rocm_amd.build). "
                      "The MI355X kernels are the only compute path of this package.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    for cname, cls in _STRUCTS.items():
        n = lib.vv_sizeof(cname.encode())
        if n != C.sizeof(cls):
            raise VVError(f"ABI mismatch for {cname}: C side {n} bytes, ctypes mirror {C.sizeof(cls)} bytes")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise VVError(f"{what} failed ({rc}): {load().vv_last_error().decode()}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()
