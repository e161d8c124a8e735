"""Session-local import shim so the reference's own modules can run in THIS container
(fixture generation only; never shipped to the GPU box, never imported by the product).

The reference pins transformers==4.51.3 + diffusers; this image has transformers 5.x and no
diffusers.  The shim only patches plumbing: AutoModel.register collisions,
diffusers base classes (no arithmetic) and the removed `tokenization_qwen2_fast` module.  All
arithmetic executed afterwards is the reference's own source under /root/reference.
"""
import sys, types, functools, inspect, enum, dataclasses, json, os
import torch

REF_ROOT = os.environ.get("VV_REFERENCE_ROOT", "/root/reference")


def install():
    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    from transformers.models.auto import AutoModel, AutoModelForCausalLM

    def _patch(A):
        orig = A.register.__func__
        A.register = classmethod(lambda cls, cfg, model, exist_ok=False: orig(cls, cfg, model, exist_ok=True))

    for A in (AutoModel, AutoModelForCausalLM):
        _patch(A)

    def _mod(n):
        m = types.ModuleType(n)
        sys.modules[n] = m
        return m

    _mod("diffusers")
    cu = _mod("diffusers.configuration_utils")
    ut = _mod("diffusers.utils")
    tu = _mod("diffusers.utils.torch_utils")
    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    class ConfigMixin:
        def register_to_config(self, **kw):
            self.config.update(kw)

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **kw):
            ba = inspect.signature(init).bind(self, *a, **kw)
            ba.apply_defaults()
            self.config = _Cfg({k: v for k, v in ba.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return inner

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    ut.deprecate = lambda *a, **k: None
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)

    class KarrasDiffusionSchedulers(enum.Enum):
        DPMSolverMultistepScheduler = 1

    class SchedulerMixin:
        pass

    @dataclasses.dataclass
    class SchedulerOutput:
        prev_sample: torch.Tensor

    su.KarrasDiffusionSchedulers = KarrasDiffusionSchedulers
    su.SchedulerMixin = SchedulerMixin
    su.SchedulerOutput = SchedulerOutput
    from transformers.models.qwen2 import tokenization_qwen2 as _tq
    _mod("transformers.models.qwen2.tokenization_qwen2_fast").Qwen2TokenizerFast = _tq.Qwen2Tokenizer
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    install._done = True


def load_ref_config(name="qwen2.5_1.5b_64k.json", **overrides):
    """Build the reference's VibeVoiceConfig from its in-repo JSON (model_type keys dropped)."""
    install()
    from vibevoice.modular.configuration_vibevoice import VibeVoiceConfig
    with open(os.path.join(REF_ROOT, "vibevoice", "configs", name)) as f:
        cfg = json.load(f)

    def strip(d, keep_model_type=False):
        d = dict(d)
        if not keep_model_type:
            d.pop("model_type", None)
        d.pop("_attn_implementation_autoset", None)
        return d

    cfg = strip(cfg)
    for k in ("acoustic_tokenizer_config", "semantic_tokenizer_config", "decoder_config", "diffusion_head_config"):
        cfg[k] = strip(cfg[k], keep_model_type=(k == "decoder_config"))
        cfg[k].update(overrides.get(k, {}))
    for k, v in overrides.items():
        if not k.endswith("_config"):
            cfg[k] = v
    return VibeVoiceConfig(**cfg)
