from vibevoice_rocm_amd.modeling import VibeVoiceForConditionalGenerationInference, VibeVoiceGenerationOutput  # noqa: F401
