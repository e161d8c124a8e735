from vibevoice_rocm_amd.processor import VibeVoiceProcessor  # noqa: F401
