"""MI355X-native VibeVoice inference hot path (hand-written HIP kernels behind a C ABI, Python host)."""
from .config import VVConfig  # noqa: F401

__all__ = ["VVConfig"]
