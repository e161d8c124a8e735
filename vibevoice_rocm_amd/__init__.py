"""MI355X-native VibeVoice inference hot path (hand-written HIP kernels behind a C ABI, Python host)."""
import os as _os

# Batched / concurrent dialogues run one HIP stream per dialogue; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) and a batch of 4 lanes plus the copy stream oversubscribes them (measured: batch-of-4 generate 50 -> 69 audio-s/s
# with 8 queues).  Only a default: an explicit setting wins, and it is read when the HIP runtime initialises, so it has no effect in
# a process that already touched the GPU.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import VVConfig  # noqa: F401,E402

__all__ = ["VVConfig"]
