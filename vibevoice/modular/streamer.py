from vibevoice_rocm_amd.streamer import AudioStreamer, AsyncAudioStreamer  # noqa: F401
