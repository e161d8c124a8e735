"""AudioStreamer / AsyncAudioStreamer with the reference's queue semantics (vibevoice/modular/streamer.py:13-264):
one queue per sample, `put(audio_chunks, sample_indices)` from the generating thread, `end(sample_indices=None)`,
`finished_flags` polled by generate() for cooperative cancellation, iterators for the consuming thread."""
from __future__ import annotations

import asyncio
from queue import Queue, Empty
from typing import Optional

import torch


class AudioStreamer:
    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None):
        self.batch_size = batch_size
        self.stop_signal = stop_signal
        self.timeout = timeout
        self.audio_queues = [Queue() for _ in range(batch_size)]
        self.finished_flags = [False for _ in range(batch_size)]
        self.sample_indices_map = {}

    def put(self, audio_chunks: torch.Tensor, sample_indices: torch.Tensor):
        for i, sample_idx in enumerate(sample_indices):
            idx = int(sample_idx)
            if idx < self.batch_size and not self.finished_flags[idx]:
                self.audio_queues[idx].put(audio_chunks[i].detach().cpu(), timeout=self.timeout)

    def end(self, sample_indices=None):
        idxs = range(self.batch_size) if sample_indices is None else [int(i) for i in sample_indices]
        for idx in idxs:
            if idx < self.batch_size and not self.finished_flags[idx]:
                self.audio_queues[idx].put(self.stop_signal, timeout=self.timeout)
                self.finished_flags[idx] = True

    def __iter__(self):
        return AudioBatchIterator(self)

    def get_stream(self, sample_idx: int):
        if sample_idx >= self.batch_size:
            raise ValueError(f"Sample index {sample_idx} exceeds batch size {self.batch_size}")
        return AudioSampleIterator(self, sample_idx)


class AudioSampleIterator:
    def __init__(self, streamer: AudioStreamer, sample_idx: int):
        self.streamer, self.sample_idx = streamer, sample_idx

    def __iter__(self):
        return self

    def __next__(self):
        value = self.streamer.audio_queues[self.sample_idx].get(timeout=self.streamer.timeout)
        if value is self.streamer.stop_signal:
            raise StopIteration()
        return value


class AudioBatchIterator:
    def __init__(self, streamer: AudioStreamer):
        self.streamer = streamer
        self.active_samples = set(range(streamer.batch_size))

    def __iter__(self):
        return self

    def __next__(self):
        import time
        while self.active_samples:
            batch_chunks, done = {}, set()
            for idx in self.active_samples:
                try:
                    value = self.streamer.audio_queues[idx].get(block=False)
                except Empty:
                    continue
                if value is self.streamer.stop_signal:
                    done.add(idx)
                else:
                    batch_chunks[idx] = value
            self.active_samples -= done
            if batch_chunks:
                return batch_chunks
            if self.active_samples:
                time.sleep(0.01)
        raise StopIteration()


class AsyncAudioStreamer(AudioStreamer):
    """asyncio flavour: queues live on the consumer's event loop, put/end are thread-safe from the generating thread."""

    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None):
        super().__init__(batch_size, stop_signal, timeout)
        self.audio_queues = [asyncio.Queue() for _ in range(batch_size)]
        self.loop = asyncio.get_running_loop()

    def put(self, audio_chunks: torch.Tensor, sample_indices: torch.Tensor):
        for i, sample_idx in enumerate(sample_indices):
            idx = int(sample_idx)
            if idx < self.batch_size and not self.finished_flags[idx]:
                self.loop.call_soon_threadsafe(self.audio_queues[idx].put_nowait, audio_chunks[i].detach().cpu())

    def end(self, sample_indices=None):
        idxs = range(self.batch_size) if sample_indices is None else [int(i) for i in sample_indices]
        for idx in idxs:
            if idx < self.batch_size and not self.finished_flags[idx]:
                self.loop.call_soon_threadsafe(self.audio_queues[idx].put_nowait, self.stop_signal)
                self.finished_flags[idx] = True

    async def get_stream(self, sample_idx: int):
        if sample_idx >= self.batch_size:
            raise ValueError(f"Sample index {sample_idx} exceeds batch size {self.batch_size}")
        while True:
            value = await self.audio_queues[sample_idx].get()
            if value is self.stop_signal:
                break
            yield value

    def __aiter__(self):
        raise NotImplementedError("iterate per sample with get_stream(idx)")
