"""Streaming delivery of generated audio: `AudioStreamer` (thread consumer) and `AsyncAudioStreamer` (asyncio consumer).

The caller-visible contract is the reference's (vibevoice/modular/streamer.py:13-264): one FIFO per sample in `audio_queues`,
`put(audio_chunks, sample_indices)` and `end(sample_indices=None)` called from the generating thread, `finished_flags` polled by
generate() for cooperative cancellation, `get_stream(i)` / iteration on the consuming side, `stop_signal` as the end marker.
The machinery behind it is this package's own: both flavours share one producer side (`_ChunkFanout`) and differ only in how an
item reaches a queue (`_push`); the consumers are generators; the async batch iterator keeps ONE pending `get()` per live sample
across calls instead of creating and cancelling a task per sample per call (a cancelled `Queue.get()` that had already been handed
its item drops that chunk).
"""
from __future__ import annotations

import asyncio
import queue
import time
from typing import Dict, Iterable, Iterator, Optional

import torch


def _as_indices(sample_indices, n: int) -> Iterable[int]:
    if sample_indices is None:
        return range(n)
    return [int(i) for i in sample_indices]


class _ChunkFanout:
    """Producer side shared by both streamers: routes chunk i of a put() to the FIFO of sample_indices[i] until that sample ended."""

    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None):
        self.batch_size = int(batch_size)
        self.stop_signal = stop_signal
        self.timeout = timeout
        self.finished_flags = [False] * self.batch_size
        self.audio_queues = [self._new_queue() for _ in range(self.batch_size)]

    def _new_queue(self):
        raise NotImplementedError

    def _push(self, idx: int, item) -> None:
        raise NotImplementedError

    def _live(self, idx: int) -> bool:
        return 0 <= idx < self.batch_size and not self.finished_flags[idx]

    def put(self, audio_chunks: torch.Tensor, sample_indices) -> None:
        for chunk, idx in zip(audio_chunks, _as_indices(sample_indices, self.batch_size)):
            if self._live(idx):
                self._push(idx, chunk.detach().cpu())

    def end(self, sample_indices=None) -> None:
        for idx in _as_indices(sample_indices, self.batch_size):
            if self._live(idx):
                self._push(idx, self.stop_signal)
                self.finished_flags[idx] = True

    def _check(self, sample_idx: int) -> None:
        if sample_idx >= self.batch_size:
            raise ValueError(f"Sample index {sample_idx} exceeds batch size {self.batch_size}")


class AudioStreamer(_ChunkFanout):
    """Thread flavour: `for chunk in streamer.get_stream(i)` blocks on sample i's FIFO; `for d in streamer` yields {sample: chunk}
    dicts of whatever is ready, until every sample has ended."""

    def _new_queue(self):
        return queue.Queue()

    def _push(self, idx, item):
        self.audio_queues[idx].put(item, timeout=self.timeout)

    def get_stream(self, sample_idx: int) -> Iterator[torch.Tensor]:
        self._check(sample_idx)
        return self._drain(sample_idx)

    def _drain(self, sample_idx):
        q = self.audio_queues[sample_idx]
        while True:
            item = q.get(timeout=self.timeout)
            if item is self.stop_signal:
                return
            yield item

    def __iter__(self) -> Iterator[Dict[int, torch.Tensor]]:
        live = set(range(self.batch_size))
        while live:
            ready: Dict[int, torch.Tensor] = {}
            for idx in sorted(live):
                try:
                    item = self.audio_queues[idx].get_nowait()
                except queue.Empty:
                    continue
                if item is self.stop_signal:
                    live.discard(idx)
                else:
                    ready[idx] = item
            if ready:
                yield ready
            elif live:
                time.sleep(0.01)


class AsyncAudioStreamer(_ChunkFanout):
    """asyncio flavour: must be constructed inside the consumer's running event loop; put/end are safe from the generating thread
    (items are handed over with call_soon_threadsafe).  `async for chunk in streamer.get_stream(i)`, or `async for d in streamer`
    for {sample: chunk} dicts across the batch."""

    def __init__(self, batch_size: int, stop_signal=None, timeout: Optional[float] = None):
        self.loop = asyncio.get_running_loop()
        super().__init__(batch_size, stop_signal, timeout)

    def _new_queue(self):
        return asyncio.Queue()

    def _push(self, idx, item):
        self.loop.call_soon_threadsafe(self.audio_queues[idx].put_nowait, item)

    async def get_stream(self, sample_idx: int):
        self._check(sample_idx)
        q = self.audio_queues[sample_idx]
        while True:
            item = await q.get()
            if item is self.stop_signal:
                return
            yield item

    def __aiter__(self):
        return self._batches()

    async def _batches(self):
        pending = {asyncio.ensure_future(self.audio_queues[i].get()): i for i in range(self.batch_size)}
        try:
            while pending:
                done, _ = await asyncio.wait(pending.keys(), return_when=asyncio.FIRST_COMPLETED, timeout=self.timeout)
                if not done:
                    raise asyncio.TimeoutError(f"no audio chunk within {self.timeout} s")
                ready: Dict[int, torch.Tensor] = {}
                for fut in done:
                    idx = pending.pop(fut)
                    item = fut.result()
                    if item is self.stop_signal:
                        continue
                    ready[idx] = item
                    pending[asyncio.ensure_future(self.audio_queues[idx].get())] = idx
                if ready:
                    yield ready
        finally:
            for fut in pending:
                fut.cancel()
