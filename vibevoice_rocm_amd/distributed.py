"""Multi-GPU sharding of independent dialogues: one process per GPU, one resident replica each,
no collective inside the generation loop.  Two collectives exist, both outside the per-frame path:
  * `broadcast_state_dict`  rank `src` owns the checkpoint; replicas receive it as two packed blobs (matrices in the
    weight dtype, vectors in fp32): two large collectives instead of ~800 small ones.  Default: one RCCL broadcast per
    blob.  Opt-in (`VV_BCAST=scatter_allgather`): scatter + all-gather, where on a fully connected xGMI node the root
    pushes 1/N of the blob down each of its links and the peers exchange the pieces over theirs instead of a ring-bound
    broadcast (7 links x ~153 GB/s per GPU, point to point) - not yet measured on an 8-GPU node, hence not the default;
    exercised on CPU by a world-size-3 gloo test.
  * `gather_waveforms`      ragged gather of the generated fp32 waveforms to `dst`.
The backend is whatever the process group was created with: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
`shard_items` is the dialogue -> rank assignment (dialogue i -> rank i mod world).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .config import VVConfig
from .synth import state_dict_shapes


def shard_items(n_items: int, rank: int, world: int) -> List[int]:
    """Indices of the dialogues rank `rank` serves."""
    return list(range(rank, n_items, world))


def _layout(cfg: VVConfig, dtype: torch.dtype):
    mats, vecs = [], []
    for name, shape in state_dict_shapes(cfg).items():
        (vecs if len(shape) <= 1 else mats).append((name, tuple(shape)))
    return mats, vecs


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= d
    return n


_ALIGN = 128     # elements: every tensor starts on a 256-byte boundary of its blob (the kernels want 16-byte aligned rows)


def _padded(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _broadcast_flat(flat: torch.Tensor, src: int):
    """In-place broadcast of a 1-D tensor whose length is a multiple of the world size.  VV_BCAST=scatter_allgather selects the
    scatter + all-gather form on any backend and world size >= 2 (the gloo tests drive it on CPU); default: one broadcast."""
    import os
    world = dist.get_world_size()
    if world < 2 or os.environ.get("VV_BCAST", "broadcast") != "scatter_allgather":
        dist.broadcast(flat, src=src)
        return
    chunk = flat.numel() // world
    mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
    parts = [p.contiguous() for p in flat.split(chunk)] if dist.get_rank() == src else None
    dist.scatter(mine, parts, src=src)
    if dist.get_backend() == "gloo":        # no all_gather_into_tensor on gloo: gather into the chunk views of the flat buffer
        dist.all_gather(list(flat.split(chunk)), mine)
    else:
        dist.all_gather_into_tensor(flat, mine)


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], cfg: VVConfig, dtype: torch.dtype, device, src: int = 0):
    """Returns the full state dict on every rank (views into two packed blobs on the receivers)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mats, vecs = _layout(cfg, dtype)
    out: Dict[str, torch.Tensor] = {}
    for group, gdtype in ((mats, dtype), (vecs, torch.float32)):
        total = sum(_padded(_numel(s)) for _, s in group)
        padded = (total + world * _ALIGN - 1) // (world * _ALIGN) * (world * _ALIGN)
        flat = torch.empty(padded, dtype=gdtype, device=device)
        if rank == src:
            off = 0
            for name, shape in group:
                n = _numel(shape)
                flat[off: off + n].copy_(sd[name].reshape(-1).to(device=device, dtype=gdtype))
                off += _padded(n)
        _broadcast_flat(flat, src)
        off = 0
        for name, shape in group:
            n = _numel(shape)
            out[name] = flat[off: off + n].view(shape)
            off += _padded(n)
    return out


def gather_waveforms(wav: Optional[torch.Tensor], dst: int = 0) -> Optional[List[Optional[torch.Tensor]]]:
    """Ragged gather: every rank contributes one waveform [1, T_r] (or None); `dst` gets the list, others None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = wav.device if wav is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() != "gloo" else torch.device("cpu"))
    n = torch.tensor([0 if wav is None else wav.numel()], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(t.item()) for t in lens]
    mx = max(max(lens), 1)
    buf = torch.zeros(mx, dtype=torch.float32, device=dev)
    if wav is not None:
        buf[: wav.numel()] = wav.reshape(-1).float()
    if rank == dst:
        recv = [torch.empty(mx, dtype=torch.float32, device=dev) for _ in range(world)]
        dist.gather(buf, recv, dst=dst)
        return [(r[:l][None] if l > 0 else None) for r, l in zip(recv, lens)]
    dist.gather(buf, None, dst=dst)
    return None
