"""Multi-GPU partitioning of the hot path: one process per GPU, RCCL over xGMI through ``torch.distributed``.

Rays are independent (no cross-ray op between render_class.py:284 and :352), so a frame is split into
contiguous row blocks, each rank runs the whole coarse->fine pipeline on its block and ONE collective — an
all-gather of the ``[rays/N, 5]`` fp32 tiles (rgb, disp, acc) — reassembles the frame on every rank.
Identity lists (``render_refine_trainSet.py:158-159``'s begin_person/end_person) shard the same way with no
data-path collective at all.  Backend: ``nccl`` (= RCCL) on GPUs, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's env; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous block ``[begin, end)`` of ``n`` units for ``rank``; block sizes differ by at most ``align`` and
    every boundary is a multiple of ``align`` (e.g. an image row)."""
    units = (n + align - 1) // align
    base, rem = divmod(units, world)
    b = rank * base + min(rank, rem)
    e = b + base + (1 if rank < rem else 0)
    return min(b * align, n), min(e * align, n)


def shard_list(items: List, rank: int, world: int) -> List:
    b, e = shard_range(len(items), rank, world)
    return items[b:e]


def all_gather_tiles(local: torch.Tensor, n_total: int, world: int, rank: int, align: int = 1, group=None) -> torch.Tensor:
    """All-gather row blocks produced under :func:`shard_range` into the full ``[n_total, C]`` tensor on every rank.
    Uneven blocks are padded to the largest block so that a single ``all_gather_into_tensor`` (one RCCL call) does it."""
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world, align) for r in range(world)]
    mx = max(e - b for b, e in sizes)
    pad = torch.zeros(mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty(world * mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + (e - b)] for r, (b, e) in enumerate(sizes)], 0)


def barrier_max(seconds: float, device) -> float:
    """Max over ranks of a local duration (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
