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


def _rccl_env() -> None:
    """The host driver of these nodes only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's cross-process
    buffer registration fails with "hipIpcGetMemHandle: invalid argument".  The variable must be in the environment before the
    HIP runtime initialises (lazy in torch), so it is defaulted HERE — when a multi-process group is about to be created — not
    as a side effect of importing the package; an explicit setting of the caller wins."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def force_collectives() -> bool:
    """``MOFA_DIST_FORCE_COLLECTIVES=1``: create the process group and issue every collective even when the job has ONE rank.
    A 1-rank job normally short-circuits (nothing to exchange); with the flag the same branches run that an N-rank job takes —
    ``init_process_group(backend="nccl")`` (RCCL communicator creation), ``all_gather_into_tensor`` into the frame buffer,
    the flat-bucket ``all_reduce``, ``barrier(device_ids=...)``, the max-over-ranks reduction on a device tensor — which is how
    the RCCL path is exercised on a single-GPU box (tests/test_gpu_rccl.py, tools/rccl_world1.py).  Results are unchanged: a
    1-rank all-gather / all-reduce / average is the identity."""
    return os.environ.get("MOFA_DIST_FORCE_COLLECTIVES", "") not in ("", "0")


def active(group=None) -> bool:
    """True when collectives must be issued: a group exists and it has more than one rank (or the force flag is set)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or force_collectives()


def ranks_share_a_device(local_world: int, n_devices: int) -> bool:
    """True when the ranks of this node outnumber its GPUs, i.e. at least two processes drive one device (the single-GPU functional
    form of an N-rank job; never the deployment: one process per GPU)."""
    return n_devices > 0 and local_world > n_devices


shared_device = None     # (ranks on this node, visible GPUs) when init_from_env found ranks sharing a device, else None


def default_backend(cuda: bool, sharing: bool) -> str:
    """``nccl`` (= RCCL over xGMI) with one rank per GPU; ``gloo`` on the CPU and when ranks SHARE a device — RCCL refuses two ranks on one
    GPU, so the single-GPU functional form of an N-rank job (tensors staged through the host, see ``all_gather_tiles``) needs no flag."""
    return "nccl" if cuda and not sharing else "gloo"


def per_layer_launches_when_sharing(local_world: int, n_devices: int) -> bool:
    """The chained launch (csrc/mofa_mlp.hip ``k_net_chain``) is a persistent kernel whose workgroups WAIT for one another.  That is
    safe while one process owns the device: a ticket drawn earlier is always held by a resident or finished workgroup.  When several
    processes share a device the hardware scheduler time-slices their queues (compute-wave save / restore) and restores a queue's
    waves piecemeal into whatever slots other processes' persistent workgroups leave free — waiting workgroups can then hold the chip
    while the workgroups they wait for sit saved in memory.  Observed with eight ranks at the full benchmark size on one MI355X
    (`profiles/r06_shared_device_chain.md`): a dependency wait ran out of its budget (seconds; eight times the budget changed
    nothing), the launch ended incomplete and the library — as designed — poisoned its outputs and raised ``MofaError``.  Loud, never
    wrong, but not a frame.  So ranks that share a device take the per-layer launches (bit-identical results, about 1 % slower,
    no inter-workgroup waits), and say so once.  An explicit ``MOFA_CHAIN`` of the caller wins.  Returns True when it switched.
    (The test is the launcher's view — ranks on this node against VISIBLE devices.  A launcher that narrows each rank's visibility to its
    own GPU makes every rank see one device and N ranks: the per-layer launches are then taken needlessly — correct, about 1 % slower;
    export MOFA_CHAIN=1 there.  torch.distributed.run, the driver's launcher, leaves all GPUs visible to every rank.)"""
    if not ranks_share_a_device(local_world, n_devices) or os.environ.get("MOFA_CHAIN") is not None:
        return False
    os.environ["MOFA_CHAIN"] = "0"
    from . import lib
    if lib._lib is not None:            # the knobs are read once at load: a library that is already loaded re-reads them
        lib.reload_env()
    import warnings
    warnings.warn(f"{local_world} ranks share {n_devices} device(s): taking the per-layer launches (MOFA_CHAIN=0; bit-identical, about 1 % "
                  "slower) — the chained launch's workgroups wait for one another and can starve when the hardware scheduler time-slices "
                  "several processes on one device", lib.MofaWarning, stacklevel=3)
    return True


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's env; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force_collectives():
        _rccl_env()                      # (before anything below can initialise the HIP runtime)
    global shared_device
    sharing = False
    if world > 1 and torch.cuda.is_available():
        local_world, n_dev = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), torch.cuda.device_count()
        sharing = ranks_share_a_device(local_world, n_dev)
        shared_device = (local_world, n_dev) if sharing else None
        per_layer_launches_when_sharing(local_world, n_dev)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:      # MOFA_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the N>1 path)
            backend = os.environ.get("MOFA_DIST_BACKEND") or default_backend(torch.cuda.is_available(), sharing)
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
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


def all_gather_tiles(local: torch.Tensor, n_total: int, world: int, rank: int, align: int = 1, group=None,
                     out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather row blocks produced under :func:`shard_range` into the full ``[n_total, C]`` tensor on every rank with ONE
    collective.  When the blocks are equal (every benchmark shape: 512 rows over 1/2/4/8 ranks) the collective writes straight
    into the result — no padding, no copies; pass ``out`` to reuse the frame buffer across calls.  Uneven blocks are padded to
    the largest block and compacted afterwards."""
    if world == 1 and not active(group):
        return local
    sizes = [shard_range(n_total, r, world, align) for r in range(world)]
    mx = max(e - b for b, e in sizes)
    even = all(e - b == mx for b, e in sizes)
    stage_host = local.is_cuda and dist.get_backend(group) == "gloo"      # test configuration only: stage through the host
    if even:
        if out is None or out.shape[0] != n_total:
            out = torch.empty(n_total, *local.shape[1:], dtype=local.dtype, device=local.device)
        if stage_host:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros(mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty(world * mx, *local.shape[1:], dtype=local.dtype, device=local.device)
    if stage_host:
        host = torch.empty(buf.shape, dtype=buf.dtype)
        dist.all_gather_into_tensor(host, pad.cpu(), group=group)
        buf.copy_(host)
    else:
        dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * mx: r * mx + (e - b)] for r, (b, e) in enumerate(sizes)], 0)


def barrier(group=None) -> None:
    """Barrier that names this rank's own GPU under RCCL (without ``device_ids`` torch has to guess the device of a
    communicator that has not been used yet)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend(group) == "nccl":
        dist.barrier(group=group, device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier(group=group)


def barrier_max(seconds: float, device) -> float:
    """Max over ranks of a local duration (bench timing contract)."""
    if not active():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradBucket:
    """One flat fp32 gradient buffer for data-parallel training (BASELINE config 5: N_rand rays per GPU, RCCL
    all-reduce of ~32.6 M gradients = 130 MB per step).

    Every parameter's ``.grad`` is made a VIEW into the flat buffer, so the backward pass writes gradients in place and
    the synchronisation is ONE ``all_reduce`` on ONE tensor — no per-parameter launches, no flatten/unflatten copies.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a single large message lets RCCL stripe its rings over all
    links, whereas many small buckets would be latency-bound; the ~1.5 ms it takes is <1 % of a 4096-ray step, so no
    overlap with backward is attempted.  Parameters that never receive a gradient (``texEncoder.encoder.logstd``,
    tex_encoder_mod.py:56) simply keep their zeros."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        if not self.params:
            raise ValueError("GradBucket needs at least one parameter that requires grad")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket expects fp32 parameters on one device")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p in self.params:            # an optimizer's zero_grad(set_to_none=True) would detach the views
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr():
                raise RuntimeError("a parameter lost its bucket view; call GradBucket.zero() instead of zero_grad()")

    def sync(self):
        """Average the gradients over the ranks (no-op for a single process)."""
        if active(self.group):
            if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":   # functional-test configuration: stage via host
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))
        return self.flat
