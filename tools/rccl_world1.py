#!/usr/bin/env python3
"""Drive the REAL RCCL branches of mofanerf_amd/dist.py on a box with ONE GPU.

RCCL refuses two ranks on the same device ("Duplicate GPU detected"), so a single-GPU box cannot run a 2-rank `nccl` job; what
it can run is a ONE-rank `nccl` process group with `MOFA_DIST_FORCE_COLLECTIVES=1`, under which every collective of the
multi-GPU path is issued exactly as an N-rank job issues it (same tensors, same devices, same call sequence) instead of being
short-circuited: `init_process_group(backend="nccl")` (librccl loaded, communicator created), `all_gather_into_tensor` of the
[rays/N, 5] tiles straight into the reused frame buffer, ONE `all_reduce` of the flat 32.6 M-float gradient bucket,
`barrier(device_ids=[...])`, the max-over-ranks reduction on a device tensor, plus the two collectives of tools/train_dp.py
(`all_gather` of a device int64) and tools/bulk_render.py (`all_reduce` of a device float64).  A host tensor handed to any of
them, a communicator bound to the wrong device or a missing `device_ids` fails HERE rather than in the 8-GPU run.
What a 1-rank group cannot show is the xGMI data path itself (RCCL turns a 1-rank collective into a device copy).

    python tools/rccl_world1.py            # prints one JSON line
"""
import json
import os
import socket
import sys
import time

os.environ["MOFA_DIST_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("WORLD_SIZE", "1"), os.environ.setdefault("RANK", "0"), os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    _s = socket.socket(); _s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(_s.getsockname()[1]); _s.close()

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import dist as mdist, factory, synth


def loaded(name):
    with open("/proc/self/maps") as f:
        return sorted({l.split()[-1] for l in f if name in l})


def main():
    full = "--small" not in sys.argv
    rank, world, local = mdist.init_from_env()
    assert dist.is_initialized() and world == 1 and mdist.active()
    dev = torch.device("cuda", local)
    out = {"backend": dist.get_backend(), "world": world, "librccl": loaded("librccl"), "device": torch.cuda.get_device_name(dev)}
    assert out["backend"] == "nccl", out
    t0 = time.perf_counter()
    mdist.barrier()                                                    # first use of the communicator: ncclCommInitRank happens here
    out["first_barrier_s"] = round(time.perf_counter() - t0, 3)

    # 1. render: this rank's [rays/N, 5] tile -> the frame, written by the collective into the reused buffer
    n_total, W = 512 * 512, 512
    tile = torch.rand(n_total, 5, device=dev)
    frame = mdist.all_gather_tiles(tile, n_total, world, rank, align=W)
    assert frame.data_ptr() != tile.data_ptr() and torch.equal(frame, tile)
    tile2 = torch.rand(n_total, 5, device=dev)
    frame2 = mdist.all_gather_tiles(tile2, n_total, world, rank, align=W, out=frame)
    assert frame2.data_ptr() == frame.data_ptr() and torch.equal(frame2, tile2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        mdist.all_gather_tiles(tile2, n_total, world, rank, align=W, out=frame)
    e1.record(); torch.cuda.synchronize()
    out["all_gather_tiles_ms"] = round(e0.elapsed_time(e1) / 10, 4)

    # 2. train: ONE all-reduce of the flat gradient bucket of the shipped networks (coarse 256x8 + fine 1024x10 + encoders)
    Dc, Wc, Df, Wf = (8, 256, 10, 1024) if full else (8, 64, 10, 64)
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, no_reload=True, device=dev,
                                basedir="/nonexistent")
    _, _, _, grad_vars, _, _, render = factory.create_nerf(args)
    params = [p for p in grad_vars if p.requires_grad]
    bucket = mdist.GradBucket(params)
    bucket.flat.copy_(torch.randn(bucket.numel, device=dev))
    before = bucket.flat.clone()
    assert params[0].grad.data_ptr() == bucket.flat.data_ptr()         # gradients are views of the bucket
    bucket.sync()
    assert torch.equal(bucket.flat, before)                            # sum over one rank / 1
    e0.record()
    for _ in range(5):
        bucket.sync()
    e1.record(); torch.cuda.synchronize()
    out["bucket_floats"] = bucket.numel
    out["bucket_all_reduce_ms"] = round(e0.elapsed_time(e1) / 5, 4)

    # 3. timing contract: barrier naming this rank's device, max over ranks on a device tensor
    mdist.barrier()
    assert mdist.barrier_max(1.25, dev) == 1.25
    # 4. the side collectives of tools/train_dp.py and tools/bulk_render.py (device tensors, as RCCL needs them)
    t = torch.tensor([123456789], dtype=torch.int64, device=dev)
    got = [torch.zeros_like(t)]
    dist.all_gather(got, t)
    assert int(got[0].item()) == 123456789
    tot = torch.tensor([3.0], dtype=torch.float64, device=dev)
    dist.all_reduce(tot)
    assert float(tot.item()) == 3.0
    # 5. a host tensor must be REFUSED by this backend (the class of bug this script exists to catch)
    try:
        dist.all_reduce(torch.ones(1))
        out["host_tensor_refused"] = False
    except Exception as e:       # noqa: BLE001
        out["host_tensor_refused"] = True
        out["host_tensor_error"] = type(e).__name__
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:            # noqa: BLE001
        pass
    mdist.barrier()
    dist.destroy_process_group()
    out["ok"] = True
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
