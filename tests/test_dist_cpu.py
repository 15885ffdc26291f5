"""world_size-2 gloo tests (CPU) of the multi-GPU partitioning used by bench.py / bulk rendering."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mofanerf_amd import dist as mdist


def test_shard_range_covers_everything():
    for n in (1, 7, 512 * 512, 65536 + 3):
        for world in (1, 2, 3, 8):
            for align in (1, 512):
                blocks = [mdist.shard_range(n, r, world, align) for r in range(world)]
                assert blocks[0][0] == 0 and blocks[-1][1] == n
                assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
                assert all(b % align == 0 or b == n for b, _ in blocks)
    assert mdist.shard_list(list(range(300)), 3, 8) == list(range(114, 152))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, align):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    b, e = mdist.shard_range(n_total, rank, world, align)
    full = torch.arange(n_total * 5, dtype=torch.float32).reshape(n_total, 5)      # stand-in for (rgb, disp, acc) tiles
    out = mdist.all_gather_tiles(full[b:e].clone(), n_total, world, rank, align)
    assert torch.equal(out, full)
    t = mdist.barrier_max(float(rank + 1), torch.device("cpu"))
    assert t == float(world)
    dist.destroy_process_group()


def test_all_gather_tiles_gloo_world2():
    for n_total, align in ((64 * 64, 64), (1000, 1), (3, 1)):
        mp.spawn(_worker, args=(2, _free_port(), n_total, align), nprocs=2, join=True)


def _bucket_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    mdist.init_from_env("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    extra = torch.nn.Parameter(torch.ones(5))                     # never used in the loss: gradient stays zero
    bucket = mdist.GradBucket(list(net.parameters()) + [extra])
    assert bucket.numel == sum(p.numel() for p in net.parameters()) + 5
    x = torch.full((4, 8), float(rank + 1))
    bucket.zero()
    net(x).pow(2).sum().backward()                                # autograd accumulates INTO the bucket views
    local = bucket.flat.clone()
    assert local.abs().sum() > 0 and all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in bucket.params)
    bucket.sync()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(bucket.flat, sum(gathered) / world, atol=1e-6)
    assert torch.equal(extra.grad, torch.zeros(5))
    dist.destroy_process_group()


def test_grad_bucket_allreduce_gloo_world2():
    mp.spawn(_bucket_worker, args=(2, _free_port()), nprocs=2, join=True)


def _forced_worker(rank, world, port):
    """ONE rank with MOFA_DIST_FORCE_COLLECTIVES=1: the group is created and every collective is issued (the switch the GPU box
    uses to run the RCCL branches on one device, tests/test_gpu_rccl.py) — here on gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      MOFA_DIST_FORCE_COLLECTIVES="1")
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (0, 1) and dist.is_initialized() and mdist.active()
    tile = torch.arange(64 * 5, dtype=torch.float32).reshape(64, 5)
    frame = mdist.all_gather_tiles(tile, 64, 1, 0, align=8)
    assert frame.data_ptr() != tile.data_ptr() and torch.equal(frame, tile)          # went through the collective, not the early return
    again = mdist.all_gather_tiles(tile + 1, 64, 1, 0, align=8, out=frame)
    assert again.data_ptr() == frame.data_ptr() and torch.equal(again, tile + 1)
    w0 = torch.nn.Parameter(torch.ones(7))
    bucket = mdist.GradBucket([w0])
    bucket.flat.fill_(3.0)
    assert torch.equal(bucket.sync(), torch.full((7,), 3.0))
    assert mdist.barrier_max(0.5, torch.device("cpu")) == 0.5
    mdist.barrier()
    dist.destroy_process_group()


def test_forced_collectives_on_one_rank():
    assert not mdist.active()                                                        # no group in the test process itself
    mp.spawn(_forced_worker, args=(1, _free_port()), nprocs=1, join=True)
    os.environ.pop("MOFA_DIST_FORCE_COLLECTIVES", None)


def test_all_gather_tiles_gloo_world8_benchmark_frame_shape():
    """The shape the 8-GPU run takes: a 512 x 512 frame split into eight 64-row blocks, ONE all-gather written straight into the
    frame (equal blocks: no padding path), max-over-ranks timing — on gloo / CPU."""
    mp.spawn(_worker, args=(8, _free_port(), 512 * 512, 512), nprocs=8, join=True)


def test_ranks_sharing_a_device_take_the_per_layer_launches(monkeypatch):
    """Round 6: eight processes on ONE MI355X at the full benchmark size ended a chained launch on a dependency-wait time-out (the
    hardware scheduler time-slices the processes; waiting workgroups held the chip while the workgroups they waited for sat saved in
    memory) — loud (NaN + MofaError), never wrong, but not a frame.  Ranks that share a device therefore select MOFA_CHAIN=0 (the
    per-layer launches: same bits) with a MofaWarning; one rank per device — the deployment — and an explicit setting are left alone."""
    import warnings
    from mofanerf_amd import lib
    assert mdist.ranks_share_a_device(8, 1) and mdist.ranks_share_a_device(2, 1) and mdist.ranks_share_a_device(9, 8)
    assert not mdist.ranks_share_a_device(8, 8) and not mdist.ranks_share_a_device(1, 1) and not mdist.ranks_share_a_device(4, 8)
    assert not mdist.ranks_share_a_device(2, 0)                       # no GPU at all (the CPU tests): nothing to decide
    assert mdist.default_backend(True, False) == "nccl" and mdist.default_backend(True, True) == "gloo" and mdist.default_backend(False, False) == "gloo"
    monkeypatch.setattr(lib, "_lib", None)                            # (no library in a CPU run: nothing to re-read)
    monkeypatch.delenv("MOFA_CHAIN", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert mdist.per_layer_launches_when_sharing(8, 8) is False and "MOFA_CHAIN" not in os.environ and not w
        assert mdist.per_layer_launches_when_sharing(8, 1) is True and os.environ["MOFA_CHAIN"] == "0"
        assert len(w) == 1 and issubclass(w[0].category, lib.MofaWarning) and "per-layer" in str(w[0].message)
    monkeypatch.setenv("MOFA_CHAIN", "1")                             # the caller's explicit choice wins
    assert mdist.per_layer_launches_when_sharing(8, 1) is False and os.environ["MOFA_CHAIN"] == "1"
