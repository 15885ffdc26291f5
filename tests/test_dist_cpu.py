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
