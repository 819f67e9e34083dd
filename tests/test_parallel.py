"""CPU, gloo, world_size 2: tile sharding of a frame and the one-collective gather (the N > 1 path of bench.py).
The render itself is replaced by a deterministic function of the ray (no GPU here); the sharding / gather logic is
what is under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quadraturefields_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(o, d):
    return torch.cat([o * 2 + d, (o * d).sum(-1, keepdim=True), o[:, :1] - d[:, 2:]], dim=1)   # [n,5]


def _worker(rank, world, port, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    r, lr, ws = parallel.init_from_env("gloo")
    assert (r, ws) == (rank, world)
    g = torch.Generator().manual_seed(0)
    o, d = torch.rand(w * h, 3, generator=g), torch.rand(w * h, 3, generator=g)
    lo, ld, ids = parallel.local_rays(o, d, w, h, rank, world)
    frame = parallel.gather_frame(_fake_render(lo, ld), w, h, rank, world)
    ok = torch.equal(frame, _fake_render(o, d))
    owned = int((ids >= 0).sum())
    q.put((rank, ok, owned))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(32, 16), (37, 21)])
def test_two_rank_tile_gather(w, h):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert sum(owned for _, _, owned in results) == w * h          # every pixel rendered exactly once


def test_tile_layout_single_process():
    w, h = 20, 12
    for world in (1, 2, 3, 8):
        seen = torch.zeros(w * h, dtype=torch.int32)
        per = None
        for r in range(world):
            ids = parallel.tile_ray_indices(parallel.shard_tiles(w, h, r, world), w, h)
            assert ids.shape[1] == 64
            per = ids.shape[0] if per is None else per
            assert ids.shape[0] == per                             # equal buffer sizes for the all_gather
            v = ids[ids >= 0]
            seen[v] += 1
        assert bool((seen == 1).all())
    g = torch.Generator().manual_seed(1)
    o, d = torch.rand(w * h, 3, generator=g), torch.rand(w * h, 3, generator=g)
    lo, ld, ids = parallel.local_rays(o, d, w, h, 0, 1)
    assert torch.equal(parallel.gather_frame(_fake_render(lo, ld), w, h, 0, 1), _fake_render(o, d))
