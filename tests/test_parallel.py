"""CPU, gloo, world_size 2: the host logic of quadraturefields_amd/parallel.py -- row-band sharding of ONE frame
(cost-balanced cuts, band cameras, the padded one-collective gather, the fixed-lag profile that keeps the ranks'
cuts identical) and the round-1 tile sharding for arbitrary ray sets.  The render itself is replaced by a
deterministic function of the ray (no GPU here); the same band path with the HIP render on a real device is
tests/test_gpu_parallel.py, and bench.py --gpus N times it (its "sharded_frame" object)."""
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


# ----------------------------------------------------------------------------------------------------------
# row bands

class _FakeRenderer:
    """Stands in for render.FrameRenderer: a deterministic function of each ray, alpha > 0 on a disc of the frame."""

    def __init__(self):
        self.cameras = []

    def render(self, o, d, camera=None):
        self.cameras.append((camera.width, camera.height, camera.cy))
        assert o.shape[0] == camera.width * camera.height
        v = _fake_render(o, d)
        alpha = ((o[:, :1] - 0.5) ** 2 + (o[:, 1:2] - 0.5) ** 2 < 0.09).float()
        return v[:, :3], alpha, v[:, 4:5], 0


def _frame_inputs(w, h):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    o = torch.stack([(xs.flatten() + 0.5) / w, (ys.flatten() + 0.5) / h, torch.zeros(w * h)], dim=1)
    d = torch.rand(w * h, 3, generator=torch.Generator().manual_seed(3))
    return o, d


def _band_worker(rank, world, port, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    parallel.init_from_env("gloo")
    o, d = _frame_inputs(w, h)
    c2w = torch.eye(4)[:3]
    ref = _FakeRenderer().render(o, d, camera=parallel.band_camera(c2w, 50.0, w, h, 0, h))
    ref = torch.cat([ref[0], ref[1], ref[2]], dim=1)
    sr = parallel.ShardedFrameRenderer(_FakeRenderer(), rank, world)
    ok, cuts = True, []
    for _ in range(4):                       # frames 0-1 use uniform cuts, 2-3 the profile of frames 0-1
        frame = sr.render(o, d, c2w, 50.0, w, h)
        ok = ok and torch.equal(frame, ref)
        cuts.append(list(sr.last_cuts))
    # per-row numbers ride in the same collective (the bands' sample counts): uneven bands, more rows than one padded
    # row of the payload holds
    if world == 2:
        uneven = [0, 8 * (h // 24 + 1), h]
    elif world == 3:
        uneven = [0, 8, 8 * (h // 16), h]
    else:                                    # ragged, one empty band in the middle, the rest to the last rank
        uneven = [0, 8, 8, 32, 40, 56, 64, 72, h][:world] + [h]
    rows = uneven[rank + 1] - uneven[rank]
    local = torch.full((rows * w, 5), float(rank))
    for mode in ("exact", "padded"):         # all_to_all_single with split sizes; all_gather of padded bands
        fr, m = parallel.gather_bands(local, uneven, w, rank, world, meta=torch.arange(uneven[rank], uneven[rank + 1]).float(),
                                      mode=mode)
        ok = ok and torch.equal(m, torch.arange(h).float()) and fr.shape == (h * w, 5)
        ok = ok and all(bool((fr[uneven[r] * w:uneven[r + 1] * w] == r).all()) for r in range(world))
        fr2 = parallel.gather_bands(local, uneven, w, rank, world, mode=mode, async_op=True)()      # without meta, async
        ok = ok and torch.equal(fr2, fr)
    q.put((rank, ok, cuts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h,world", [(24, 64, 2), (17, 43, 2), (2, 80, 2), (24, 64, 3), (16, 200, 8)])
def test_two_rank_band_render_and_gather(w, h, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert all(r[2] == results[0][2] for r in results)       # every rank cuts every frame identically
    cuts = results[0][2]
    assert cuts[0] == cuts[1] == parallel.band_cuts(h, world)    # uniform until the lagged profile arrives
    assert cuts[2][0] == 0 and cuts[2][-1] == h and cuts[2][1] % parallel.BAND_ALIGN == 0
    assert len(cuts[2]) == world + 1


def test_band_cuts_properties():
    import numpy as np
    for h, world in [(800, 1), (800, 8), (1080, 8), (100, 3), (37, 2), (7, 2), (20, 8), (8, 8)]:
        cuts = parallel.band_cuts(h, world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == h
        assert all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
        assert all(c % parallel.BAND_ALIGN == 0 for c in cuts if c < h)
        n_blocks = -(-h // parallel.BAND_ALIGN)
        if n_blocks >= world:
            assert all(b > a for a, b in zip(cuts[:-1], cuts[1:]))       # nobody idles while there are blocks
    # a profile that is heavy in the middle: bands there are shorter, costs come out nearly equal
    rc = np.full(800, 0.15 * 800)
    rc[230:570] += 3.9 * 600
    cuts = parallel.band_cuts(800, 8, rc)
    cost = [rc[a:b].sum() for a, b in zip(cuts[:-1], cuts[1:])]
    assert max(cost) <= 1.15 * (sum(cost) / 8)
    assert cuts[1] - cuts[0] > cuts[4] - cuts[3]
    with pytest.raises(ValueError):
        parallel.band_cuts(800, 2, rc[:10])


def test_band_camera_is_the_frame_camera_shifted():
    c2w = torch.eye(4)[:3]
    full = parallel.band_camera(c2w, 1111.0, 800, 800, 0, 800)
    band = parallel.band_camera(c2w, 1111.0, 800, 800, 304, 400)
    assert (full.width, full.height, full.cy) == (800, 800, 400.0)
    assert (band.width, band.height, band.cx) == (800, 96, 400.0)
    assert band.cy == 400.0 - 304 and band.fx == full.fx == band.fy
    assert list(band.c2w) == list(full.c2w)


def test_gather_bands_single_rank_and_shape_check():
    x = torch.arange(40.0).reshape(8, 5)
    assert parallel.gather_bands(x, [0, 2], 4, 0, 1) is x
    with pytest.raises(ValueError):
        parallel.gather_bands(x, [0, 3], 4, 0, 1)


# ----------------------------------------------------------------------------------------------------------
# the collective self-test an N > 1 run starts with: mode selection, fallbacks, hard failure

def _raiser(name):
    def fail(*a, **k):
        raise RuntimeError(f"injected {name} failure")
    return fail


def _selftest_worker(rank, world, port, broken, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    parallel.init_from_env("gloo")
    for name in broken:                       # the collective every rank finds broken (what an RCCL build that
        setattr(dist, name, _raiser(name))    # rejects a call would look like: a synchronous error on every rank)
    try:
        info = parallel.selftest_collectives(rank, world)
        report = parallel.device_report(rank, world)
        # the selected modes are the ones the module now uses by default: a real exchange through them
        cuts = {2: [0, 8, 24], 3: [0, 8, 8, 24]}.get(world) or [8 * r for r in range(world)] + [8 * world + 16]
        w = 4
        local = torch.full(((cuts[rank + 1] - cuts[rank]) * w, 5), float(rank))
        frame = parallel.gather_bands(local, cuts, w, rank, world)
        ok = all(bool((frame[cuts[r] * w:cuts[r + 1] * w] == r).all()) for r in range(world))
        frames = parallel.gather_frames(torch.full((6, 5), float(rank)), rank, world)
        if parallel.FRAME_GATHER_MODE == "rank0" and rank != 0:
            ok = ok and frames is None
        else:
            ok = ok and all(bool((frames[r] == r).all()) for r in range(world))
        q.put((rank, "ok" if ok else "wrong", info, report))
    except parallel.CollectiveSelfTestError as e:
        q.put((rank, "error", str(e), None))
    dist.barrier()
    dist.destroy_process_group()


def _run_selftest(world, broken):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_selftest_worker, args=(r, world, port, broken, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world,broken,band,frame", [
    (2, (), "exact", "rank0"),
    (3, (), "exact", "rank0"),
    (2, ("all_to_all_single",), "padded", "rank0"),
    (3, ("all_to_all_single",), "padded", "rank0"),
    (2, ("all_to_all_single", "all_gather_into_tensor"), "broadcast", "rank0"),
    (3, ("gather",), "exact", "all"),
    (8, (), "exact", "rank0"),                       # the node the bench is meant for: eight ranks
    (8, ("all_to_all_single",), "padded", "rank0"),
])
def test_collective_selftest_picks_a_working_mode(world, broken, band, frame):
    results = _run_selftest(world, broken)
    for rank, status, info, report in results:
        assert status == "ok", (rank, status, info)
        assert info["backend"] == "gloo" and info["world_size"] == world
        assert info["band_gather_mode"] == band and info["frame_gather_mode"] == frame
        assert info["tested"]["all_reduce_f64"] == "ok"
        for name in broken:                   # the injected error text is in the record
            hit = [v for v in info["tested"].values() if f"injected {name} failure" in v]
            assert hit, info["tested"]
        assert len(report["ranks"]) == world and [r["rank"] for r in report["ranks"]] == list(range(world))
    assert all(r[2] == results[0][2] for r in results)                     # every rank chose the same modes


def test_collective_selftest_fails_loudly_when_nothing_works():
    results = _run_selftest(2, ("all_to_all_single", "all_gather_into_tensor", "broadcast"))
    for rank, status, text, _ in results:
        assert status == "error"
        assert "no working collective for the band exchange" in text and "injected broadcast failure" in text


def test_selftest_without_a_process_group_is_a_no_op():
    info = parallel.selftest_collectives(0, 1)
    assert info["backend"] is None and info["band_gather_mode"] == parallel.GATHER_MODE
    x = torch.arange(30.0).reshape(6, 5)
    assert parallel.gather_frames(x, 0, 1).shape == (1, 6, 5)
    assert parallel.gather_frames(x, 0, 1, mode="none") is None
    with pytest.raises(ValueError):
        parallel.gather_frames(x, 0, 1, mode="ring")
