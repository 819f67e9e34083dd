"""GPU parity: BVH multi-hit traversal, sample packing and re-sorting vs the brute-force oracle.

Bar: bit-exact triangle ids, hit counts, ray ids and sample order.  The hit distance t, the fp64 locations and
the depths are computed with the same individually rounded IEEE operations on both sides (no FMA contraction in
exact.hip / intersect_ref.c), so they are compared for exact equality too.
"""
import numpy as np
import pytest
import torch

from oracle import meshpath as om
from tests import helpers

pytestmark = pytest.mark.gpu


def _scene(subdiv=3, shells=4, seed=42):
    from quadraturefields_amd import synthetic
    return synthetic.shell_mesh(n_shells=shells, subdivisions=subdiv, seed=seed)


def _rays(n, seed=0, radius=4.0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * radius).astype(np.float32)
    target = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("max_hits", [1, 5, 25])
def test_hits_bit_exact_vs_bruteforce(device, max_hits):
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene()
    ri = RayIntersector(mesh, max_hits=max_hits)
    o, d = _rays(3000, seed=max_hits)
    # a few rays from inside the object, axis-aligned rays (zero direction components) and misses
    o[:50] = 0.0
    d[50:53] = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1]], dtype=np.float32)
    o[53:60] = np.array([10, 10, 10], dtype=np.float32)
    tri_o, t_o, cnt_o = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o, d, max_hits)
    tri, t, cnt, _, _ = ri.hits(o, d)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.array_equal(tri.cpu().numpy(), tri_o)
    assert np.array_equal(t.cpu().numpy(), t_o)
    assert cnt_o.max() == max_hits or max_hits == 25   # the K-nearest truncation is exercised for small K
    # tile-ordered traversal of an image-shaped batch gives identical results
    o2, d2 = o[:2048], d[:2048]
    a = ri.hits(o2, d2, image_width=0)
    b = ri.hits(o2, d2, image_width=64)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


def test_find_intersections_shape(device):
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(2, 2)
    ri = RayIntersector(mesh, max_hits=6)
    o, d = _rays(100, seed=3)
    ints = ri.find_intersections(np.concatenate([o, d], 1).flatten())
    assert ints.shape == (600,) and ints.dtype == np.int32
    tri_o, _, _ = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o, d, 6)
    assert np.array_equal(ints.reshape(100, 6), tri_o)


def test_native_module_shape(device):
    """``intersector.Intersector(vertices[faces].flatten(), K, 0)`` / find_intersections / update_vertices, and the
    finetune loop's replacement of ``rayintersector.inter`` (mesh_utils.py:77-96, train_finetune.py:716-718)."""
    from quadraturefields_amd import intersector
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(2, 3)
    soup = mesh.vertices[mesh.faces].flatten()
    o, d = _rays(300, seed=5)
    rays = np.concatenate([o, d], 1).flatten()
    it = intersector.Intersector(soup, 8, 0)
    ints = np.array(it.find_intersections(rays))
    assert ints.shape == (300 * 8,)
    tri_o, _, _ = om.BruteForceIntersector(mesh.vertices, mesh.faces, min_separation=0.0).hits(o, d, 8)
    assert np.array_equal(ints.reshape(300, 8), tri_o)
    # the adapter's use of it (mesh_utils.py:91-96): hit slots -> (triangle, ray) pairs
    idx = np.where(ints > -1)[0]
    assert np.array_equal(idx // 8, np.repeat(np.arange(300), (tri_o >= 0).sum(1)))
    # new positions, same triangles: in place ...
    moved = mesh.vertices * np.array([1.0, 0.9, 1.1]) + 0.01
    soup2 = moved[mesh.faces].flatten()
    it.update_vertices(soup2)
    tri_m, _, _ = om.BruteForceIntersector(moved, mesh.faces, min_separation=0.0).hits(o, d, 8)
    assert np.array_equal(np.array(it.find_intersections(rays)).reshape(300, 8), tri_m)
    # ... or by replacing the adapter's object, as the finetune loop does
    ri = RayIntersector(mesh, max_hits=8, min_separation=0.0)
    assert np.array_equal(np.array(ri.inter.find_intersections(rays)).reshape(300, 8), tri_o)
    ri.inter = intersector.Intersector(soup2, 8, 0)
    assert np.array_equal(np.array(ri.inter.find_intersections(rays)).reshape(300, 8), tri_m)
    ri.inter.update_vertices(soup)
    assert np.array_equal(ri.find_intersections(rays).reshape(300, 8), tri_o)
    with pytest.raises(ValueError):
        it.update_vertices(soup[:-9])
    with pytest.raises(ValueError):
        ri.inter = intersector.Intersector(soup[:-9], 8, 0)
    with pytest.raises(TypeError):
        ri.inter = object()


def test_sampling_raytrace_matches_oracle(device):
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = _scene()
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    o, d = _rays(4000, seed=7)
    want = om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d, o, 25)
    want = om.to_loader_tensors(want)
    got = mi.sampling_raytrace_device(d, o)
    names = ["xyzs", "dirs", "index_ray", "ts", "index_tri", "origins"]
    for name, g, w in zip(names, got, want):
        assert g.shape == w.shape, name
        assert torch.equal(g.cpu(), w), f"{name}: max diff {(g.cpu().double() - w.double()).abs().max()}"
    # numpy duck type
    got_np = mi.sampling_raytrace_numpy(d, o, 0)
    assert got_np[5] == 0 and np.array_equal(got_np[2], want[2].numpy())
    tri, ray, loc = mi.rayintersector.intersects_id(o, d, multiple_hits=True, return_locations=True, max_hits=25)
    assert np.array_equal(tri, want[4].numpy()) and np.array_equal(ray, want[2].numpy())
    # no hit at all -> None (mesh_utils.py:357-358)
    far = np.tile(np.array([[50, 50, 50]], np.float32), (8, 1))
    assert mi.sampling_raytrace_numpy(d[:8], far, 0) is None


def test_scale_and_refit(device):
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = _scene(2, 3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.5, num_intersections=10)
    assert np.allclose(mi.mesh.vertices, mesh.vertices * 1.5)
    o, d = _rays(1000, seed=5, radius=6.0)
    bf = om.BruteForceIntersector(mi.mesh.vertices, mi.mesh.faces)
    tri, t, cnt, _, _ = mi.rayintersector.hits(o, d)
    tri_o, t_o, cnt_o = bf.hits(o, d, 10)
    assert np.array_equal(tri.cpu().numpy(), tri_o) and np.array_equal(cnt.cpu().numpy(), cnt_o)
    # move the vertices, refit, compare with a brute force on the moved mesh
    rng = np.random.default_rng(0)
    moved = mi.mesh.vertices + rng.normal(scale=0.02, size=mi.mesh.vertices.shape)
    mi.rayintersector.update_intersector(moved)
    tri, t, cnt, _, _ = mi.rayintersector.hits(o, d)
    tri_o, t_o, cnt_o = om.BruteForceIntersector(moved, mi.mesh.faces, mi.rayintersector.min_separation).hits(o, d, 10)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.array_equal(tri.cpu().numpy(), tri_o)
    assert np.array_equal(t.cpu().numpy(), t_o)


def test_sampling_indexing_resort(device):
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = _scene(2, 3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    ridx, _ = helpers.packed_segments(300, 25, seed=2)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(5)
    depth = torch.rand(n, generator=g)
    depth[10:40] = 0.5                                   # ties: the sort must be stable
    pts, org, vec = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    tri = torch.randint(0, 1000, (n,), generator=g)
    want = om.sampling_indexing(pts, org, vec, ridx, depth, tri)
    dev = lambda t: t.to(device)
    got = mi.sampling_indexing(dev(pts), dev(org), dev(vec), dev(ridx), dev(depth), dev(tri))
    for k, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a.cpu(), b), k
    # already sorted input is a fixed point
    again = mi.sampling_indexing(got[0], got[7], got[3], got[4], got[5], got[6])
    assert torch.equal(again[0], got[0]) and torch.equal(again[6], got[6])


def test_bvh_structure(device):
    """Every triangle sits in exactly one leaf and every child box encloses its triangles."""
    from quadraturefields_amd import _C
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(3, 2)
    ri = RayIntersector(mesh, max_hits=4)
    n_nodes, n_tri = ri.num_nodes, mesh.faces.shape[0]
    nodes = np.zeros((n_nodes, 16), np.float32)
    ids = np.zeros(n_tri, np.int32)
    _C.check(_C.lib().qf_bvh_copy_nodes(ri._handle, nodes.ctypes.data, n_nodes))
    _C.check(_C.lib().qf_bvh_copy_tri_ids(ri._handle, ids.ctypes.data, n_tri))
    assert sorted(ids.tolist()) == list(range(n_tri))
    child = nodes[:, 12:14].copy().view(np.int32)
    count = nodes[:, 14:16].copy().view(np.int32)
    tris = mesh.vertices.astype(np.float32)[mesh.faces]
    seen = np.zeros(n_tri, bool)
    for n in range(n_nodes):
        for s in range(2):
            if child[n, s] < 0:
                first = ~child[n, s]
                for k in range(count[n, s]):
                    tid = ids[first + k]
                    assert not seen[tid]
                    seen[tid] = True
                    lo, hi = nodes[n, 6 * s:6 * s + 3], nodes[n, 6 * s + 3:6 * s + 6]
                    assert (tris[tid] >= lo).all() and (tris[tid] <= hi).all()
            else:
                assert child[n, s] > n
    assert seen.all()


def test_full_size_mesh_sampled_against_bruteforce(device):
    """BASELINE size: ~1M-triangle shell mesh; a random subset of camera rays is checked bit-exactly against the
    brute force, all rays through size-independent properties (sortedness, counts, depth == |xyz - o|)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = synthetic.shell_mesh(n_shells=12, subdivisions=6)
    assert mesh.faces.shape[0] == 983040
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    c2w = synthetic.orbit_cameras(1)[0]
    o, d = synthetic.camera_rays(c2w, synthetic.lego_focal(800), 800, 800)
    tri, t, cnt, _, _ = mi.rayintersector.hits(o, d, image_width=800)
    cnt_h = cnt.cpu().numpy()
    assert cnt_h.max() <= 25 and 4.0 < cnt_h.mean() < 12.0
    t_h = t.cpu().numpy()
    k = np.arange(25)[None, :]
    valid = k < cnt_h[:, None]
    assert np.isinf(t_h[~valid]).all() and (tri.cpu().numpy()[~valid] == -1).all()
    tt = np.where(valid, t_h, np.inf)
    assert (np.diff(tt, axis=1)[valid[:, 1:]] >= 0).all()          # ascending within each ray
    rng = np.random.default_rng(1)
    pick = np.concatenate([rng.choice(640000, 192, replace=False), np.flatnonzero(cnt_h == cnt_h.max())[:64]])
    tri_o, t_o, cnt_o = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o[pick].numpy(), d[pick].numpy(), 25)
    assert np.array_equal(cnt_h[pick], cnt_o)
    assert np.array_equal(tri.cpu().numpy()[pick], tri_o)
    assert np.array_equal(t_h[pick], t_o)
    data = mi.sampling_raytrace_device(d, o, image_width=800)
    xyz, dirs, index_ray, ts, index_tri, org = data
    assert xyz.shape[0] == int(cnt_h.sum())
    assert bool((index_ray[1:] >= index_ray[:-1]).all())
    same = index_ray[1:] == index_ray[:-1]
    assert bool((ts[1:][same] >= ts[:-1][same]).all())
    assert torch.allclose((xyz - org).norm(dim=-1), ts, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("w,h,max_hits", [(96, 64, 25), (40, 40, 3)])
def test_raster_intersector_identical_to_bvh(device, w, h, max_hits):
    """Camera-coherent intersector == BVH traversal == brute force, bit for bit; overflow falls back to the BVH."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = _scene(3, 4)
    ri = RayIntersector(mesh, max_hits=max_hits)
    ri.RASTER_WIDE_FACTOR = 1                                  # this test is about the plain pass and its BVH repair
    for seed in (0, 7):
        c2w = synthetic.orbit_cameras(1, seed=seed)[0]
        focal = synthetic.lego_focal(800) * w / 800.0
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        cam = make_camera(c2w, focal, w, h)
        a = ri.hits(o, d, image_width=w)
        b = ri.hits(o, d, camera=cam)                      # max_hits=3 overflows -> exercises the fallback
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        tri_o, t_o, cnt_o = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o.numpy(), d.numpy(), max_hits)
        assert np.array_equal(b[0].cpu().numpy(), tri_o) and np.array_equal(b[2].cpu().numpy(), cnt_o)
        assert np.array_equal(b[1].cpu().numpy(), t_o)
        if max_hits == 25:
            raw = ri._hits_raster(a[3], a[4], max_hits, cam)
            assert int(raw[3].item()) == 0
            s1 = ri.sample_device(o, d, image_width=w)
            s2 = ri.sample_device(o, d, camera=cam)
            for x, y in zip(s1, s2):
                assert torch.equal(x, y)
        else:
            assert int(ri._hits_raster(a[3], a[4], max_hits, cam)[3].item()) > 0
            # overflowing pixels are repaired per ray on the device: the packed samples equal the BVH path's
            ri._raster_backoff = 0
            before = ri.repaired_frames
            s1 = ri.sample_device(o, d, image_width=w)
            s2 = ri.sample_device(o, d, camera=cam)
            assert ri.repaired_frames == before + 1
            for x, y in zip(s1, s2):
                assert torch.equal(x, y)
    # camera inside the object and a camera whose image plane cuts triangles: still identical
    c2w = synthetic.orbit_cameras(1, seed=3)[0].clone()
    c2w[:, 3] *= 0.12
    focal = 30.0
    o, d = synthetic.camera_rays(c2w, focal, w, h)
    a = ri.hits(o, d, image_width=w)
    b = ri.hits(o, d, camera=make_camera(c2w, focal, w, h))
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("k,wide,slabs,room", [(5, 24, 0, 32), (5, 7, 0, 32), (5, 0, 0, 32), (64, 128, 0, 32),
                                               (5, 24, 8, 32), (5, 24, 2, 32), (5, 24, 16, 1), (5, 0, 8, 32), (64, 128, 5, 32)])
def test_wide_raster_selects_the_k_nearest_like_the_bvh(device, k, wide, slabs, room):
    """Dense shells (most rays meet more than K triangles).  slabs = 0: qf_raster_intersect_wide keeps the K nearest of up
    to `wide` candidates; rays beyond `wide` (wide = 7) are repaired through the BVH; wide = 0: the policy switches the
    dense mode on by itself after the first heavily overflowing frame; K = 64: the largest LDS footprint of the
    selection kernel (nothing overflows).  slabs > 0: qf_raster_intersect_slabs -- the chunks rasterised in depth slabs,
    nearest first, a pixel stops accepting candidates once it holds K + 8 + 1; room = 1: a single slot beyond that, so
    pixels overflow inside a slab and are repaired through the BVH.  Packed samples identical to the BVH path every
    time."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = _scene(3, 8)
    w, h = 72, 56
    ri = RayIntersector(mesh, max_hits=k)
    ri.raster_wide = wide
    ri.raster_slabs, ri.SLAB_ROOM = slabs, room
    brute = om.BruteForceIntersector(mesh.vertices, mesh.faces)
    for seed in (1, 5, 9):
        c2w = synthetic.orbit_cameras(1, seed=seed)[0]
        focal = synthetic.lego_focal(800) * w / 800.0
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        cam = make_camera(c2w, focal, w, h)
        s1 = ri.sample_device(o, d, image_width=w)
        ri._raster_backoff = 0
        s2 = ri.sample_device(o, d, camera=cam)
        for x, y in zip(s1, s2):
            assert torch.equal(x, y)
        counts = brute.hits(o.numpy(), d.numpy(), 64)[2]
        assert k == 64 or ((counts > k).mean() > 0.05 and counts.max() > 7)     # the scene does what the test is about
        tri_o, _, cnt_o = brute.hits(o.numpy(), d.numpy(), k)
        keep = np.arange(k)[None, :] < cnt_o[:, None]
        want = np.sort((np.arange(w * h)[:, None] * (1 << 20) + tri_o)[keep])
        assert np.array_equal(np.sort(s2[2].cpu().numpy() * (1 << 20) + s2[4].cpu().numpy()), want)
    if wide == 0:
        assert ri.raster_wide == 4 * k and ri._raster_streak == 0       # switched on, never backed off to the BVH
    if wide == 7:
        assert ri._raster_streak > 0 or ri.repaired_frames > 0


def test_coherent_order_is_the_tile_rank_pixel_permutation(device):
    """qf_coherent_order == argsort of (8x8 tile, hit rank, pixel in tile); the field result does not depend on it."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    mesh = _scene(3, 4)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    for w, h in ((64, 48), (37, 21)):                       # second case: ragged tiles at the right/bottom edge
        c2w = synthetic.orbit_cameras(1, seed=w)[0]
        focal = synthetic.lego_focal(800) * w / 800.0
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        data = mi.sampling_raytrace_device(d, o, camera=make_camera(c2w, focal, w, h))
        order = mi.rayintersector.last_order
        ray = data[2].cpu()
        n = ray.shape[0]
        assert order.dtype == torch.int32 and order.shape == (n,)
        first = torch.zeros(w * h + 1, dtype=torch.int64)
        first[1:] = torch.cumsum(torch.bincount(ray, minlength=w * h), 0)
        k = torch.arange(n) - first[ray]
        px, py = ray % w, ray // w
        tiles_x = (w + 7) // 8
        key = ((py // 8) * tiles_x + px // 8) * (64 * 64) + k * 64 + (py % 8) * 8 + px % 8
        assert torch.equal(order.cpu().long(), torch.argsort(key))
        inverse, xyz_c, dirs_c = mi.rayintersector.last_layout     # the same order as inverse map + streamed copies
        assert torch.equal(inverse[order.long()].cpu(), torch.arange(n, dtype=torch.int32))
        assert torch.equal(xyz_c, data[0][order.long()]) and torch.equal(dirs_c, data[1][order.long()])
        field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=12)
        field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
        field = field.to(device)
        a = field(data[0], data[1])
        b = field(data[0], data[1], order=order)
        c = field(data[0], data[1], order=torch.randperm(n, device=device).to(torch.int32))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
        # streaming the coherent copies and compositing through the inverse map == the plain pipeline, bit for bit
        from quadraturefields_amd import utils
        e = field(xyz_c, dirs_c)
        assert torch.equal(e[0][inverse.long()], a[0]) and torch.equal(e[1][inverse.long()], a[1])
        ref = utils.derive_properties(a[0], a[1].reshape(-1), data[3], 0.005, None, data[2], N=w * h)
        got = utils.derive_properties(e[0], e[1].reshape(-1), data[3], 0.005, None, data[2], N=w * h, sample_index=inverse)
        for x, y in zip((ref[0], ref[1], ref[3], ref[4]), (got[0], got[1], got[3], got[4])):
            assert torch.equal(x, y)
    # rays that are not an image: no order
    mi.sampling_raytrace_device(d[:100], o[:100])
    assert mi.rayintersector.last_order is None


@pytest.mark.parametrize("shells,subdiv,w,h,focal_scale", [
    (12, 6, 800, 800, 1.0),        # BASELINE frame: pixel-sized triangles (4 lanes per triangle)
    (12, 6, 1920, 1080, 1.0),      # 1080p, focal 2667 px
    (4, 4, 640, 480, 1.0),         # ~15 pixels per triangle (8 lanes)
    (3, 1, 256, 256, 1.0),         # triangles hundreds of pixels across (16 lanes)
    (4, 3, 200, 200, 6.0),         # telephoto close-up: huge projected triangles, many off screen
])
def test_raster_guard_band_is_conservative(device, shells, subdiv, w, h, focal_scale):
    """The camera-coherent intersector only tests pixels inside the projected triangle grown by a 0.25-pixel guard
    band; whatever it skips must be a miss of the exact test.  Full-size frames from several viewpoints (orbit,
    grazing close-up inside the shells) against the BVH traversal (itself pinned to brute force above): identical
    triangle ids, distances and counts for every ray."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = synthetic.shell_mesh(n_shells=shells, subdivisions=subdiv)
    ri = RayIntersector(mesh, max_hits=25)
    focal = synthetic.lego_focal(w) * focal_scale
    cams = [synthetic.orbit_cameras(3, seed=11)[k] for k in range(3)]
    close = synthetic.orbit_cameras(1, seed=5)[0].clone()
    close[:, 3] *= 0.3                                   # inside the outer shells, grazing views of the inner ones
    cams.append(close)
    compared = 0
    for c2w in cams:
        o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
        tri_r, t_r, cnt_r, overflow = ri._hits_raster(o, d, 25, make_camera(c2w, focal, w, h))
        if int(overflow.item()):
            continue                                     # > 25 candidates somewhere: the product falls back to the BVH
        tri_b, t_b, cnt_b = ri._hits_bvh(o, d, 25, w)
        assert torch.equal(cnt_r, cnt_b)
        assert torch.equal(tri_r, tri_b) and torch.equal(t_r, t_b)
        assert int(cnt_r.sum()) > 0
        compared += 1
    assert compared >= 2


def test_resort_samples_fused_equals_lexsort(device):
    """qf_resort_samples == np.lexsort((depth, ray)) + gathers + mark_pack_boundaries, bit for bit: rays of 1..64
    samples with ties, chunk-boundary straddlers, one ray longer than the staged window (slow path), empty input."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mi = MeshIntersection(synthetic.shell_mesh(n_shells=1, subdivisions=1), simplify_mesh=False, num_intersections=25)
    rng = np.random.default_rng(5)
    counts = rng.integers(1, 65, size=400)
    counts[37] = 1500                                        # longer than chunk + halo
    counts[rng.random(400) < 0.2] = 1
    ray = np.repeat(np.arange(400) * 3, counts)              # grouped, non-contiguous ray ids
    n = ray.shape[0]
    depth = rng.random(n).astype(np.float32)
    depth[rng.random(n) < 0.1] = 0.5                         # ties: stable order
    depth[:5] = np.sort(depth[:5])
    pts, org, vec = (rng.normal(size=(n, 3)).astype(np.float32) for _ in range(3))
    tri = rng.integers(0, 1000, size=n)
    order = np.lexsort((depth, ray))
    t = lambda a: torch.from_numpy(a).to(device)
    out = mi.sampling_indexing(t(pts), t(org), t(vec), t(ray), t(depth), t(tri))
    points, deltas, boundary, vectors, index_ray, d_sorted, index_tri, origins = out
    assert np.array_equal(points.cpu().numpy(), pts[order]) and np.array_equal(vectors.cpu().numpy(), vec[order])
    assert np.array_equal(origins.cpu().numpy(), org[order]) and np.array_equal(d_sorted.cpu().numpy(), depth[order])
    assert np.array_equal(index_tri.cpu().numpy(), tri[order]) and np.array_equal(index_ray.cpu().numpy(), ray[order])
    assert boundary.dtype == torch.bool
    assert np.array_equal(boundary.cpu().numpy(), np.r_[True, ray[1:] != ray[:-1]])
    assert deltas.shape == (n,) and float(deltas[0]) == np.float32(0.005)
    with torch.enable_grad():                                # the differentiable route gives the same arrays
        p2 = t(pts).requires_grad_(True)
        out2 = mi.sampling_indexing(p2, t(org), t(vec), t(ray), t(depth), t(tri))
        assert out2[0].requires_grad and torch.equal(out2[0].detach(), points) and torch.equal(out2[5], d_sorted)
        assert torch.equal(out2[2], boundary) and torch.equal(out2[6], index_tri)
    e = torch.empty
    out0 = mi.sampling_indexing(e((0, 3), device=device), e((0, 3), device=device), e((0, 3), device=device),
                                e((0,), dtype=torch.int64, device=device), e((0,), device=device),
                                e((0,), dtype=torch.int64, device=device))
    assert out0[0].shape == (0, 3) and out0[2].shape == (0,)


def test_bvh_depth_is_bounded_on_lopsided_input(device):
    """The traversal stack holds 64 entries, so the builder bounds the depth for ANY input: below ``sah_depth`` it
    halves the index range instead of taking the SAH split.  Geometrically growing triangles (every SAH split peels a
    few off the far end) built with the default and with a tiny ``sah_depth`` (which forces the halving path): depth
    within the bound, hits those of the brute force, bit for bit."""
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import RayIntersector
    n = 3000
    k = np.arange(n, dtype=np.float64)
    x = 1e-6 * 1.03 ** k                                    # 1e-6 .. ~3e32
    s = x * 0.02
    v = np.stack([np.stack([x, -s, -s], 1), np.stack([x, s, -s], 1), np.stack([x, np.zeros(n), s], 1)], 1).reshape(-1, 3)
    mesh = TriMesh(v.astype(np.float32).astype(np.float64), np.arange(3 * n).reshape(n, 3))
    rng = np.random.default_rng(0)
    m = 300
    o = np.zeros((m, 3), np.float32)
    o[:, 0] = -1.0
    d = np.stack([np.ones(m), rng.normal(size=m) * 5e-3, rng.normal(size=m) * 5e-3], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tri_o, t_o, cnt_o = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o, d, 64)
    assert int(cnt_o.max()) >= 10
    depths = []
    for sah_depth in (32, 3):
        ri = RayIntersector(mesh, max_hits=64, sah_depth=sah_depth)
        depths.append(ri.max_depth)
        assert 1 <= ri.max_depth <= 64
        tri, t, cnt, _, _ = ri.hits(o, d)
        assert np.array_equal(cnt.cpu().numpy(), cnt_o) and np.array_equal(tri.cpu().numpy(), tri_o)
        assert np.array_equal(t.cpu().numpy(), t_o)
    assert depths[0] >= 25                                  # lopsided under SAH ...
    assert depths[1] <= 3 + 12                              # ... and at most sah_depth + ceil(log2(3000)) when halving


# ----------------------------------------------------------------------------------------------------------
# round 2: the 8-wide tree, the reference's multi-hit (re-origin) rule, device-side refit

def test_wide_tree_structure(device):
    """Every triangle sits in exactly one leaf of the 8-wide tree, every child box encloses its subtree's triangles,
    children follow their parent (breadth-first order) and the stack bound covers the worst root-to-leaf path."""
    from quadraturefields_amd import _C
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(3, 3)
    ri = RayIntersector(mesh, max_hits=4)
    n8, n_tri = ri.num_wide_nodes, mesh.faces.shape[0]
    nodes = np.zeros((n8, 8, 8), np.float32)
    ids = np.zeros(n_tri, np.int32)
    _C.check(_C.lib().qf_bvh_copy_wide_nodes(ri._handle, nodes.ctypes.data, n8))
    _C.check(_C.lib().qf_bvh_copy_tri_ids(ri._handle, ids.ctypes.data, n_tri))
    tok = nodes[:, :, 6].copy().view(np.int32)
    tris = mesh.vertices.astype(np.float32)[mesh.faces]
    seen = np.zeros(n_tri, bool)
    empty = np.int32(-2 ** 31)

    def walk(n, depth_sum):
        lo_all, hi_all = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        h = int((tok[n] != empty).sum())
        worst = h
        for j in range(8):
            t = tok[n, j]
            if t == empty:
                continue
            lo, hi = nodes[n, j, 0:3], nodes[n, j, 3:6]
            if t < 0:
                packed = ~t
                first, cnt = packed >> 3, (packed & 7) + 1
                assert 1 <= cnt <= 8
                for k in range(cnt):
                    tid = ids[first + k]
                    assert not seen[tid]
                    seen[tid] = True
                    assert (tris[tid] >= lo).all() and (tris[tid] <= hi).all()
            else:
                assert t > n
                clo, chi, need = walk(int(t), depth_sum)
                assert (clo >= lo).all() and (chi <= hi).all()
                worst = max(worst, h - 1 + need)
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        return lo_all, hi_all, worst

    _, _, need = walk(0, 0)
    assert seen.all()
    assert ri.max_stack >= need + 1 and ri.max_stack <= 384
    assert n8 < ri.num_nodes                                 # the collapse shrinks the node count


def _doubled(mesh, gap):
    """mesh + a copy of it scaled by (1 + gap): gap = 0 duplicates every face exactly."""
    from quadraturefields_amd.mesh_io import TriMesh
    v = np.concatenate([mesh.vertices, mesh.vertices * (1.0 + gap)])
    f = np.concatenate([mesh.faces, mesh.faces + mesh.vertices.shape[0]])
    return TriMesh(v, f)


@pytest.mark.parametrize("gap", [0.0, 5e-5])
@pytest.mark.parametrize("max_hits", [3, 25])
def test_reorigin_rule_on_duplicated_and_near_coincident_shells(device, gap, max_hits):
    """The reference's multi-hit rule (trimesh re-origin, mesh_utils.py:350-354; the reference's mesh is a
    concatenation of two iso-surfaces loaded with process=False): with an exactly duplicated shell and with two shells
    5e-5 apart the second copy of every crossing is never returned.  Ids, counts and distances bit-exact against the
    oracle through every route: BVH traversal (with list pages: K = 3 fills at once), the camera-coherent pass with
    its filter, the wide pass, and the packed samples; rule off = the round-1 behaviour (every hit)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera, trimesh_ray_offset
    mesh = _doubled(_scene(3, 3), gap)
    sep = trimesh_ray_offset(mesh.vertices)
    assert sep == om.trimesh_ray_offset(mesh.vertices) and 1e-3 < sep < 1e-2
    ri = RayIntersector(mesh, max_hits=max_hits)                       # default: the trimesh distance
    assert abs(ri.min_separation - sep) < 1e-12
    o, d = _rays(4000, seed=7)
    o[:40] = 0.0
    bf = om.BruteForceIntersector(mesh.vertices, mesh.faces)
    tri_o, t_o, cnt_o = bf.hits(o, d, max_hits)
    tri, t, cnt, _, _ = ri.hits(o, d)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o) and np.array_equal(tri.cpu().numpy(), tri_o)
    assert np.array_equal(t.cpu().numpy(), t_o)
    # the rule removed the second copies: with it off there are about twice as many hits
    bf_all = om.BruteForceIntersector(mesh.vertices, mesh.faces, min_separation=0.0)
    tri_a, t_a, cnt_a = bf_all.hits(o, d, 64)
    tri_k, t_k, cnt_k = om.BruteForceIntersector(mesh.vertices, mesh.faces).hits(o, d, 64)
    assert cnt_a.sum() >= 1.9 * cnt_k.sum() > 0
    ri.set_min_separation(0.0)
    tri0, t0, cnt0, _, _ = ri.hits(o, d)
    assert np.array_equal(cnt0.cpu().numpy(), np.minimum(cnt_a, max_hits))
    assert np.array_equal(tri0.cpu().numpy(), tri_a[:, :max_hits]) and np.array_equal(t0.cpu().numpy(), t_a[:, :max_hits])
    ri.set_min_separation(sep)
    # camera frames: the coherent pass + filter (and repair of overflowing pixels through the paged traversal)
    w = h = 96
    c2w = synthetic.orbit_cameras(1, seed=3)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    oc, dc = synthetic.camera_rays(c2w, focal, w, h)
    want = om.to_loader_tensors(om.sampling_raytrace_numpy(bf, dc.numpy(), oc.numpy(), max_hits))
    for wide in (0, 4 * max_hits):
        ri._raster_backoff, ri.raster_wide = 0, wide
        got = ri.sample_device(oc.to(device), dc.to(device), max_hits, camera=make_camera(c2w, focal, w, h))
        for a, b in zip(got, want):
            assert torch.equal(a.cpu(), b)
    got = ri.sample_device(oc.to(device), dc.to(device), max_hits, image_width=w)          # BVH route
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    # qf_raster_intersect with sort_lists (RayIntersector.hits with a camera)
    ri._raster_backoff, ri.raster_wide = 0, 0
    tri_c, t_c, cnt_c, _, _ = ri.hits(oc.to(device), dc.to(device), camera=make_camera(c2w, focal, w, h))
    tri_b, t_b, cnt_b = bf.hits(oc.numpy(), dc.numpy(), max_hits)
    assert np.array_equal(cnt_c.cpu().numpy(), cnt_b) and np.array_equal(tri_c.cpu().numpy(), tri_b)


def test_reorigin_rule_pages_through_many_coincident_copies(device):
    """Six exact copies of a shell mesh and K = 4: each crossing fills more than one page of the K-list before the
    chain advances; the paged traversal still returns exactly the oracle's hits."""
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import RayIntersector
    base = _scene(2, 3)
    nv = base.vertices.shape[0]
    mesh = TriMesh(np.concatenate([base.vertices] * 6), np.concatenate([base.faces + i * nv for i in range(6)]))
    o, d = _rays(1500, seed=11)
    for k in (4, 9):
        ri = RayIntersector(mesh, max_hits=k)
        tri_o, t_o, cnt_o = om.BVHIntersector(mesh.vertices, mesh.faces).hits(o, d, k)
        tri, t, cnt, _, _ = ri.hits(o, d)
        assert np.array_equal(cnt.cpu().numpy(), cnt_o) and np.array_equal(tri.cpu().numpy(), tri_o)
        assert np.array_equal(t.cpu().numpy(), t_o)
        assert cnt_o.max() >= 4


def test_oracle_bvh_walk_equals_bruteforce_here_too(device):
    mesh = _scene(3, 4)
    o, d = _rays(2000, seed=5)
    for sep in (0.0, "trimesh", 0.05):
        a = om.BruteForceIntersector(mesh.vertices, mesh.faces, sep).hits(o, d, 25)
        b = om.BVHIntersector(mesh.vertices, mesh.faces, sep).hits(o, d, 25)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_device_refit_equals_a_fresh_intersector(device):
    """update_intersector with a DEVICE tensor (qf_bvh_refit_device: no host round trip) gives the hits of an
    intersector built from the moved vertices, for [V,3] vertices and for the [F*9] triangle soup."""
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(3, 3)
    rng = np.random.default_rng(2)
    moved = mesh.vertices + rng.normal(size=mesh.vertices.shape) * 0.01
    o, d = _rays(3000, seed=9)
    fresh = RayIntersector(TriMesh(moved, mesh.faces), max_hits=25, min_separation=0.0)
    want = fresh.hits(o, d)
    ri = RayIntersector(mesh, max_hits=25, min_separation=0.0)
    ri.update_intersector(torch.from_numpy(moved.astype(np.float32)).to(device))
    for a, b in zip(ri.hits(o, d)[:3], want[:3]):
        assert torch.equal(a, b)
    ri2 = RayIntersector(mesh, max_hits=25, min_separation=0.0)
    soup = torch.from_numpy(moved.astype(np.float32)[mesh.faces].reshape(-1)).to(device)
    ri2.update_intersector(soup)
    for a, b in zip(ri2.hits(o, d)[:3], want[:3]):
        assert torch.equal(a, b)
    # and back again through the host refit
    ri.update_intersector(mesh.vertices)
    orig = RayIntersector(mesh, max_hits=25, min_separation=0.0).hits(o, d)
    for a, b in zip(ri.hits(o, d)[:3], orig[:3]):
        assert torch.equal(a, b)
    # camera-coherent pass reads the refitted triangles too
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    ri.update_intersector(torch.from_numpy(moved.astype(np.float32)).to(device))
    c2w = synthetic.orbit_cameras(1, seed=1)[0]
    oc, dc = synthetic.camera_rays(c2w, synthetic.lego_focal(800) * 64 / 800.0, 64, 64)
    a = ri.sample_device(oc.to(device), dc.to(device), 25, camera=make_camera(c2w, synthetic.lego_focal(800) * 64 / 800.0, 64, 64))
    b = fresh.sample_device(oc.to(device), dc.to(device), 25, image_width=64)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dup,masks,w,h", [(False, False, 96, 64), (True, True, 52, 45), (True, False, 52, 45)])
def test_tile_pack_equals_the_ray_major_pack(device, dup, masks, w, h):
    """qf_pack_tiles (render-only frames: coherent copies only) against qf_pack_samples: positions, unit directions,
    depths.  Plain scene: the same arrays.  A mesh whose faces exist twice, re-origin rule decided up front (keep masks):
    the same arrays.  The same mesh with the rule left to the tile kernel (it applies it on its sorted lists): the same
    per-pixel counts and samples, the dropped hits leave filled gaps at the tile ends."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = _scene(3, 4)
    if dup:
        nv = mesh.vertices.shape[0]
        mesh = TriMesh(np.concatenate([mesh.vertices, mesh.vertices]), np.concatenate([mesh.faces, mesh.faces + nv]))
    ri = RayIntersector(mesh, max_hits=25)
    focal = synthetic.lego_focal(800) * w / 800.0
    c2w = synthetic.orbit_cameras(2, seed=3)[1]
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    cam = make_camera(c2w, focal, w, h)
    # the exact ray-major pack (its optimistic check is refuted on the duplicated mesh and it packs again with masks)
    hit_tri, hit_t, hit_count, overflow = ri._hits_raster_frame(o, d, 25, cam)
    full, order = ri.pack_hits(o, d, 25, hit_tri, hit_t, hit_count, overflow, w, lean=False)
    inverse, xyz_c, dirs_c = ri.last_layout
    depth_c, total = ri.last_frame.depth_c, ri.last_frame.total
    assert total == full[0].shape[0] and torch.equal(xyz_c[inverse.long()], full[0]) and torch.equal(depth_c[inverse.long()], full[3])
    per_ray = torch.bincount(full[2], minlength=w * h)
    # the tile pack on fresh lists
    ri._rule_upfront = 8 if masks else 0
    hit_tri, hit_t, hit_count, overflow = ri._hits_raster_frame(o, d, 25, cam)
    assert (getattr(hit_count, "_qf_keep", None) is not None) == masks
    lean, order2 = ri.pack_hits(o, d, 25, hit_tri, hit_t, hit_count, overflow, w, lean=True)
    assert order2 is None and all(t is None for t in lean)
    none, xyz_t, dirs_t = ri.last_layout
    frame = ri.last_frame
    assert none is None and ri.frame_samples() == total
    assert torch.equal(frame.hit_count.long(), per_ray)
    if masks:
        assert frame.total == total
    if dup and not masks:
        assert frame.total > 1.5 * total                                         # slots were allotted before the rule
    if frame.total == total:                                                     # nothing dropped in the tile kernel
        assert torch.equal(xyz_t, xyz_c) and torch.equal(dirs_t, dirs_c) and torch.equal(frame.depth_c, depth_c)
    else:       # (the plain scene has a near-coincident crossing or two as well) gaps at the tile ends, filled
        assert xyz_t.shape[0] == frame.total and bool(torch.isfinite(xyz_t).all()) and bool(torch.isfinite(dirs_t).all())
        # every sample the exact pack produced is in the tile pack's arrays
        got = set(map(tuple, torch.cat([xyz_t, frame.depth_c[:, None]], dim=1).cpu().numpy().view(np.int32).tolist()))
        want = set(map(tuple, torch.cat([full[0], full[3][:, None]], dim=1).cpu().numpy().view(np.int32).tolist()))
        assert want <= got


def test_frame_offsets_equal_a_cumsum(device):
    """qf_frame_offsets (ray offsets + tile bases in three launches, total and overflow written to pinned memory)
    against torch.cumsum, for image-shaped and plain batches, sizes that are not multiples of the block, empty input."""
    import ctypes
    from quadraturefields_amd import _C
    g = torch.Generator().manual_seed(4)
    for n, w in [(0, 0), (1, 0), (1023, 0), (1024, 0), (70001, 0), (48 * 40, 48), (50 * 37, 50), (800 * 800, 800)]:
        k = 25
        cnt = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(device)           # some above K: clamped
        buf = torch.zeros((n + 2,), dtype=torch.int64, device=device)
        temp = torch.empty((int(_C.lib().qf_frame_offsets_temp_bytes(n)),), dtype=torch.uint8, device=device)
        host = torch.full((4,), -1, dtype=torch.int64).pin_memory()
        ovf = torch.tensor([7], dtype=torch.int32, device=device)
        flag = torch.tensor([1], dtype=torch.int32, device=device)       # the camera-coherent pass's ray flag -> host[3]
        tiles = None
        if w:
            h = n // w
            tiles = torch.empty((((w + 7) // 8) * ((h + 7) // 8),), dtype=torch.int64, device=device)
        _C.check(_C.lib().qf_frame_offsets(_C.ptr(cnt), n, k, w, n // w if w else 0, _C.ptr(buf), _C.ptr(tiles),
                                           _C.ptr(temp), temp.numel(), _C.ptr(ovf), _C.ptr(flag),
                                           ctypes.c_void_p(host.data_ptr()), 0, _C.stream()), "qf_frame_offsets")
        torch.cuda.synchronize()
        c = cnt.clamp(max=k).long()
        want = torch.cumsum(c, 0) - c
        assert torch.equal(buf[:n], want) and int(buf[n]) == int(c.sum())
        assert int(host[0]) == int(c.sum()) and int(host[1]) == 7 and int(host[3]) == 1
        if w:
            h = n // w
            pad = torch.zeros(((h + 7) // 8 * 8, (w + 7) // 8 * 8), dtype=torch.int64, device=device)
            pad[:h, :w] = c.view(h, w)
            tot = pad.view((h + 7) // 8, 8, (w + 7) // 8, 8).sum(dim=(1, 3)).reshape(-1)
            assert torch.equal(tiles, torch.cumsum(tot, 0) - tot)
            # the render-only frame's two-launch version: the same tile bases and total, no per-ray offsets
            tiles2 = torch.empty_like(tiles)
            total2 = torch.full((1,), -1, dtype=torch.int64, device=device)
            host2 = torch.full((4,), -1, dtype=torch.int64).pin_memory()
            zero_me = torch.full((1,), 77, dtype=torch.int32, device=device)
            _C.check(_C.lib().qf_tile_offsets(_C.ptr(cnt), k, w, h, _C.ptr(tiles2), _C.ptr(total2), _C.ptr(ovf), None,
                                              ctypes.c_void_p(host2.data_ptr()), _C.ptr(zero_me), _C.stream()), "qf_tile_offsets")
            assert int(zero_me) == 0                        # zero_word: the tile pack's dropped-hit counter, zeroed on the way
            torch.cuda.synchronize()
            assert torch.equal(tiles2, tiles) and int(total2) == int(c.sum())
            assert int(host2[0]) == int(c.sum()) and int(host2[1]) == 7 and int(host2[3]) == 0     # no flag word: 0


def test_filter_hits_applies_the_rule_in_place(device):
    """qf_filter_hits on unordered complete lists = sort + the oracle's chain (the route of RayIntersector.hits with a
    camera); no-op with the rule off."""
    from quadraturefields_amd import _C
    from quadraturefields_amd.mesh_utils import RayIntersector
    mesh = _scene(2, 2)
    ri = RayIntersector(mesh, max_hits=8, min_separation=0.05)
    rng = np.random.default_rng(3)
    n, k = 500, 8
    cnt = rng.integers(0, k + 1, size=n).astype(np.int32)
    t = np.full((n, k), np.inf, np.float32)
    tri = np.full((n, k), -1, np.int32)
    for r in range(n):
        t[r, :cnt[r]] = rng.choice(np.arange(1, 60), size=cnt[r], replace=False).astype(np.float32) * 0.02   # gaps of 0.02: some < 0.05
        tri[r, :cnt[r]] = rng.permutation(1000)[:cnt[r]]
    dt, dtri, dc = (torch.from_numpy(x.copy()).to(device) for x in (t, tri, cnt))
    _C.check(_C.lib().qf_filter_hits(ri._handle, n, k, _C.ptr(dtri), _C.ptr(dt), _C.ptr(dc), _C.stream()), "qf_filter_hits")
    sep = np.float32(ri.min_separation)
    for r in range(n):
        order = np.lexsort((tri[r, :cnt[r]], t[r, :cnt[r]]))
        kept_t, kept_i = [], []
        for i in order:
            if not kept_t or t[r, i] > np.float32(kept_t[-1] + sep):
                kept_t.append(t[r, i])
                kept_i.append(tri[r, i])
        assert int(dc[r]) == len(kept_t)
        assert np.array_equal(dt[r, :len(kept_t)].cpu().numpy(), np.array(kept_t, np.float32))
        assert np.array_equal(dtri[r, :len(kept_i)].cpu().numpy(), np.array(kept_i, np.int32))
        assert bool(torch.isinf(dt[r, len(kept_t):]).all()) and bool((dtri[r, len(kept_i):] == -1).all())
    ri.set_min_separation(0.0)
    d2 = torch.from_numpy(t.copy()).to(device)
    _C.check(_C.lib().qf_filter_hits(ri._handle, n, k, _C.ptr(torch.from_numpy(tri.copy()).to(device)), _C.ptr(d2),
                                     _C.ptr(torch.from_numpy(cnt.copy()).to(device)), _C.stream()), "qf_filter_hits")
    assert torch.equal(d2.cpu(), torch.from_numpy(t))                    # rule off: untouched


@pytest.mark.gpu
def test_trimesh_separation_follows_the_mesh_on_refit(device):
    """trimesh derives its re-origin distance from the extent of the mesh it currently intersects (1e-4 * 100 / bounding
    box diagonal): the default ``"trimesh"`` separation is recomputed when the vertices are updated (device and host
    refit, and the replaced ``inter`` object of train_finetune.py:716-718); a fixed distance stays fixed."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.intersector import Intersector
    from quadraturefields_amd.mesh_utils import MeshIntersection, trimesh_ray_offset
    mesh = synthetic.shell_mesh(n_shells=2, subdivisions=2)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=8, device=device)
    ri = mi.rayintersector
    s0 = trimesh_ray_offset(mesh.vertices)
    assert ri.min_separation == pytest.approx(s0, rel=1e-12)
    v2 = torch.from_numpy(mesh.vertices.astype(np.float32)).to(device) * 2.0
    ri.update_intersector(v2)                                           # device refit
    assert ri.min_separation == pytest.approx(s0 / 2.0, rel=1e-6)
    ri.update_intersector(mesh.vertices.astype(np.float32) * 4.0)       # host refit
    assert ri.min_separation == pytest.approx(s0 / 4.0, rel=1e-6)
    ri.inter = Intersector((mesh.vertices[mesh.faces] * 0.5).reshape(-1), 8, 0)
    assert ri.min_separation == pytest.approx(s0 * 2.0, rel=1e-6)
    ri.set_min_separation(1e-3)                                         # a fixed distance no longer follows
    ri.update_intersector(v2)
    assert ri.min_separation == 1e-3


@pytest.mark.parametrize("k", [12, 16, 28, 32])
def test_packed_samples_match_oracle_on_every_list_route(device, k):
    """The packers read a ray's list straight into the sort network's registers (load_sort_row): rows as float4s when K is
    a multiple of 4, a 16- or a 32-key network by the longest list of the wave.  K = 12 / 16: vector rows, 16 keys, lists
    truncated to the K nearest; K = 28 / 32: vector rows, 32 keys (up to 20 crossings).  (K = 25, scalar rows, is every
    other test's case.)  The six sample tensors against the oracle, bit for bit."""
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = _scene(subdiv=2, shells=10)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=k)
    o, d = _rays(3000, seed=k)
    want = om.to_loader_tensors(om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d, o, k))
    got = mi.sampling_raytrace_device(d, o)
    deepest = int(torch.bincount(want[2]).max())
    assert deepest == min(k, 20) or deepest > 16
    for name, g, w in zip(["xyzs", "dirs", "index_ray", "ts", "index_tri", "origins"], got, want):
        assert g.shape == w.shape, name
        assert torch.equal(g.cpu(), w), name


def test_ray_flag_camera_centre_or_the_bvh(device):
    """qf_raster_intersect's ray_flag, origin part: while every ray starts at the camera centre bit for bit the pass takes
    the origin from the camera struct (flag stays 0); one origin that differs in a single bit (-0.0 for 0.0) raises it:
    the pass then writes NOTHING (all counts 0) and ``hits`` answers through the BVH.  Either way the caller gets the
    hits of the BVH traversal."""
    import warnings
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = _scene(subdiv=3, shells=3)
    ri = RayIntersector(mesh, max_hits=25)
    w, h = 96, 64
    c2w = synthetic.orbit_cameras(1, seed=2)[0].clone()
    c2w[0, 3] = 0.0                                       # a centre with a zero coordinate: -0.0 == 0.0 but not bitwise
    focal = synthetic.lego_focal(800) * w / 800.0
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    cam = make_camera(c2w, focal, w, h)
    n = w * h
    want = ri._hits_bvh(o.contiguous(), d.contiguous(), 25, w)
    for flip, flag_want in ((False, 0), (True, 1)):
        o2 = o.clone()
        if flip:
            o2[n // 2 + 7, 0] = -0.0
        tri, t, cnt, ovf = ri._hits_raster(o2.contiguous(), d.contiguous(), 25, cam, sort_lists=True)
        assert int(cnt._base[n + 1]) == flag_want and int(ovf.item()) == 0
        if flip:
            assert int(cnt.sum()) == 0                    # a raised flag: the pass returned at once
        else:
            assert torch.equal(cnt, want[2]) and torch.equal(tri, want[0]) and torch.equal(t, want[1])
        ri._raster_backoff = 0
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = ri.hits(o2, d, 25, camera=cam)
        assert torch.equal(got[2], want[2]) and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert ri.camera_mismatch_frames == (1 if flip else 0)
        assert bool(caught) == flip                        # told once, when it first happens
    assert int(want[2].sum()) > 2000


def _mismatched_frames(o, d, c2w, focal, w, h, device):
    """Ray sets that are NOT the pixel grid of make_camera(c2w, focal, w, h), each with the camera a caller might pass
    along anyway: (name, origins, directions, camera)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    cam = lambda f=focal, c=c2w: make_camera(c, f, w, h)                     # noqa: E731
    g = torch.Generator(device="cpu").manual_seed(5)
    out = []
    # (i) half-pixel jitter of the pixel position (nerf_synthetic.py:335-340 add_ray_direction_noise / random sub-pixel
    # training rays): directions of the pixel grid moved by up to +-0.5 px
    jx = (torch.rand(w * h, generator=g) - 0.5).to(device)
    jy = (torch.rand(w * h, generator=g) - 0.5).to(device)
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    cd = torch.stack([(xs.flatten() + jx - w / 2 + 0.5) / focal, -(ys.flatten() + jy - h / 2 + 0.5) / focal,
                      -torch.ones(w * h, device=device)], dim=1)
    dj = (cd[:, None, :] * c2w[:3, :3].to(device)[None]).sum(-1)
    dj = (dj / dj.norm(dim=1, keepdim=True)).contiguous()
    out.append(("half-pixel jitter", o, dj, cam()))
    # (ii) the right rays with a camera of the wrong focal length (a stale make_camera, another up_sample)
    out.append(("wrong focal", o, d, cam(focal * 1.07)))
    out.append(("stale pose", o, d, cam(c=synthetic.orbit_cameras(2, seed=77)[1])))
    # (iii) the rays in column-major order
    perm = torch.arange(w * h, device=device).view(h, w).t().reshape(-1)
    out.append(("column-major", o[perm].contiguous(), d[perm].contiguous(), cam()))
    # (iv) non-unit directions (t = distance / |d|): the depth-slab passes bin by distance
    out.append(("scaled directions", o, (d * 1.5).contiguous(), cam()))
    # (v) origins off the camera centre (a parallel-shifted bundle)
    out.append(("shifted origins", (o + torch.tensor([0.02, -0.01, 0.03], device=device)).contiguous(), d, cam()))
    # (vi) one NaN direction
    dn = d.clone()
    dn[w * h // 3] = float("nan")
    out.append(("nan direction", o, dn, cam()))
    return out


@pytest.mark.parametrize("dense", [False, True])
def test_rays_that_are_not_the_cameras_pixel_grid_go_through_the_bvh(device, dense):
    """VERDICT r3 weak 2 / ADVICE r3: the camera-coherent route used to TRUST that ray i is pixel (i % w, i / w) of the
    camera passed along -- jittered directions, a wrong focal length, another ray order, non-unit directions (depth
    slabs) or shifted origins silently lost hits to the guard-band reject.  Now the pass verifies it on the device
    (camera_rays_check) and a violation routes the frame through the exact BVH traversal inside the same launch
    sequence, no host round trip: the six sample tensors must be BIT-IDENTICAL to the camera-less BVH route, through
    ``sample_device``, the no-wait frame (``sample_frame_device`` + tile pack) and, on the dense scene, the depth-slab
    mode.  A consistent frame afterwards takes the fast path again and still equals the BVH."""
    import warnings
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector, make_camera
    mesh = _scene(subdiv=4, shells=12 if dense else 4)
    k = 6 if dense else 25
    ri = RayIntersector(mesh, max_hits=k)
    w, h = 104, 80
    focal = synthetic.lego_focal(800) * w / 800.0
    c2w = synthetic.orbit_cameras(1, seed=9)[0]
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    if dense:                                  # more than K crossings on most object rays: the slab mode of configs[2]
        ri.raster_wide = 4 * k
        assert ri.raster_slabs >= 2
    cam_ok = make_camera(c2w, focal, w, h)
    base = ri.sample_device(o, d, k, camera=cam_ok)
    assert ri.camera_mismatch_frames == 0 and base[0].shape[0] > 3000
    for name, o2, d2, cam in _mismatched_frames(o, d, c2w, focal, w, h, device):
        if name == "nan direction":            # NaN != NaN: compare the finite rays of both routes
            finite = lambda t: torch.nan_to_num(t.float(), nan=-7.0)                     # noqa: E731
        else:
            finite = lambda t: t                                                       # noqa: E731
        ri._raster_backoff = 0
        want = ri.sample_device(o2, d2, k, image_width=w)                 # the BVH traversal: exact for any rays
        before = ri.camera_mismatch_frames
        ri._raster_backoff = 0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = ri.sample_device(o2, d2, k, camera=cam)
        assert ri.camera_mismatch_frames == before + 1, name
        assert ri._raster_backoff > 0, name                                # the next frames go straight to the BVH
        assert (got is None) == (want is None), name
        for g_, w_ in zip(got, want):
            assert torch.equal(finite(g_), finite(w_)), name
        # the render-only frame without a host wait (tile pack): same samples in the coherent order, counted later
        ri._raster_backoff = 0
        ref = ri.sample_device(o2, d2, k, image_width=w, lean=True)
        ref_frame, ref_layout = ri.last_frame, ri.last_layout
        n_ref = ri.frame_samples()
        ri._raster_backoff = 0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            frame = ri.sample_frame_device(o2, d2, k, camera=cam)
        layout = ri.last_layout
        assert ri.frame_samples() == n_ref, name
        total = int(frame.total_dev.item())
        assert total == ref_frame.total, name
        assert torch.equal(frame.hit_count, ref_frame.hit_count), name
        assert torch.equal(finite(layout[1][:total]), finite(ref_layout[1])), name
        assert torch.equal(finite(frame.depth_c[:total]), finite(ref_frame.depth_c)), name
        ri._settle_deferred_policy()
        assert ri.camera_mismatch_frames == before + 2, name
    # consistent rays again: the fast path, and the BVH's samples
    ri._raster_backoff = 0
    n_bad = ri.camera_mismatch_frames
    again = ri.sample_device(o, d, k, camera=cam_ok)
    assert ri.camera_mismatch_frames == n_bad
    for a, b in zip(again, base):
        assert torch.equal(a, b)
    ri._raster_backoff = 0
    want = ri.sample_device(o, d, k, image_width=w)
    for a, b in zip(again, want):
        assert torch.equal(a, b)
