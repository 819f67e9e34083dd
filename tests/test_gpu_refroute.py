"""GPU: the pieces behind the reference-shaped eval loop (bench.py ``reference_route``): the split layout derived from
a window's ray ids, the re-sort launch that also writes the streamed copies, ``generate_splits`` on device arrays,
``MeshFinetune.update_d`` as one launch, the hand-written ``qf_sample_offsets`` -- and that the eval loop of
train_finetune.py:575-629, restated over the reference-named entry points, gives the FrameRenderer's pixels."""
import numpy as np
import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu


def _scene(device, log2_T=14, shells=4, subdiv=3):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    mesh = synthetic.shell_mesh(n_shells=shells, subdivisions=subdiv)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=log2_T)
    field.load_state_dict(synthetic.seeded_ngp_state(log2_T, field.mlp_base.grid.n_rows), strict=False)
    return mesh, mi, field.to(device)


def _frame(mi, w, h, device, seed=0):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    c2w = synthetic.orbit_cameras(1, seed=seed)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    return o, d, make_camera(c2w, focal, w, h)


def test_split_layout_equals_the_frame_layout_on_tile_aligned_windows(device):
    """A window of whole tile rows is a contiguous slice of the frame's coherent order: the layout derived from the
    window's ids alone must be that slice, re-based; the whole frame must reproduce the frame's own inverse map."""
    _, mi, _ = _scene(device)
    w, h = 96, 80
    o, d, cam = _frame(mi, w, h, device)
    data = mi.sampling_raytrace_device(d, o, camera=cam)
    ri = mi.rayintersector
    inv_frame = ri.last_layout[0].clone()
    order_frame = ri.last_order.clone()
    index_ray = data[2]
    order, inverse, invalid = ri.split_layout(index_ray, w, h)
    assert int(invalid) == 0
    assert torch.equal(inverse, inv_frame) and torch.equal(order, order_frame)
    for y0, y1 in ((0, 24), (24, 64), (64, 80)):             # multiples of 8 rows
        lo, hi = y0 * w, y1 * w
        a = int(torch.searchsorted(index_ray, torch.tensor(lo, device=device)))
        b = int(torch.searchsorted(index_ray, torch.tensor(hi, device=device)))
        assert b > a
        o_s, inv_s, bad = ri.split_layout(index_ray[a:b], w, h)
        assert int(bad) == 0
        assert torch.equal(inv_s, inv_frame[a:b] - a)
        assert torch.equal(o_s, order_frame[a:b] - a)
        # a permutation of [0, n)
        assert torch.equal(torch.sort(inv_s.long()).values, torch.arange(b - a, device=device))
    # a window that is NOT tile aligned still yields a valid permutation (its own tiles)
    lo, hi = 5 * w + 7, 37 * w + 3
    a = int(torch.searchsorted(index_ray, torch.tensor(lo, device=device)))
    b = int(torch.searchsorted(index_ray, torch.tensor(hi, device=device)))
    o_s, inv_s, bad = ri.split_layout(index_ray[a:b], w, h)
    assert int(bad) == 0 and torch.equal(torch.sort(inv_s.long()).values, torch.arange(b - a, device=device))
    assert torch.equal(o_s.long()[inv_s.long()], torch.arange(b - a, device=device))


def test_split_layout_refuses_unsorted_or_out_of_range_ids(device):
    _, mi, _ = _scene(device)
    ri = mi.rayintersector
    w = h = 32
    ids = torch.tensor([5, 5, 9, 7, 7, 100], dtype=torch.int64, device=device)          # 9 > 7: not ascending
    order, inverse, bad = ri.split_layout(ids, w, h)
    assert int(bad) == 1
    assert inverse.tolist() == list(range(6)) and order.tolist() == list(range(6))
    ids = torch.tensor([1, 2, 3, w * h], dtype=torch.int64, device=device)               # outside the frame
    _, inverse, bad = ri.split_layout(ids, w, h)
    assert int(bad) == 1 and inverse.tolist() == [0, 1, 2, 3]
    ids = torch.tensor([-1, 2, 3], dtype=torch.int64, device=device)
    assert int(ri.split_layout(ids, w, h)[2]) == 1
    # long runs (more samples per ray than any tile step assumes) and empty input
    ids = torch.repeat_interleave(torch.tensor([3, 40, 41, 1000], device=device), torch.tensor([70, 1, 300, 2], device=device))
    order, inverse, bad = ri.split_layout(ids, w, h)
    assert int(bad) == 0 and torch.equal(torch.sort(inverse.long()).values, torch.arange(ids.shape[0], device=device))
    _, inverse, bad = ri.split_layout(ids[:0], w, h)
    assert inverse.numel() == 0 and int(bad) == 0


def test_resort_writes_the_streamed_copies(device):
    """qf_resort_samples(inverse=...): the coherent copies are the re-sorted arrays permuted by the inverse map, incl.
    rays whose depths really change order and a ray longer than the kernel's staging window."""
    _, mi, _ = _scene(device)
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(0, 12, (400,), generator=g)
    counts[17] = 1500                                      # longer than the 1088-sample staging window
    index_ray = torch.repeat_interleave(torch.arange(400), counts).to(device)
    n = index_ray.shape[0]
    depth = torch.rand(n, generator=g).to(device)          # unsorted within the rays
    pts, org, vec = (torch.randn(n, 3, generator=g).to(device) for _ in range(3))
    tri = torch.randint(0, 1000, (n,), generator=g).to(device)
    ref = mi.sampling_indexing(pts, org, vec, index_ray, depth, tri)
    _, inverse, bad = mi.rayintersector.split_layout(index_ray, 20, 20, want_order=False)
    assert int(bad) == 0
    out = mi.sampling_indexing(pts, org, vec, index_ray, depth, tri, layout_inverse=inverse)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
    p_c, v_c = mi.last_resort_layout
    assert torch.equal(p_c[inverse.long()], out[0]) and torch.equal(v_c[inverse.long()], out[3])
    # sorted within each ray
    same = out[4][1:] == out[4][:-1]
    assert bool((out[5][1:][same] >= out[5][:-1][same]).all())


def test_generate_splits_views_equal_the_masked_copies(device):
    from quadraturefields_amd import utils
    _, mi, _ = _scene(device)
    w, h = 64, 48
    o, d, cam = _frame(mi, w, h, device, seed=1)
    data = mi.sampling_raytrace_device(d, o, camera=cam)

    def masked(data, num_rays, chunk):                     # the reference's body (train_finetune.py:419-439), restated
        chunks = []
        for i in range(0, num_rays, chunk):
            m = (data[2] < i + chunk) & (data[2] >= i)
            if m.sum() == 0:
                continue
            chunks.append(tuple(t[m].contiguous() for t in data))
        return chunks

    for chunk in (160000, 1000, 517, 64):
        fast = utils.generate_splits(data, w * h, chunk)
        slow = masked(data, w * h, chunk)
        assert len(fast) == len(slow) > 0
        for a, b in zip(fast, slow):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
    # ids that do not ascend take the masks (same result as the reference's body)
    perm = torch.randperm(data[2].shape[0], device=device)
    shuffled = [t[perm] for t in data]
    fast, slow = utils.generate_splits(shuffled, w * h, 1000), masked(shuffled, w * h, 1000)
    assert len(fast) == len(slow) and all(torch.equal(x, y) for a, b in zip(fast, slow) for x, y in zip(a, b))
    # host tensors (the reference's DataLoader hands those over) take the masks too
    host = [t.cpu() for t in data]
    assert len(utils.generate_splits(host, w * h, 1000)) == len(slow)


def test_mesh_update_d_matches_index_add(device):
    from quadraturefields_amd.mesh_utils import MeshFinetune
    g = torch.Generator().manual_seed(0)
    n_v, n_f, n = 300, 500, 20000
    verts = torch.randn(n_v, 3, generator=g).numpy()
    faces = torch.randint(0, n_v, (n_f, 3), generator=g).numpy()
    mf = MeshFinetune(verts, faces, 0.04, device=device)
    d = torch.randn(n, 3, generator=g).to(device) * 0.01
    wgt = torch.rand(n, generator=g).to(device)
    tri = torch.randint(0, n_f, (n,), generator=g).to(device)
    cd = torch.zeros(n_f, 3, device=device).index_add_(0, tri, d * wgt[:, None])
    cw = (torch.ones(n_f, device=device) * 1e-8).index_add_(0, tri, wgt)
    mf.update_d(d, wgt, tri)
    assert torch.allclose(mf.cache_d, cd, atol=1e-5) and torch.allclose(mf.cache_w, cw, rtol=1e-5)
    before = mf.cache_d.clone()
    mf.update_d(None, wgt, tri)                            # zero displacement: cache_d untouched, cache_w moves
    assert torch.equal(mf.cache_d, before) and torch.allclose(mf.cache_w, cw + (cw - 1e-8), rtol=1e-5)
    mf.reset_d()
    assert float(mf.cache_d.abs().max()) == 0.0
    # a batch too small for the chunk-merging kernel takes the plain one: same sums
    m = 3000
    mf.update_d(d[:m], wgt[:m], tri[:m])
    cd_s = torch.zeros(n_f, 3, device=device).index_add_(0, tri[:m], (d * wgt[:, None])[:m])
    cw_s = (torch.ones(n_f, device=device) * 1e-8).index_add_(0, tri[:m], wgt[:m])
    assert torch.allclose(mf.cache_d, cd_s, atol=1e-5) and torch.allclose(mf.cache_w, cw_s, rtol=1e-5)
    # heavy duplication inside a chunk (what a frame's samples look like: x-neighbours share triangles) -- merged in LDS
    few = torch.randint(0, 7, (n,), generator=g).to(device)
    mf.reset_d()
    mf.update_d(d, wgt, few)
    cd_f = torch.zeros(n_f, 3, device=device).index_add_(0, few, d * wgt[:, None])
    cw_f = (torch.ones(n_f, device=device) * 1e-8).index_add_(0, few, wgt)
    assert torch.allclose(mf.cache_d, cd_f, rtol=1e-4, atol=1e-4) and torch.allclose(mf.cache_w, cw_f, rtol=1e-4)
    mf.reset_d()
    # ids outside the mesh (the reference's scatter_add raises on them): skipped, counted, and raised by the next
    # update_faces / check_ids -- not silently dropped
    mf.check_ids()
    bad = tri.clone()
    bad[5], bad[77], bad[900] = n_f, -1, n_f + 12345
    mf.update_d(d, wgt, bad)
    keep = torch.ones(n, dtype=torch.bool, device=device)
    keep[[5, 77, 900]] = False
    cd2 = torch.zeros(n_f, 3, device=device).index_add_(0, tri[keep], (d * wgt[:, None])[keep])
    assert torch.allclose(mf.cache_d, cd2, atol=1e-5)
    with pytest.raises(IndexError, match="3 sample"):
        mf.update_faces()
    mf.check_ids()                                         # the count was consumed
    mf.update_d(None, wgt, bad)                            # the weight-only route counts too
    with pytest.raises(IndexError, match="3 sample"):
        mf.check_ids()


def test_sample_offsets_hand_written_scan(device):
    """qf_sample_offsets (no library scan behind it any more): exclusive sums of min(count, K) + the grand total."""
    from quadraturefields_amd import _C
    g = torch.Generator().manual_seed(1)
    for n in (1, 7, 1024, 1025, 70001):
        cnt = torch.randint(-2, 40, (n,), generator=g, dtype=torch.int32).to(device)
        k = 25
        out = torch.empty((n + 1,), dtype=torch.int64, device=device)
        nb = int(_C.lib().qf_sample_offsets_temp_bytes(n))
        temp = torch.empty((nb,), dtype=torch.uint8, device=device)
        _C.check(_C.lib().qf_sample_offsets(_C.ptr(cnt), n, k, _C.ptr(out), _C.ptr(temp), nb, _C.stream()), "qf_sample_offsets")
        c = cnt.clamp(0, k).long()
        ref = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(c, 0)])
        assert torch.equal(out, ref)


@pytest.mark.parametrize("scaling", [0.0, 0.05])
def test_reference_eval_loop_equals_the_frame_renderer(device, scaling):
    """train_finetune.py:575-629 restated over the package's names (SubjectLoader item -> generate_splits ->
    render_image_finetune_with_occgrid per split -> rgb[split[2]] = color[split[2]]) gives the same pixels as the
    tile-order FrameRenderer, and the automatic split layout changes nothing: identical to the same call with the
    layout switched off (ray-major field evaluation)."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune
    from quadraturefields_amd.render import FrameRenderer
    mesh, mi, field = _scene(device)
    w = h = 96
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(2, seed=5)])
    ds = SubjectLoader.from_arrays(np.zeros((2, h, w, 4), np.uint8), cams, synthetic.lego_focal(800) * w / 800.0,
                                   split="test", mesh_intersect=mi, device=device)
    torch.manual_seed(0)
    field_net = Field(scale=1.5, precision=16, log2_T=14, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                      num_features=2, back_prop=False, nl="relu").to(device)
    with torch.no_grad():
        field_net.xyz_encoder.params.uniform_(-0.5, 0.5)
    mf = MeshFinetune(mi.mesh.vertices, mi.mesh.faces, 0.05, device=device)

    def loop(item, auto):
        rays = item["rays"]
        n = rays.origins.shape[0]
        rgb = torch.ones((n, 3), device=device)
        depth = torch.zeros((n,), device=device)
        shape = mi.rayintersector.last_image_shape
        if not auto:
            mi.rayintersector.last_image_shape = None
        for split in utils.generate_splits(item["data"], n, chunk_size=24 * w):
            color, _, dd, _, _, _, _, _, _ = utils.render_image_finetune_with_occgrid(
                field, field_net, None, rays, split, render_step_size=5e-3, render_bkgd=item["color_bkgd"],
                mesh_intersect=mi, mesh_finetune=mf, scaling=scaling)
            rgb[split[2]] = color[split[2]]
            depth[split[2]] = dd.squeeze()[split[2]]
        mi.rayintersector.last_image_shape = shape
        return rgb, depth

    for i in range(2):
        item = ds[i]
        assert mi.rayintersector.last_image_shape == (w, h)
        rgb_a, dep_a = loop(item, True)
        rgb_b, dep_b = loop(item, False)
        assert torch.equal(rgb_a, rgb_b) and torch.equal(dep_a, dep_b)
        rays = item["rays"]
        fr = FrameRenderer(mi, field, field_net=field_net if scaling else None, render_step_size=5e-3)
        rgb_f, _, dep_f, _ = fr.render(rays.origins, rays.viewdirs, scaling=scaling, camera=item["camera"])
        assert torch.equal(rgb_a, rgb_f)
        assert torch.equal(dep_a, dep_f.reshape(-1))


@pytest.mark.parametrize("scaling", [0.0, 0.05])
@pytest.mark.parametrize("window_rows", [24, 20, 96, 200])
def test_loader_items_cut_for_their_windows_give_the_same_frame(device, scaling, window_rows):
    """Round 4: an evaluation item of the device loader carries ONE frame-wide coherent layout whose
    tile grid restarts at every window of ``generate_splits`` (``SampleSet`` / ``qf_frame_offsets`` band_rows), so the
    windows are views handed out without a search or a host wait and ``render_image_finetune_with_occgrid`` streams each
    of them without deriving a layout (and, with scaling = 0, without the identity re-sort).  The assembled frame, the
    per-window 9-tuples' sample tensors and the triangle weights must equal the round-3 route (plain six tensors ->
    searchsorted windows -> qf_split_layout per window) BIT FOR BIT -- for windows that are whole tile rows (24), that cut
    tile rows (20: the up_sample-2 case, 100 rows of 1600), one window = the frame (96) and a window larger than it."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune, SampleSet, SampleWindow
    mesh, mi, field = _scene(device)
    w = h = 96
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(2, seed=6)])
    ds = SubjectLoader.from_arrays(np.zeros((2, h, w, 4), np.uint8), cams, synthetic.lego_focal(800) * w / 800.0,
                                   split="test", mesh_intersect=mi, device=device)
    ds.WINDOW_RAYS = window_rows * w
    torch.manual_seed(0)
    field_net = Field(scale=1.5, precision=16, log2_T=14, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                      num_features=2, back_prop=False, nl="relu").to(device)
    with torch.no_grad():
        field_net.xyz_encoder.params.uniform_(-0.5, 0.5)

    def loop(data, rays, bkgd, mf):
        n = rays.origins.shape[0]
        rgb = torch.ones((n, 3), device=device)
        depth = torch.zeros((n,), device=device)
        outs = []
        splits = utils.generate_splits(data, n, chunk_size=window_rows * w)
        for split in splits:
            out = utils.render_image_finetune_with_occgrid(
                field, field_net, None, rays, split, render_step_size=5e-3, render_bkgd=bkgd, mesh_intersect=mi,
                mesh_finetune=mf, scaling=scaling)
            rgb[split[2]] = out[0][split[2]]
            depth[split[2]] = out[2].squeeze()[split[2]]
            outs.append(out)
        return rgb, depth, outs, splits

    for i in range(2):
        item = ds[i]
        data, rays = item["data"], item["rays"]
        assert isinstance(data, SampleSet) and len(data) == 6 and data.window_rays == window_rows * w
        assert data.cuts[0] == 0 and data.cuts[-1] == data[0].shape[0] and len(data.cuts) == -(-h // window_rows) + 1
        # the layout is a permutation of every window onto itself, and the streamed copies are the samples in that order
        inv = data.inverse.long()
        for a, b in zip(data.cuts[:-1], data.cuts[1:]):
            assert b == a or (int(inv[a:b].min()) == a and int(inv[a:b].max()) == b - 1)
        assert torch.equal(torch.sort(inv).values, torch.arange(inv.shape[0], device=device))
        assert torch.equal(data.xyz_c[inv], data[0]) and torch.equal(data.dirs_c[inv], data[1])
        assert torch.equal(data.order.long()[inv], torch.arange(inv.shape[0], device=device))
        mf_new = MeshFinetune(mi.mesh.vertices, mi.mesh.faces, 0.05, device=device)
        mf_old = MeshFinetune(mi.mesh.vertices, mi.mesh.faces, 0.05, device=device)
        rgb_n, dep_n, outs_n, splits_n = loop(data, rays, item["color_bkgd"], mf_new)
        assert all(isinstance(s, SampleWindow) for s in splits_n)
        plain = tuple(t.clone() for t in data)                    # the reference's six tensors and nothing else
        rgb_o, dep_o, outs_o, splits_o = loop(plain, rays, item["color_bkgd"], mf_old)
        assert not any(isinstance(s, SampleWindow) for s in splits_o) and len(splits_o) == len(splits_n)
        assert torch.equal(rgb_n, rgb_o) and torch.equal(dep_n, dep_o)
        for a, b in zip(outs_n, outs_o):
            assert a[3] == b[3]
            for j in (0, 1, 2, 4, 5, 6, 8):                        # colours, opacities, depths, weights, positions, ids
                assert torch.equal(a[j], b[j]), j
        # update_d saw the same (weight, triangle) pairs; the atomics' order differs
        assert torch.allclose(mf_new.cache_w, mf_old.cache_w, rtol=1e-5, atol=1e-7)
        assert torch.allclose(mf_new.cache_d, mf_old.cache_d, rtol=1e-4, atol=1e-7)
        # a different chunk size than the item was cut for: the general route, same frame
        n = rays.origins.shape[0]
        other = utils.generate_splits(data, n, chunk_size=17 * w)
        assert not any(isinstance(s, SampleWindow) for s in other)
        assert sum(s[0].shape[0] for s in other) == data[0].shape[0]
    # the baked-texture renderer takes the same items (and skips its identity re-sort): covered by
    # test_gpu_render.py::test_baked_texture_render_matches_oracle through plain tuples and below through SampleSets
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    from quadraturefields_amd.texture_utils import FeatureCompression
    lobes, size = 3, 256
    tex = synthetic.random_textures(size, lobes, seed=3)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="sigmoid", lambda_thres=7.5, device=device)
    uv = torch.from_numpy(synthetic.scaled_uv(mesh, size)).to(device)
    sg = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=14).to(device)
    item = ds[0]
    kw = dict(texture=None, uv=uv, render_step_size=5e-3, render_bkgd=item["color_bkgd"], mesh_intersect=mi,
              mesh_finetune=None, scaling=0, discretize=False, compressor=comp)
    a = utils.render_image_bake_texture_images_with_occgrid(sg, item["rays"], item["data"], **kw)
    b = utils.render_image_bake_texture_images_with_occgrid(sg, item["rays"], tuple(t.clone() for t in item["data"]), **kw)
    for j in (0, 1, 2, 4, 5):
        assert torch.equal(a[j], b[j]), j
    assert a[3] == b[3] > 500


def test_eval_rng_advance_can_follow_the_reference(device):
    """SURVEY B-14 (VERDICT r3 missing 3): the reference draws torch.rand((S, 3)) per split even at evaluation
    (examples/utils.py:543-546).  Off by default (the pixels do not depend on it); with
    ``utils.REPRODUCE_EVAL_RNG_ADVANCE`` the device generator ends an evaluation render exactly where one such draw per
    split leaves it, so a seeded train / eval interleave keeps the reference's random stream.  Pixels identical either way."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    mesh, mi, field = _scene(device)
    w = h = 64
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(1, seed=2)])
    ds = SubjectLoader.from_arrays(np.zeros((1, h, w, 4), np.uint8), cams, synthetic.lego_focal(800) * w / 800.0,
                                   split="test", mesh_intersect=mi, device=device)
    item = ds[0]
    splits = utils.generate_splits(item["data"], w * h, chunk_size=16 * w)
    assert len(splits) == 4

    def render_all():
        return [utils.render_image_finetune_with_occgrid(field, None, None, item["rays"], sp, render_step_size=5e-3,
                                                         render_bkgd=item["color_bkgd"], mesh_intersect=mi, scaling=0)[0]
                for sp in splits]

    torch.manual_seed(11)
    before = torch.cuda.get_rng_state(device)
    plain = render_all()
    assert torch.equal(torch.cuda.get_rng_state(device), before)              # default: the stream does not move
    utils.REPRODUCE_EVAL_RNG_ADVANCE = True
    try:
        torch.manual_seed(11)
        drawn = render_all()
        state = torch.cuda.get_rng_state(device)
    finally:
        utils.REPRODUCE_EVAL_RNG_ADVANCE = False
    torch.manual_seed(11)
    for sp in splits:
        torch.rand((sp[0].shape[0], 3), device=device)                        # what the reference's loop draws
    assert torch.equal(state, torch.cuda.get_rng_state(device))
    assert all(torch.equal(a, b) for a, b in zip(plain, drawn))


def test_reference_eval_loop_at_the_scripts_up_sample_2_size(device):
    """The scripts' evaluation size (run_nerfsynthetic_finetune.sh:9: up_sample 2 -> 1600 x 1600 rays per item, 16 windows
    of 160 000 rays = 100 rows: NOT a multiple of the 8-row tiles, so every window's tile grid ends in a partial row) on a
    245 760-triangle scene at K = 25: loader item -> generate_splits -> render_image_finetune_with_occgrid per window ->
    the harness's assembly equals the tile-order FrameRenderer frame BIT FOR BIT (~15 M quadrature points)."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.mesh_utils import MeshIntersection, SampleSet, SampleWindow
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer
    mesh = synthetic.shell_mesh(n_shells=12, subdivisions=5)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25, device=device)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=17)
    field.load_state_dict(synthetic.seeded_ngp_state(17, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(1, seed=4)])
    ds = SubjectLoader.from_arrays(np.zeros((1, 800, 800, 4), np.uint8), cams, synthetic.lego_focal(800), split="test",
                                   mesh_intersect=mi, device=device, upsample=2)
    item = ds[0]
    data, rays = item["data"], item["rays"]
    n = rays.origins.shape[0]
    assert n == 1600 * 1600 and isinstance(data, SampleSet) and data.window_rays == 160000 and len(data.cuts) == 17
    assert data[0].shape[0] > 8_000_000
    rgb = torch.ones((n, 3), device=device)
    depth = torch.zeros((n,), device=device)
    splits = utils.generate_splits(data, n)
    assert len(splits) >= 12 and all(isinstance(s, SampleWindow) for s in splits)
    total = 0
    for split in splits:
        color, _, d, n_s, _, _, _, _, _ = utils.render_image_finetune_with_occgrid(
            field, None, None, rays, split, render_step_size=5e-3, render_bkgd=item["color_bkgd"], mesh_intersect=mi,
            scaling=0)
        rgb[split[2]] = color[split[2]]
        depth[split[2]] = d.squeeze()[split[2]]
        total += n_s
    assert total == data[0].shape[0]
    rgb_f, _, dep_f, n_f = FrameRenderer(mi, field, render_step_size=5e-3).render(rays.origins, rays.viewdirs,
                                                                                  camera=item["camera"])
    assert n_f == total
    assert torch.equal(rgb, rgb_f) and torch.equal(depth, dep_f.reshape(-1))
