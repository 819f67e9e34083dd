"""GPU tests for the callers on either side of the path: device ray generation,
the evaluation loader, texture baking producer, triangle pruning."""
import os

import numpy as np
import pytest
import torch

from oracle import fields as ofields
from oracle import meshpath as om
from oracle import quantize as oq
from tests import helpers

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_generate_rays_vs_reference_fixture(device):
    """qf_generate_rays against the rays produced by the reference's own fetch_data body (rays_ref.npz)."""
    from quadraturefields_amd.datasets.nerf_synthetic import generate_rays
    z = np.load(os.path.join(GOLD, "rays_ref.npz"))
    o, d, cam = generate_rays(torch.from_numpy(z["c2w"]), float(z["focal"]), int(z["width"]), int(z["height"]))
    assert torch.equal(o.cpu(), torch.from_numpy(z["origins"]))
    # same individually rounded operations; the only freedom is torch's reduction order inside norm(): <= 1 ulp
    assert torch.allclose(d.cpu(), torch.from_numpy(z["viewdirs"]), rtol=0, atol=1.2e-7)
    assert (cam.width, cam.height) == (int(z["width"]), int(z["height"]))
    # 800x800 Lego camera: unit directions, centre pixel looks down -z of the camera
    from quadraturefields_amd import synthetic
    c2w = synthetic.orbit_cameras(1)[0]
    o, d, _ = generate_rays(c2w, synthetic.lego_focal(800), 800, 800)
    o_t, d_t = synthetic.camera_rays(c2w, synthetic.lego_focal(800), 800, 800)
    assert torch.equal(o.cpu(), o_t) and torch.allclose(d.cpu(), d_t, rtol=0, atol=1.2e-7)
    assert torch.allclose(d.norm(dim=-1), torch.ones(640000, device=device), atol=1e-6)


def test_loader_item_renders_like_the_frame_renderer(device):
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import area_downsample
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=12)
    field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    w = h = 40
    images = np.zeros((2, h, w, 4), dtype=np.uint8)
    images[..., 3] = 255
    images[..., :3] = 128
    c2w = synthetic.orbit_cameras(2, seed=4).numpy()
    ds = SubjectLoader.from_arrays(images, c2w, synthetic.lego_focal(800) * w / 800.0, mesh_intersect=mi, upsample=2)
    assert len(ds.images) == 2 and (ds.WIDTH, ds.HEIGHT) == (80, 80) and not ds.training
    item = ds[1]
    assert item["pixels"].shape == (h * w, 3) and item["rays"].origins.shape == (80 * 80, 3)
    out = utils.render_image_finetune_with_occgrid(field, None, None, item["rays"], item["data"], render_step_size=5e-3,
                                                   mesh_intersect=mi, scaling=0)
    rgb = out[0]
    # oracle on the same rays
    o, d = item["rays"].origins.cpu(), item["rays"].viewdirs.cpu()
    sample = om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d.numpy(), o.numpy(), 25)
    rgb_o = om.render_image_finetune(helpers.oracle_ngp_weights(field), None, om.to_loader_tensors(sample), 80 * 80)[0]
    assert (rgb.cpu() - rgb_o).abs().max().item() <= 2e-4
    img = area_downsample(rgb.reshape(80, 80, 3), 2)
    assert img.shape == (40, 40, 3)


def test_training_batches_of_the_loader(device):
    """num_rays set on a training split (train_finetune.py:296-305): random pixels of random images, intersected on
    the device through the general BVH route; the batch's samples equal the oracle's for the same rays."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.mesh_utils import MeshIntersection
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    w = h = 40
    g = torch.Generator().manual_seed(1)
    images = torch.randint(0, 256, (3, h, w, 4), generator=g, dtype=torch.uint8).numpy()
    c2w = synthetic.orbit_cameras(3, seed=4).numpy()
    ds = SubjectLoader.from_arrays(images, c2w, synthetic.lego_focal(800) * w / 800.0, split="train", num_rays=4096,
                                   mesh_intersect=mi, upsample=2, color_bkgd_aug="random", device=device)
    assert ds.training and ds.images.is_cuda
    item = ds[0]
    o, d = item["rays"].origins, item["rays"].viewdirs
    assert o.shape == (4096, 3) and item["pixels"].shape == (4096, 3) and "order" not in item
    assert torch.allclose(d.norm(dim=-1), torch.ones(4096, device=device), atol=1e-6)
    # every origin is one of the three camera centres
    centres = torch.from_numpy(c2w[:, :3, 3]).to(device)
    assert bool(((o[:, None, :] - centres[None]).abs().sum(-1) == 0).any(dim=1).all())
    xyzs, dirs, index_ray, ts, index_tri, origins = item["data"]
    want = om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d.cpu().numpy(), o.cpu().numpy(), 25)
    assert np.array_equal(index_ray.cpu().numpy(), want[2]) and np.array_equal(index_tri.cpu().numpy(), want[4])
    assert np.allclose(ts.cpu().numpy(), want[3], rtol=0, atol=2e-6)
    # pixels = rgb * a + bkgd * (1 - a) with one random background colour for the batch
    assert item["color_bkgd"].shape == (3,) and float(item["pixels"].min()) >= 0.0 and float(item["pixels"].max()) <= 1.0


def test_scatter_max_and_pruning(device):
    from quadraturefields_amd import baking, synthetic
    g = torch.Generator().manual_seed(0)
    n, f = 200000, 5000
    idx = torch.randint(0, f, (n,), generator=g)
    w = torch.rand(n, generator=g)
    w[::7] = 0.0
    out = torch.zeros(f, device=device)
    baking.triangle_max_weights(w.to(device), idx.to(device), out)
    want = torch.zeros(f).scatter_reduce(0, idx, w, reduce="amax", include_self=True)
    assert torch.equal(out.cpu(), want)
    # accumulating over several views = elementwise maximum (prune_mesh_after_finetuning.py:357)
    w2 = torch.rand(n, generator=g) * 0.5
    baking.triangle_max_weights(w2.to(device), idx.to(device), out)
    assert torch.equal(out.cpu(), torch.maximum(want, torch.zeros(f).scatter_reduce(0, idx, w2, reduce="amax")))
    neg = torch.full((4,), -1.0, device=device)
    baking.triangle_max_weights(torch.tensor([-0.5, -2.0, 3.0], device=device), torch.tensor([0, 0, 1], device=device), neg)
    assert neg.cpu().tolist() == [-0.5, 3.0, -1.0, -1.0]
    mesh = synthetic.shell_mesh(n_shells=2, subdivisions=1)
    tw = torch.rand(mesh.faces.shape[0])
    pruned = baking.prune_faces(mesh, tw, 0.5)
    assert pruned.faces.shape[0] == int((tw > 0.5).sum()) and pruned.vertices.shape == mesh.vertices.shape


def test_bake_texture_images_producer(device):
    """Field -> uint8 texture set (bake_texture_images_shelly.py:284-291) -> decode: codes match the oracle's
    quantisers up to one step where a transcendental lands on a rounding boundary."""
    from quadraturefields_amd import baking, synthetic
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    from quadraturefields_amd.texture_utils import FeatureCompression
    lobes, t = 3, 64
    aabb = [-1.5] * 3 + [1.5] * 3
    sg = NGPRadianceFieldSGNew(aabb=aabb, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=12)
    sg.load_state_dict(synthetic.seeded_ngp_state(12, sg.mlp_base.grid.n_rows, sg_lobes=lobes), strict=False)
    nf = NGPRadianceField(aabb=aabb, log2_hashmap_size=12)
    nf.load_state_dict(synthetic.seeded_ngp_state(12, nf.mlp_base.grid.n_rows, seed=7), strict=False)
    sg, nf = sg.to(device), nf.to(device)
    rng = np.random.default_rng(0)
    V = rng.uniform(-1.2, 1.2, size=(t, t, 3)).astype(np.float32)
    V[rng.random((t, t)) < 0.3] = 0.0                                      # empty texels
    comp = FeatureCompression(lobes, initialize=True, texture_size=t, compression_type="sigmoid", lambda_thres=7.5)
    mask = baking.bake_texture_images(sg, nf, V, comp, batch_size=1000)
    assert mask.sum() == (V.sum(-1) != 0).sum()
    assert int((comp.alpha.cpu().numpy()[~mask] != 0).sum()) == 0          # untouched texels stay zero
    ind = np.argwhere(mask)
    pts = torch.from_numpy(V[ind[:, 0], ind[:, 1]])
    feats = ofields.sg_features(pts, helpers.oracle_ngp_weights(sg))
    feats[:, -1] = ofields.query_density(pts, helpers.oracle_ngp_weights(nf)).flatten()
    want = oq.compress_features(feats, lobes, "sigmoid", 7.5)
    r, c = ind[:, 0], ind[:, 1]
    def off_by(a, b):
        return (a.cpu().to(torch.int16) - b.to(torch.int16)).abs()
    d_alpha = off_by(comp.alpha[r, c], want["alpha"])
    d_diff = off_by(comp.diffuse[r, c], want["diffuse"])
    assert int(d_alpha.max()) <= 1 and float((d_alpha > 0).float().mean()) < 0.02
    assert int(d_diff.max()) <= 1 and float((d_diff > 0).float().mean()) < 0.02
    for i in range(lobes):
        d_col = off_by(comp.sg_colors[i][r, c], want["colors"][i])
        assert int(d_col.max()) <= 1 and float((d_col > 0).float().mean()) < 0.02
        lam = off_by(comp.lambdas[i][r, c][:, 0], want["lambdas"][i][:, 0])
        assert int(lam.max()) <= 1
        # azimuth (channel 1) wraps mod 256 (code 0 and 255 are neighbours on the circle, ngp.py:236-248),
        # elevation (channel 2) does not
        az = (comp.lambdas[i][r, c][:, 1].cpu().to(torch.int16) - want["lambdas"][i][:, 1].to(torch.int16)) % 256
        az = torch.minimum(az, 256 - az)
        assert int(az.max()) <= 1 and float((az > 0).float().mean()) < 0.02
        el = off_by(comp.lambdas[i][r, c][:, 2], want["lambdas"][i][:, 2])
        assert int(el.max()) <= 1 and float((el > 0).float().mean()) < 0.02
    # save / reload round trip of the PNG set
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        comp.save_to_file(tmp + "/")
        back = FeatureCompression(lobes, initialize=False, texture_size=t, path=tmp + "/", compression_type="sigmoid",
                                  lambda_thres=7.5)
        assert torch.equal(back.alpha, comp.alpha) and torch.equal(back.diffuse, comp.diffuse)
        assert all(torch.equal(back.lambdas[i], comp.lambdas[i]) for i in range(lobes))
