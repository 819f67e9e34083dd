"""GPU parity, end to end: whole-image renders through the reference-named entry points vs the CPU oracle.

Tolerance (north_star: stated per-pixel float tolerance, bit-exact intersection ids/counts): per-pixel
|rgb - oracle| <= 2e-4 absolute (values in [0,1]; field outputs agree to ~1e-5 relative and are amplified by up to
exp() in the density and by the 25-sample transmittance product), PSNR(HIP, oracle) >= 70 dB, i.e. far inside the
0.05 dB budget against any ground truth.
"""
import numpy as np
import pytest
import torch

from oracle import fields as ofields
from oracle import meshpath as om
from tests import helpers

pytestmark = pytest.mark.gpu


def _scene(device, lobes=0, log2_T=14, shells=4, subdiv=3):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    mesh = synthetic.shell_mesh(n_shells=shells, subdivisions=subdiv)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    aabb = [-1.5] * 3 + [1.5] * 3
    if lobes:
        field = NGPRadianceFieldSGNew(aabb=aabb, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=log2_T)
    else:
        field = NGPRadianceField(aabb=aabb, log2_hashmap_size=log2_T)
    field.load_state_dict(synthetic.seeded_ngp_state(log2_T, field.mlp_base.grid.n_rows, sg_lobes=lobes), strict=False)
    return mesh, mi, field.to(device)


def _camera(w, h, seed=0):
    from quadraturefields_amd import synthetic
    c2w = synthetic.orbit_cameras(1, seed=seed)[0]
    return synthetic.camera_rays(c2w, synthetic.lego_focal(800) * w / 800.0, w, h)


@pytest.mark.parametrize("lobes,bg", [(0, "white"), (0, "black"), (6, "white")])
def test_finetune_render_matches_oracle(device, lobes, bg):
    from quadraturefields_amd import utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.render import psnr
    mesh, mi, field = _scene(device, lobes)
    w = h = 100
    o, d = _camera(w, h)
    # oracle: brute-force quadrature points -> loader tensors -> splits -> render
    sample = om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d.numpy(), o.numpy(), 25)
    data_o = om.to_loader_tensors(sample)
    wts = helpers.oracle_ngp_weights(field)
    rgb_o, alpha_o, dep_o, n_o, w_o, pts_o, ridx_o, tri_o = om.render_image_finetune(
        wts, None, data_o, w * h, scaling=0.0, bg_color=bg, sg=bool(lobes))
    # product: DataLoader-style host tensors in, the reference's call signature
    data = mi.sampling_raytrace_numpy(d.numpy(), o.numpy(), 0)
    data = om.to_loader_tensors(data)
    for a, b in zip(data, data_o):
        assert torch.equal(a, b)                                   # quadrature points: bit exact
    rays = Rays(origins=o.reshape(h, w, 3), viewdirs=d.reshape(h, w, 3))
    out = utils.render_image_finetune_with_occgrid(field, None, None, rays, data, render_step_size=5e-3,
                                                   mesh_intersect=mi, mesh_finetune=None, scaling=0, bg_color=bg)
    colors, opac, depths, n_samples, weights, positions, index_ray, loss, index_tri = out
    assert colors.shape == (h, w, 3) and opac.shape == (h, w, 1) and depths.shape == (h, w, 1)
    assert n_samples == n_o and torch.equal(index_ray.cpu(), ridx_o) and torch.equal(index_tri.cpu(), tri_o)
    err = (colors.reshape(-1, 3).cpu() - rgb_o).abs().max().item()
    assert err <= 2e-4, err
    assert psnr(colors.reshape(-1, 3).cpu(), rgb_o) >= 70.0
    assert torch.allclose(opac.reshape(-1, 1).cpu(), alpha_o, atol=2e-4)
    assert torch.allclose(depths.reshape(-1, 1).cpu(), dep_o, atol=1e-3)
    assert torch.allclose(weights.cpu(), w_o, atol=2e-4)
    # the render does not depend on how the samples are windowed (train_finetune.py:590-617)
    full = torch.ones(w * h, 3, device=device) if bg != "black" else torch.zeros(w * h, 3, device=device)
    for split in utils.generate_splits(data, w * h, chunk_size=3000):
        c = utils.render_image_finetune_with_occgrid(field, None, None, rays, split, render_step_size=5e-3,
                                                     mesh_intersect=mi, scaling=0, bg_color=bg)[0]
        full[split[2].to(device)] = c.reshape(-1, 3)[split[2].to(device)]
    assert torch.allclose(full, colors.reshape(-1, 3), atol=1e-6)


def test_deformed_render_matches_oracle(device):
    """scaling != 0: deformation field moves the points along their rays and the per-ray order is re-established."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.field import Field
    mesh, mi, field = _scene(device, 0)
    net = Field(scale=1.5, precision=16, log2_T=14, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                num_features=2, back_prop=False, nl="relu")
    net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)
    net = net.to(device)
    w = h = 64
    o, d = _camera(w, h, seed=2)
    data = om.to_loader_tensors(mi.sampling_raytrace_numpy(d.numpy(), o.numpy(), 0))
    scaling = 0.0434 * 3          # larger than the script value so that some neighbours swap order
    rgb_o, alpha_o, dep_o, _, w_o, pts_o, ridx_o, _ = om.render_image_finetune(
        helpers.oracle_ngp_weights(field), helpers.oracle_deform_weights(net), data, w * h, scaling=scaling)
    rays = Rays(origins=o, viewdirs=d)
    out = utils.render_image_finetune_with_occgrid(field, net, None, rays, data, render_step_size=5e-3,
                                                   mesh_intersect=mi, scaling=scaling)
    assert torch.equal(out[6].cpu(), ridx_o)
    assert torch.allclose(out[5].cpu(), pts_o, atol=2e-6)
    assert (out[0].cpu() - rgb_o).abs().max().item() <= 3e-4
    assert out[7].shape == (1,)
    # the frame driver runs the same evaluation inside the intersector's tile order (tile pack, streamed deformation
    # field, displacement + re-sort within the tiles, field, tile compositor): bit-identical pixels, for the script's
    # scaling and for one large enough to swap neighbours
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.render import FrameRenderer
    c2w = synthetic.orbit_cameras(1, seed=2)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    o2, d2 = synthetic.camera_rays(c2w, focal, w, h)
    fr = FrameRenderer(mi, field, field_net=net)
    for sc in (0.0434, scaling, 1.5):
        data2 = mi.sampling_raytrace_device(d2, o2)
        ref = utils.render_image_finetune_with_occgrid(field, net, None, Rays(origins=o2, viewdirs=d2), data2,
                                                       render_step_size=5e-3, mesh_intersect=mi, scaling=sc)
        rgb_f, alpha_f, depth_f, n_f = fr.render(o2.to(device), d2.to(device), scaling=sc, camera=make_camera(c2w, focal, w, h))
        assert n_f == ref[3]
        assert torch.equal(rgb_f, ref[0].reshape(-1, 3)) and torch.equal(alpha_f, ref[1].reshape(-1, 1))
        assert torch.equal(depth_f, ref[2].reshape(-1, 1))
    # (1.5 moves samples by up to a whole scene radius: the per-ray order does change)
    pts = ref[5]
    ridx = ref[6]
    t_before = (data2[0] - data2[5]).norm(dim=-1)
    same = data2[2][1:] == data2[2][:-1]
    assert bool((torch.diff(t_before)[same] >= 0).all())


def test_baked_texture_render_matches_oracle(device):
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.texture_utils import FeatureCompression
    lobes, size = 6, 256
    mesh, mi, field = _scene(device, lobes)
    tex = synthetic.random_textures(size, lobes, seed=1)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="sigmoid", lambda_thres=7.5)
    uv = synthetic.scaled_uv(mesh, size)
    w = h = 100
    o, d = _camera(w, h, seed=4)
    data = om.to_loader_tensors(mi.sampling_raytrace_numpy(d.numpy(), o.numpy(), 0))
    t = {"alpha": torch.from_numpy(tex["alpha"]), "diffuse": torch.from_numpy(tex["diffuse"]),
         "colors": [torch.from_numpy(c) for c in tex["colors"]], "lambdas": [torch.from_numpy(c) for c in tex["lambdas"]]}
    rgb_o, alpha_o, dep_o, n_o, w_o, pts_o, texel_o = om.render_image_bake_texture(
        data, w * h, mesh.vertices, mesh.faces, torch.from_numpy(uv), t, lobes, "sigmoid", 7.5)
    rays = Rays(origins=o.reshape(h, w, 3), viewdirs=d.reshape(h, w, 3))
    out = utils.render_image_bake_texture_images_with_occgrid(
        field, rays, data, uv=torch.from_numpy(uv).to(device), render_step_size=5e-3, mesh_intersect=mi,
        compressor=comp, discretize=False)
    colors, opac, depths, n_samples, weights, positions, rays_out, zero = out
    assert zero == 0 and n_samples == n_o and colors.shape == (h, w, 3)
    assert (colors.reshape(-1, 3).cpu() - rgb_o).abs().max().item() <= 2e-4
    assert torch.allclose(opac.reshape(-1, 1).cpu(), alpha_o, atol=2e-4)
    # the discretize=True branch (quantise-dequantise sigma, utils.py:1070-1072) through the unfused calls
    out2 = utils.render_image_bake_texture_images_with_occgrid(
        field, rays, data, uv=torch.from_numpy(uv).to(device), render_step_size=5e-3, mesh_intersect=mi,
        compressor=comp, discretize=True)
    assert torch.isfinite(out2[0]).all()
    # the whole-frame driver keeps a camera frame in the intersector's tile order (tile pack with triangle ids, texel
    # lookup, shading, tile compositor): the same pixels and sample count as the reference-shaped function, bit for bit
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.render import FrameRenderer
    c2w = synthetic.orbit_cameras(1, seed=4)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    o2, d2 = synthetic.camera_rays(c2w, focal, w, h)
    rays2 = Rays(origins=o2, viewdirs=d2)
    data2 = mi.sampling_raytrace_device(d2, o2)
    ref = utils.render_image_bake_texture_images_with_occgrid(
        field, rays2, data2, uv=torch.from_numpy(uv).to(device), render_step_size=5e-3, mesh_intersect=mi, compressor=comp)
    fr = FrameRenderer(mi, field)
    rgb_f, alpha_f, depth_f, n_f = fr.render_baked(o2.to(device), d2.to(device), torch.from_numpy(uv).to(device), comp,
                                                   camera=make_camera(c2w, focal, w, h))
    assert n_f == ref[3] and mi.rayintersector.last_frame.tri_c is not None
    assert torch.equal(rgb_f, ref[0].reshape(-1, 3)) and torch.equal(alpha_f, ref[1].reshape(-1, 1))
    assert torch.equal(depth_f, ref[2].reshape(-1, 1))
    rgb_g = fr.render_baked(o2.to(device), d2.to(device), torch.from_numpy(uv).to(device), comp, image_width=w)[0]
    assert torch.equal(rgb_g, rgb_f)                       # without a camera: the ray-major route
    # no host wait: the shading launch reads the sample count from device memory, worst-case buffers
    rgb_a, alpha_a, depth_a, frame = fr.render_baked_async(o2.to(device), d2.to(device), torch.from_numpy(uv).to(device), comp,
                                                           make_camera(c2w, focal, w, h))
    assert torch.equal(rgb_a, rgb_f) and torch.equal(alpha_a, alpha_f) and torch.equal(depth_a, depth_f)
    assert mi.rayintersector.frame_samples() == n_f and frame.tri_c.dtype == torch.int32


@pytest.mark.parametrize("w,h,k", [(40, 24, 64), (9, 1, 25), (1, 13, 25), (8, 8, 1), (40, 24, 32), (24, 16, 16)])
def test_tile_order_frame_equals_the_ray_major_route_at_the_edges(device, w, h, k):
    """The frame driver's tile-order path (qf_pack_tiles / qf_composite_tiles) against the function-by-function route on
    the six ray-major tensors, bit for bit: the largest K the ABI allows (64 hits per ray: the tile kernels' LDS
    columns), one-row / one-column images (tiles mostly outside the image), K = 1 (every ray overflows and is
    repaired), plain and deformed."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer
    mesh = synthetic.shell_mesh(n_shells=20, subdivisions=2)          # up to 40 crossings per ray
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=k)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=12)
    field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    net = Field(scale=1.5, precision=16, log2_T=12, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                num_features=2, back_prop=False, nl="relu")
    net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)
    net = net.to(device)
    c2w = synthetic.orbit_cameras(1, seed=6)[0]
    focal = synthetic.lego_focal(800) * max(w, h) / 800.0 * 2.0
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    cam = make_camera(c2w, focal, w, h)
    for fnet, sc in ((None, 0.0), (net, 0.5)):
        fr = FrameRenderer(mi, field, field_net=fnet)
        rgb, alpha, depth, n = fr.render(o, d, scaling=sc, camera=cam)
        data = mi.sampling_raytrace_device(d, o)
        assert data is not None and n == data[0].shape[0] and n > 0
        if k == 64:
            assert int(torch.bincount(data[2]).max()) > 25          # deeper than the usual K
        ref = utils.render_image_finetune_with_occgrid(field, fnet, None, Rays(origins=o, viewdirs=d), data,
                                                       render_step_size=5e-3, mesh_intersect=mi, scaling=sc)
        assert torch.equal(rgb, ref[0].reshape(-1, 3)) and torch.equal(alpha, ref[1].reshape(-1, 1))
        assert torch.equal(depth, ref[2].reshape(-1, 1))


def test_frame_renderer_and_upsample(device):
    """up_sample 2 frame (train_finetune.py:620-627): render at 2x, box-average down, compare with the oracle."""
    from quadraturefields_amd.render import FrameRenderer, area_downsample, psnr
    mesh, mi, field = _scene(device, 0)
    w = h = 96
    o, d = _camera(w, h, seed=5)
    fr = FrameRenderer(mi, field)
    rgb, alpha, depth, n = fr.render(o.to(device), d.to(device), image_width=w)
    sample = om.sampling_raytrace_numpy(om.BruteForceIntersector(mesh.vertices, mesh.faces), d.numpy(), o.numpy(), 25)
    rgb_o = om.render_image_finetune(helpers.oracle_ngp_weights(field), None, om.to_loader_tensors(sample), w * h)[0]
    img = area_downsample(rgb.reshape(h, w, 3), 2).cpu()
    img_o = om.area_downsample(rgb_o.reshape(h, w, 3), 2)
    assert img.shape == (48, 48, 3)
    assert psnr(img, img_o) >= 70.0
    # the streamed fast path (image-shaped rays) and the reference-shaped path (no image width: sampling_indexing,
    # field on the ray-major arrays) give the same pixels, bit for bit
    rgb_b, alpha_b, depth_b, n_b = fr.render(o.to(device), d.to(device))
    assert n_b == n and torch.equal(rgb_b, rgb) and torch.equal(alpha_b, alpha) and torch.equal(depth_b, depth)
    # nothing in view
    rgb, alpha, depth, n = fr.render(o.to(device) + 100.0, d.to(device))
    assert n == 0 and bool((rgb == 1).all()) and float(alpha.sum()) == 0.0


def test_crop_fixture_on_the_gpu(device):
    """The committed 100x100 crop image (tests/golden/crop_ref.npz, written by the CPU oracle: configs[0]) against the
    HIP path on the same rays: ids bit-exact, pixels within the stated 2e-4."""
    import os
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.render import FrameRenderer, psnr
    from tests.golden import gen_crop
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop_ref.npz"))
    mesh, field, idx, o, d = gen_crop.scene()
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    data = mi.sampling_raytrace_device(d.to(device), o.to(device), image_width=gen_crop.CROP)
    assert np.array_equal(data[2].cpu().numpy(), z["index_ray"]) and np.array_equal(data[4].cpu().numpy(), z["index_tri"])
    rgb, alpha, depth, n = FrameRenderer(mi, field.to(device)).render(o.to(device), d.to(device), image_width=gen_crop.CROP)
    assert n == int(z["n_samples"])
    assert float((rgb.cpu() - torch.from_numpy(z["rgb"])).abs().max()) <= 2e-4
    assert psnr(rgb.cpu(), torch.from_numpy(z["rgb"])) >= 70.0
    assert float((alpha.cpu() - torch.from_numpy(z["alpha"])).abs().max()) <= 2e-4


@pytest.mark.parametrize("seeded", [True, False])
def test_config3_bf16_dense_shell_frame_matches_the_bf16_oracle(device, seeded):
    """BASELINE configs[2] end to end at test size: a dense-shell scene (most object rays meet more than K = 6
    triangles), the camera-coherent pass with WIDE candidate lists -> K-nearest selection -> per-ray repair -> pack ->
    field_kernel_bf16 -> compositing, against brute-force quadrature points (bit-exact) and the bf16 oracle on them.
    Pixel tolerance 1e-2: bf16 field outputs agree to 2e-3 (test_bf16_ngp_matches_bf16_oracle) and are amplified by the
    density exponential and the transmittance product; fp32-vs-bf16 itself moves pixels by several 1e-2.
    ``seeded``: the dense mode is chosen BEFORE the first frame from the mesh's depth complexity (round 4: no first-frame
    cliff, no frame ever repaired through the BVH); unseeded, it is learned from the first frame's overflow count as in
    rounds 2-3 -- both must give the same, exact samples."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer, psnr
    K, w, h = 6, 128, 72
    mesh = synthetic.shell_mesh(n_shells=10, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=K)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=15)
    field.load_state_dict(synthetic.seeded_ngp_state(15, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    field.compute_dtype = "bf16"
    fr = FrameRenderer(mi, field)
    focal = synthetic.lego_focal(800) * w / 800.0
    wts = helpers.oracle_ngp_weights(field)
    bf = om.BVHIntersector(mesh.vertices, mesh.faces)
    ri = mi.rayintersector
    assert ri.depth_complexity > 0.5 * K                    # 10 shells: ~4.7 crossings per ray through the bounding box
    if not seeded:
        ri.SEED_WIDE_AT = float("inf")
    seen_wide = False
    for i, c2w in enumerate(synthetic.orbit_cameras(3, seed=8)):
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        ri._seed_policy(K)
        assert (ri.raster_wide > K) == (seeded or i > 0)    # the mode frame i runs in: seeded -> wide from frame 0 on;
        rgb, alpha, depth, n = fr.render(o.to(device), d.to(device), camera=make_camera(c2w, focal, w, h))     # learned -> from frame 1
        seen_wide = seen_wide or ri.raster_wide > K
        sample = om.sampling_raytrace_numpy(bf, d.numpy(), o.numpy(), K)
        data = om.to_loader_tensors(sample)
        assert n == data[0].shape[0]
        got = ri.sample_device(o.to(device), d.to(device), K, camera=make_camera(c2w, focal, w, h))
        for a, b in zip(got, data):
            assert torch.equal(a.cpu(), b)                      # quadrature points bit-exact through the wide route
        rgbs, sig = ofields.ngp_forward_bf16(data[0], data[1], wts)
        rgb_o = om.volrend.derive_properties(rgbs, sig.squeeze(-1), data[3], torch.full_like(data[3], 5e-3),
                                             om.volrend.mark_pack_boundaries(data[2]), data[2], bg_color="white",
                                             render_bkgd=None, N=w * h)[0]
        assert float((rgb.cpu() - rgb_o).abs().max()) <= 1e-2
        assert psnr(rgb.cpu(), rgb_o) >= 50.0
    assert seen_wide                                        # the dense-scene policy (wide lists) was exercised
    assert (ri.repaired_frames >= 1) == (not seeded)       # ... and only the learned route pays a repaired first frame


def test_config5_baked_render_at_full_texture_size(device):
    """BASELINE configs[4] at its full texture size (4096^2, L = 6; 1.07 GB of texel records): the packed-record shade
    equals the planar one bit for bit on a frame's samples, untextured rays come out white with alpha 0, the render is
    invariant to how the frame is split into bands, and a 64x64 crop agrees with the oracle's baked render."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    from quadraturefields_amd.render import FrameRenderer
    from quadraturefields_amd.texture_utils import FeatureCompression
    lobes, size, w, h = 6, 4096, 256, 256
    mesh = synthetic.shell_mesh(n_shells=4, subdivisions=4)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    tex = synthetic.random_textures(size, lobes, seed=2)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="sigmoid", lambda_thres=7.5, device=device)
    uv_np = synthetic.scaled_uv(mesh, size)
    uv = torch.from_numpy(uv_np).to(device)
    sg = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=12).to(device)
    fr = FrameRenderer(mi, sg)
    c2w = synthetic.orbit_cameras(1, seed=6)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    rgb, alpha, depth, n = fr.render_baked(o, d, uv, comp, camera=make_camera(c2w, focal, w, h))
    assert comp.records().shape == (size * size, 64) and n > 50000
    assert torch.isfinite(rgb).all() and float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    miss = alpha.reshape(-1) == 0
    assert bool(miss.any()) and bool((rgb[miss] == 1).all())
    # packed records vs the reference's planes on the frame's samples
    data = mi.sampling_raytrace_device(d, o, camera=make_camera(c2w, focal, w, h), layout=False)
    texel = utils.texel_indices(mi, uv, data[0], data[4], size)
    a = comp.shade(texel, data[1], packed=True)
    b = comp.shade(texel, data[1], packed=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # two bands = the frame
    from quadraturefields_amd import parallel
    parts = []
    for y0, y1 in ((0, 104), (104, h)):
        r, al, de, _ = fr.render_baked(o[y0 * w:y1 * w], d[y0 * w:y1 * w], uv, comp,
                                       camera=parallel.band_camera(c2w, focal, w, h, y0, y1))
        parts.append(torch.cat([r, al, de], dim=1))
    assert torch.equal(torch.cat(parts), torch.cat([rgb, alpha, depth], dim=1))
    # a crop against the oracle (uint8 textures on the host)
    ys, xs = torch.meshgrid(torch.arange(96, 160), torch.arange(96, 160), indexing="ij")
    idx = (ys * w + xs).reshape(-1)
    oc, dc = o.cpu()[idx], d.cpu()[idx]
    sample = om.sampling_raytrace_numpy(om.BVHIntersector(mesh.vertices, mesh.faces), dc.numpy(), oc.numpy(), 25)
    t = {"alpha": torch.from_numpy(tex["alpha"]), "diffuse": torch.from_numpy(tex["diffuse"]),
         "colors": [torch.from_numpy(c) for c in tex["colors"]], "lambdas": [torch.from_numpy(c) for c in tex["lambdas"]]}
    rgb_o = om.render_image_bake_texture(om.to_loader_tensors(sample), idx.shape[0], mesh.vertices, mesh.faces,
                                         torch.from_numpy(uv_np), t, lobes, "sigmoid", 7.5)[0]
    assert float((rgb.cpu()[idx] - rgb_o).abs().max()) <= 2e-4


def test_frame_on_duplicated_shells_applies_the_rule_in_the_tile_pack(device):
    """A mesh whose every face exists twice (the reference concatenates two iso-surfaces, marching_cubes.py:81): the
    frame renderer's tile pack applies the re-origin rule on its sorted lists, the pixels and the sample count are the
    oracle's (which skips the second copy of every crossing), and no frame is packed twice."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer, psnr
    base = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    nv = base.vertices.shape[0]
    mesh = TriMesh(np.concatenate([base.vertices, base.vertices]), np.concatenate([base.faces, base.faces + nv]))
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=14)
    field.load_state_dict(synthetic.seeded_ngp_state(14, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    fr = FrameRenderer(mi, field)
    w = h = 80
    focal = synthetic.lego_focal(800) * w / 800.0
    ri = mi.rayintersector
    wts = helpers.oracle_ngp_weights(field)
    bf = om.BVHIntersector(mesh.vertices, mesh.faces)
    for i, c2w in enumerate(synthetic.orbit_cameras(3, seed=12)):
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        rgb, alpha, depth, n = fr.render(o.to(device), d.to(device), camera=make_camera(c2w, focal, w, h))
        data = om.to_loader_tensors(om.sampling_raytrace_numpy(bf, d.numpy(), o.numpy(), 25))
        rgb_o = om.render_image_finetune(wts, None, data, w * h)[0]
        assert n == data[0].shape[0]
        assert float((rgb.cpu() - rgb_o).abs().max()) <= 2e-4 and psnr(rgb.cpu(), rgb_o) >= 70.0
        assert ri.rule_redone_frames == 0 and ri.last_frame.total > 1.5 * n      # slots of both copies, samples of one
    every = om.to_loader_tensors(om.sampling_raytrace_numpy(
        om.BVHIntersector(mesh.vertices, mesh.faces, min_separation=0.0), d.numpy(), o.numpy(), 25))
    assert every[0].shape[0] > 1.5 * data[0].shape[0]          # the rule did remove the second copies


def test_config3_bf16_frame_at_k25_and_t21_on_a_crop(device):
    """BASELINE configs[2] at ITS OWN parameters -- K = 25, T = 2^21 (22 565 520 rows), 1920x1080, bf16 tables + MLPs,
    a dense-shell mesh on which most object rays meet far more than 25 triangles -- checked on a centre crop: the full
    frame is rendered by the HIP path (wide candidate lists -> K-nearest selection -> tile pack -> field_kernel_bf16 ->
    tile compositor), the crop's rays go through the oracle (host BVH multi-hit, bf16 field, compositing).  Ray and
    triangle ids of the crop's quadrature points bit exact; pixels within the bf16 bar of the test-size frame (1e-2,
    PSNR >= 50 dB)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer, psnr
    K, w, h, log2_t = 25, 1920, 1080, 21
    mesh = synthetic.shell_mesh(n_shells=36, subdivisions=4)             # 184 320 triangles in 36 thin shells
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=K)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=log2_t)
    field.load_state_dict(synthetic.seeded_ngp_state(log2_t, field.mlp_base.grid.n_rows), strict=False)
    assert field.mlp_base.grid.n_rows == 22565520
    wts = helpers.oracle_ngp_weights(field)
    field = field.to(device)
    field.compute_dtype = "bf16"
    fr = FrameRenderer(mi, field)
    focal = synthetic.lego_focal(w)
    ri = mi.rayintersector
    bvh = om.BVHIntersector(mesh.vertices, mesh.faces)
    cw, ch = 96, 64
    y0, x0 = (h - ch) // 2, (w - cw) // 2 - 40
    idx = (torch.arange(y0, y0 + ch)[:, None] * w + torch.arange(x0, x0 + cw)[None, :]).reshape(-1)
    cams = synthetic.orbit_cameras(2, seed=12)
    for i, c2w in enumerate(cams):       # frame 0 discovers the overflow (BVH repair), frame 1 runs the wide lists
        o, d = synthetic.camera_rays(c2w, focal, w, h)
        cam = make_camera(c2w, focal, w, h)
        rgb, alpha, depth, n = fr.render(o.to(device), d.to(device), camera=cam)
    assert ri.raster_wide > K
    sample = om.sampling_raytrace_numpy(bvh, d[idx].numpy(), o[idx].numpy(), K)
    data = om.to_loader_tensors(sample)
    assert data[0].shape[0] > 10 * idx.shape[0]                         # a dense scene: > 10 points per crop ray on average
    full = ri.sample_device(o.to(device), d.to(device), K, camera=cam)
    remap = torch.full((w * h,), -1, dtype=torch.int64, device=device)
    remap[idx.to(device)] = torch.arange(idx.shape[0], device=device)
    local = remap[full[2]]
    keep = local >= 0
    assert torch.equal(local[keep].cpu(), data[2]) and torch.equal(full[4][keep].cpu(), data[4])
    assert torch.equal(full[0][keep].cpu(), data[0]) and torch.equal(full[3][keep].cpu(), data[3])
    rgbs, sig = ofields.ngp_forward_bf16(data[0], data[1], wts)
    rgb_o = om.volrend.derive_properties(rgbs, sig.squeeze(-1), data[3], torch.full_like(data[3], 5e-3),
                                         om.volrend.mark_pack_boundaries(data[2]), data[2], bg_color="white",
                                         render_bkgd=None, N=idx.shape[0])[0]
    got = rgb.cpu()[idx]
    assert float((got - rgb_o).abs().max()) <= 1e-2
    assert psnr(got, rgb_o) >= 50.0


@pytest.mark.parametrize("scaling", [0.0, 0.05])
def test_frame_without_host_wait_equals_the_frame_with_it(device, scaling):
    """FrameRenderer.render_async: the sample count never reaches the host (the field / deformation kernels read it from
    device memory, the arrays are worst-case buffers).  Pixels, alpha and depth bit-identical to ``render``; the sample
    count is still available afterwards; a frame that hits nothing is the background; several frames can be enqueued
    back to back before anything is read."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.render import FrameRenderer
    mesh, mi, field = _scene(device)
    torch.manual_seed(0)
    field_net = Field(scale=1.5, precision=16, log2_T=14, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                      num_features=2, back_prop=False, nl="relu").to(device)
    field_net.xyz_encoder.params.data.uniform_(-0.5, 0.5)
    fr = FrameRenderer(mi, field, field_net=field_net if scaling else None)
    w, h = 120, 88
    focal = synthetic.lego_focal(800) * w / 800.0
    cams = synthetic.orbit_cameras(3, seed=21)
    queued = []
    for c2w in cams:                                   # three frames enqueued before any result is looked at
        o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
        cam = make_camera(c2w, focal, w, h)
        queued.append((o, d, cam, fr.render_async(o, d, cam, scaling=scaling)))
    ri = mi.rayintersector
    n_last = ri.frame_samples()
    assert ri.frame_samples(queued[-1][3][3]) == n_last              # by frame: the newest one still owns its count words
    with pytest.raises(RuntimeError, match="a later frame has since reused"):
        ri.frame_samples(queued[0][3][3])                            # an older one does not: refused, not another frame's count
    for o, d, cam, (rgb_a, alpha_a, depth_a, frame) in queued:
        rgb, alpha, depth, n = fr.render(o, d, scaling=scaling, camera=cam)
        assert torch.equal(rgb, rgb_a) and torch.equal(alpha, alpha_a) and torch.equal(depth, depth_a)
        assert frame.total_dev.dtype == torch.int64 and frame.depth_c.shape[0] == w * h * 25      # worst-case buffers
    assert n_last == n > 1000
    # nothing hit: background, no samples
    o, d = synthetic.camera_rays(cams[0], focal, w, h, device=device)
    rgb_a, alpha_a, _, _ = fr.render_async(o + 100.0, d, make_camera(cams[0], focal, w, h), scaling=scaling)
    assert torch.equal(rgb_a, torch.ones_like(rgb_a)) and float(alpha_a.abs().max()) == 0.0
    assert ri.frame_samples() == 0


@pytest.mark.parametrize("packed", [False, True])
def test_one_call_frame_equals_the_separately_bound_launches(device, packed):
    """qf_frame_render (FrameRenderer.render_async's default route for an NGP field): the frame's six launches behind ONE
    bound call.  Pixels, per-pixel counts, sample arrays and the sample count equal those of the same launches bound one
    by one (the route taken when ``fused_frame_ready`` says no), bit for bit -- also for a camera that sees a band of
    the image (triangle-chunk culling on)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.parallel import band_camera
    from quadraturefields_amd.render import FrameRenderer
    mesh, mi, field = _scene(device)
    fr = FrameRenderer(mi, field)
    ri = mi.rayintersector
    w, h = 136, 96
    focal = synthetic.lego_focal(800) * w / 800.0
    c2w = synthetic.orbit_cameras(2, seed=5)[1]
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    full = make_camera(c2w, focal, w, h)
    band = band_camera(c2w, focal, w, h, 24, 72)
    for cam, sl in ((full, slice(0, w * h)), (band, slice(24 * w, 72 * w))):
        ob, db = o[sl].contiguous(), d[sl].contiguous()
        assert ri.fused_frame_ready(cam, mi.num_intersections)
        one = fr.render_async(ob, db, cam, packed=packed)
        n_one = ri.frame_samples()
        f_one = one[3]
        ready = ri.fused_frame_ready
        ri.fused_frame_ready = lambda *a, **k: False
        try:
            sep = fr.render_async(ob, db, cam, packed=packed)
        finally:
            ri.fused_frame_ready = ready
        n_sep = ri.frame_samples()
        f_sep = sep[3]
        assert n_one == n_sep > 500
        for a, b in zip(one[:3], sep[:3]):
            assert (a is None and b is None) or torch.equal(a, b)
        assert torch.equal(f_one.hit_count, f_sep.hit_count) and torch.equal(f_one.tile_base, f_sep.tile_base)
        total = int(f_one.total_dev.item())
        assert torch.equal(f_one.depth_c[:total], f_sep.depth_c[:total])
    # the spherical-Gaussian field takes the same call (head_sg)
    _, mi_sg, field_sg = _scene(device, lobes=6)
    fr_sg = FrameRenderer(mi_sg, field_sg)
    ri_sg = mi_sg.rayintersector
    one = fr_sg.render_async(o, d, full, packed=packed)
    ready = ri_sg.fused_frame_ready
    ri_sg.fused_frame_ready = lambda *a, **k: False
    try:
        sep = fr_sg.render_async(o, d, full, packed=packed)
    finally:
        ri_sg.fused_frame_ready = ready
    for a, b in zip(one[:3], sep[:3]):
        assert (a is None and b is None) or torch.equal(a, b)
    ref = fr_sg.render(o, d, camera=full)
    if not packed:
        assert torch.equal(one[0], ref[0]) and torch.equal(one[2], ref[2])
    # a caller-supplied background colour and the black one go through the job's bg_mode / bkgd fields
    bk = torch.tensor([0.2, 0.5, 0.7], device=device)
    for colour, bkgd in (("custom", bk), ("black", None)):
        fr2 = FrameRenderer(mi, field, bg_color=colour)
        one = fr2.render_async(o, d, full, render_bkgd=bkgd, packed=packed)
        ready = ri.fused_frame_ready
        ri.fused_frame_ready = lambda *a, **k: False
        try:
            sep = fr2.render_async(o, d, full, render_bkgd=bkgd, packed=packed)
        finally:
            ri.fused_frame_ready = ready
        for a, b in zip(one[:3], sep[:3]):
            assert (a is None and b is None) or torch.equal(a, b)
        # rays without samples keep the reference's buffer initialisation: white unless the mode is black (utils.py:863-898)
        img = one[0][:, :3]
        assert bool((img[one[3].hit_count == 0] == (1.0 if bkgd is not None else 0.0)).all())


def test_frame_job_is_validated(device):
    """qf_frame_render refuses a job whose ray count is not the camera's pixel grid, whose lists are missing, or that asks
    for a field without output arrays -- status -1 (ValueError through _C.check), nothing launched."""
    import ctypes
    from quadraturefields_amd import _C, synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    mesh, mi, field = _scene(device)
    ri = mi.rayintersector
    w, h = 40, 24
    c2w = synthetic.orbit_cameras(1, seed=3)[0]
    focal = synthetic.lego_focal(800) * w / 800.0
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    cam = make_camera(c2w, focal, w, h)

    def fresh():
        ri._raster_backoff = 0
        job, frame, token = ri.fused_frame_job(o, d, 25, cam)
        return job, frame

    job, frame = fresh()
    job.n_rays = w * h - 1
    with pytest.raises(ValueError):
        _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
    job, frame = fresh()
    job.hit_t = None
    with pytest.raises(ValueError):
        _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
    job, frame = fresh()
    desc = field._field_desc(_C.HEAD_NGP, 0)
    job.field = ctypes.addressof(desc)           # a field, but no rgb_c / sigma_c
    job.table, job.base_w = _C.ptr(field.mlp_base.grid_params()).value, _C.ptr(field.mlp_base.network_params()).value
    job.head_ngp_w = _C.ptr(field.mlp_head.params.detach()).value
    with pytest.raises(ValueError):
        _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
    # ... nor one whose field / compositor half is incomplete: no image to write, a custom background without its colour,
    # an unknown background mode, a head that does not match the descriptor.  ALL refused before the first launch
    # (ADVICE r3): the handle's state and the pinned block are untouched.
    out = torch.empty((w * h, 5), device=device)
    rgbs, sig = torch.empty((w * h * 25, 3), device=device), torch.empty((w * h * 25,), device=device)
    bad_desc = field._field_desc(_C.HEAD_NONE, 0)

    def spoil_no_image(j): j.out_packed = None
    def spoil_custom_bg(j): j.bg_mode = _C.BG_CUSTOM
    def spoil_bg_mode(j): j.bg_mode = 7
    def spoil_head_ptr(j): j.head_ngp_w = None
    def spoil_head_kind(j): j.field = ctypes.addressof(bad_desc)

    for spoil in (spoil_no_image, spoil_custom_bg, spoil_bg_mode, spoil_head_ptr, spoil_head_kind):
        job, frame = fresh()
        job.field = ctypes.addressof(desc)
        job.table, job.base_w = _C.ptr(field.mlp_base.grid_params()).value, _C.ptr(field.mlp_base.network_params()).value
        job.head_ngp_w = _C.ptr(field.mlp_head.params.detach()).value
        job.rgb_c, job.sigma_c, job.out_packed = rgbs.data_ptr(), sig.data_ptr(), out.data_ptr()
        job.delta_const, job.bg_mode = 5e-3, _C.BG_WHITE
        spoil(job)
        torch.cuda.synchronize()
        host = ri._frame_scratch(w * h)[2]
        host[:] = -5                                      # the pinned block: any launch of the frame would rewrite it
        frame.tile_base.fill_(-9)
        with pytest.raises((ValueError, _C.QFError)):
            _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
        torch.cuda.synchronize()
        assert host.tolist() == [-5, -5, -5, -5], spoil.__name__
        assert bool((frame.tile_base == -9).all()), spoil.__name__
    torch.cuda.synchronize()
    # sampling only (field = NULL) is a valid job: the tile pack's slot count arrives in total[0]
    job, frame = fresh()
    _C.check(_C.lib().qf_frame_render(ri._handle, ctypes.byref(job), _C.stream()), "qf_frame_render")
    assert int(frame.total_dev.item()) > 100


def test_one_call_frame_with_rays_that_are_not_the_cameras_goes_through_the_bvh(device):
    """qf_frame_render with a camera that does not describe the rays (jittered directions; the wrong focal length;
    ``device="cuda"`` without an index): the pass's device-side ray check routes the WHOLE frame through the
    BVH inside the same bound call, and the pixels are those of the camera-less BVH render, bit for bit -- no host round
    trip, no silently dropped hits, no RecursionError."""
    import warnings
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection, make_camera
    from quadraturefields_amd.render import FrameRenderer
    mesh, _, field = _scene(device)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25, device="cuda")     # index-less
    assert mi.device.index == torch.cuda.current_device() and mi.rayintersector.device.index is not None
    fr = FrameRenderer(mi, field)
    ri = mi.rayintersector
    w, h = 136, 96
    focal = synthetic.lego_focal(800) * w / 800.0
    c2w = synthetic.orbit_cameras(2, seed=5)[1]
    o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
    cam = make_camera(c2w, focal, w, h)
    assert ri.fused_frame_ready(cam, 25)
    good = fr.render_async(o, d, cam)                       # consistent rays: the one-call frame on the fast path
    ref = fr.render(o, d, image_width=w)
    assert torch.equal(good[0], ref[0]) and torch.equal(good[1], ref[1]) and torch.equal(good[2], ref[2])
    ri._settle_fused_policy(0)
    assert ri.camera_mismatch_frames == 0
    g = torch.Generator().manual_seed(1)
    noise = (torch.rand(w * h, 3, generator=g).to(device) - 0.5) * (1.0 / focal)          # ~ +-0.5 px
    dj = d + noise
    dj = (dj / dj.norm(dim=1, keepdim=True)).contiguous()
    for o2, d2, cam2 in ((o, dj, cam), (o, d, make_camera(c2w, focal * 0.9, w, h))):
        ri._raster_backoff = 0
        want = fr.render(o2, d2, image_width=w)             # BVH traversal
        ri._raster_backoff = 0
        assert ri.fused_frame_ready(cam2, 25)
        before = ri.camera_mismatch_frames
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = fr.render_async(o2, d2, cam2)
            n_got = ri.frame_samples()
            ri._settle_fused_policy(0)
        assert ri.camera_mismatch_frames == before + 1
        assert n_got == want[3] > 1000
        for a, b in zip(got[:3], want[:3]):
            assert torch.equal(a, b)
