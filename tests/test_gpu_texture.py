"""GPU parity: baked SG texture path vs the CPU oracle.

Bar: texel indices (int64) bit-exact; decoded features / rgb / sigma within 2e-6 + 2e-5*|x| (libm differences in
log / exp / sin / cos only).  All 256 code points of every codec are exercised.
"""
import numpy as np
import pytest
import torch

from oracle import fields as ofields
from oracle import quantize as oq

pytestmark = pytest.mark.gpu


def _close(a, b, atol=2e-6, rtol=2e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {err.max().item():.3e}"


@pytest.mark.parametrize("codec,lobes,thres", [("sigmoid", 6, 7.5), ("linear", 3, 5.0), ("sigma", 2, 7.5)])
def test_all_code_points(device, codec, lobes, thres):
    from quadraturefields_amd.texture_utils import FeatureCompression
    t = 256
    # texel (r, c): every channel walks all 256 codes along the columns, rows permute the pairing
    base = np.arange(t, dtype=np.uint8)[None, :].repeat(t, 0)
    rows = np.arange(t, dtype=np.uint8)[:, None].repeat(t, 1)
    alpha = base.copy()
    diffuse = np.stack([base, rows, base ^ rows], -1).astype(np.uint8)
    colors = [np.stack([base + np.uint8(i), rows, base], -1).astype(np.uint8) for i in range(lobes)]
    lambdas = [np.stack([base, rows, (base + rows).astype(np.uint8)], -1).astype(np.uint8) for i in range(lobes)]
    comp = FeatureCompression.from_arrays(alpha, diffuse, colors, lambdas, compression_type=codec, lambda_thres=thres)
    ii, jj = np.meshgrid(np.arange(t), np.arange(t), indexing="ij")
    idx = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)).long()
    want = oq.features_from_texture_map(idx, torch.from_numpy(alpha), torch.from_numpy(diffuse),
                                        [torch.from_numpy(c) for c in colors], [torch.from_numpy(l) for l in lambdas],
                                        codec, thres)
    got = comp.get_features_from_texture_map(idx.to(device))
    assert got.shape == want.shape == (t * t, 3 + 7 * lobes + 1)
    finite = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got.cpu()), finite)      # sigmoid codec: code 255 -> log(inf) clipped, 0 -> log(1e-8)
    _close(got.cpu()[finite], want[finite])
    # fused shade == decode + features_to_rgb
    d = torch.randn(t * t, 3, generator=torch.Generator().manual_seed(17))
    d = d / d.norm(dim=-1, keepdim=True)
    rgb, sigma = comp.shade(idx.to(device), d.to(device))
    _close(sigma, want[:, -1])
    # lambda reaches e^5 = 148 and colours +-12: a 1-ulp difference in (axis . d) moves the exponent by ~2e-5 and the
    # pre-sigmoid sum by up to ~1e-3 relative, hence the wider bound on rgb (sigmoid slope <= 1/4)
    _close(rgb, ofields.features_to_rgb(want[:, :-1], d, lobes), atol=5e-5, rtol=1e-4)


def test_texel_indices_bit_exact(device):
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from oracle import meshpath as om
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    size = 512
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    c2w = synthetic.orbit_cameras(1, seed=3)[0]
    o, d = synthetic.camera_rays(c2w, synthetic.lego_focal(96), 96, 96)
    data = mi.sampling_raytrace_device(d, o, image_width=96)
    xyz, _, _, _, index_tri, _ = data
    uv = synthetic.scaled_uv(mesh, size)
    f = mesh.faces[index_tri.cpu().numpy()]
    want = oq.texel_indices(mesh.vertices[f], xyz.cpu().numpy(), torch.from_numpy(uv)[torch.from_numpy(f)], size)
    got = utils.texel_indices(mi, uv, xyz, index_tri, size)
    assert got.dtype == torch.int64
    assert torch.equal(got.cpu(), want)
    assert int(got.min()) >= 0 and int(got.max()) <= size - 1
    # points far off the triangle: barycentrics clamp + renormalise (utils.py:1058-1061)
    far = xyz + 0.5
    want = oq.texel_indices(mesh.vertices[f], far.cpu().numpy(), torch.from_numpy(uv)[torch.from_numpy(f)], size)
    assert torch.equal(utils.texel_indices(mi, uv, far, index_tri, size).cpu(), want)
    # the per-triangle record table (default) and the per-sample faces -> vertices -> uv walk give the same texels; the
    # table follows the uv it was built from (cached per uv tensor, rebuilt when that tensor is modified in place)
    uv_d = torch.from_numpy(uv).to(device)
    a = utils.texel_indices(mi, uv_d, xyz, index_tri, size, packed=True)
    b = utils.texel_indices(mi, uv_d, xyz, index_tri, size, packed=False)
    assert torch.equal(a, b) and torch.equal(a, got)
    # the fused lookup + shading launch of the frame path = the two calls
    from quadraturefields_amd.texture_utils import FeatureCompression
    tex = synthetic.random_textures(size, 3, seed=2)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="linear", lambda_thres=5.0, device=device)
    dirs = torch.nn.functional.normalize(torch.randn(xyz.shape[0], 3, generator=torch.Generator().manual_seed(1)), dim=-1).to(device)
    rgb2, sig2 = comp.shade(a, dirs)
    rgb1, sig1 = utils.shade_baked_points(mi, uv_d, comp, xyz, index_tri, dirs)
    assert torch.equal(rgb1, rgb2) and torch.equal(sig1, sig2)
    uv_d.mul_(0.5)
    c = utils.texel_indices(mi, uv_d, xyz, index_tri, size, packed=True)
    assert torch.equal(c, utils.texel_indices(mi, uv_d, xyz, index_tri, size, packed=False)) and not torch.equal(c, a)


def test_compress_roundtrip_through_textures(device):
    """encode (reference's compress) -> device decode: decoded values are within one quantisation step."""
    from quadraturefields_amd.texture_utils import FeatureCompression
    lobes, t = 3, 64
    comp = FeatureCompression(lobes, initialize=True, texture_size=t, compression_type="linear", lambda_thres=7.5)
    g = torch.Generator().manual_seed(0)
    n = t * t
    feats = torch.randn(n, 3 + 7 * lobes + 1, generator=g) * 2
    feats[:, -1] = torch.rand(n, generator=g) * 200
    ii, jj = np.meshgrid(np.arange(t), np.arange(t), indexing="ij")
    idx = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)).long().to(device)
    comp.load_features_into_maps(feats.to(device), idx)
    back = comp.get_features_from_texture_map(idx).cpu()
    assert torch.allclose(back[:, :3], feats[:, :3].clamp(-12, 12), atol=24 / 255 + 1e-4)
    a = 1 - torch.exp(-feats[:, -1] * 0.005)
    assert torch.allclose(1 - torch.exp(-back[:, -1] * 0.005), a, atol=1 / 255 + 1e-4)
    data = oq.compress_features(feats, lobes, "linear", 7.5)
    assert torch.equal(comp.alpha.cpu().reshape(-1), data["alpha"])
    assert torch.equal(comp.diffuse.cpu().reshape(-1, 3), data["diffuse"])
    # compress_features_and_save (texture_utils.py:108-117) writes the same PNG set as filling the maps and saving them
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        comp.save_to_file(td + "/a_")
        comp.compress_features_and_save(feats.to(device).reshape(t, t, -1), td + "/b_")
        a = FeatureCompression(lobes, path=td + "/a_", compression_type="linear")
        b = FeatureCompression(lobes, path=td + "/b_", compression_type="linear")
        assert torch.equal(a.alpha, b.alpha) and torch.equal(a.diffuse, b.diffuse) and torch.equal(a.alpha, comp.alpha)
        for i in range(lobes):
            assert torch.equal(a.sg_colors[i], b.sg_colors[i]) and torch.equal(a.lambdas[i], b.lambdas[i])


@pytest.mark.gpu
@pytest.mark.parametrize("lobes", [1, 3, 6, 8])
def test_packed_texel_records_equal_planes(device, lobes):
    """The interleaved 64-byte texel records (one sector per sample) shade to the same bits as the reference's planes,
    and are rebuilt when a plane changes."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.texture_utils import FeatureCompression
    size = 96
    tex = synthetic.random_textures(size, lobes, seed=lobes)
    comp = FeatureCompression.from_arrays(tex["alpha"], tex["diffuse"], tex["colors"], tex["lambdas"],
                                          compression_type="linear", lambda_thres=5.0, device=device)
    g = torch.Generator().manual_seed(0)
    n = 5000
    idx = torch.randint(0, size, (n, 2), generator=g).to(device)
    idx[0] = torch.tensor([0, 0])
    idx[1] = torch.tensor([size - 1, size - 1])
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(device)
    rgb_p, sig_p = comp.shade(idx, d, packed=True)
    rgb_r, sig_r = comp.shade(idx, d, packed=False)
    assert torch.equal(rgb_p, rgb_r) and torch.equal(sig_p, sig_r)
    rec = comp.records()
    assert rec.shape == (size * size, 64) and rec.dtype == torch.uint8
    assert torch.equal(rec[:, 0].reshape(size, size), comp.alpha)
    assert torch.equal(rec[:, 1:4].reshape(size, size, 3), comp.diffuse)
    assert torch.equal(rec[:, 4:7].reshape(size, size, 3), comp.lambdas[0])
    assert torch.equal(rec[:, 7:10].reshape(size, size, 3), comp.sg_colors[0])
    assert int(rec[:, 4 + 6 * lobes:].max()) == 0
    # two-call form (texture_utils.py:144-147 on the fetched rows) = the fused shade, to rounding
    feats = comp.get_features_from_texture_map(idx)
    assert (comp.features_to_rgb(feats[:, :-1], d) - rgb_p).abs().max() <= 2e-6
    assert (comp.features_to_rgb(feats, d) - rgb_p).abs().max() <= 2e-6
    mix = comp.spherical_gaussian_mixture(feats[:, 3:-1], d)
    assert (torch.sigmoid(feats[:, :3] + mix) - rgb_p).abs().max() <= 2e-6
    comp.alpha[idx[:, 0], idx[:, 1]] = 77                 # in-place edit: the records follow
    rgb2, sig2 = comp.shade(idx, d)
    assert torch.equal(sig2, comp.shade(idx, d, packed=False)[1]) and not torch.equal(sig2, sig_p)
