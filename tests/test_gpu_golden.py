"""GPU: the reference's own outputs, fed STRAIGHT to the HIP entry points (one hop: fixture -> kernel).

tests/golden/*_ref.npz hold inputs and outputs of the reference's function bodies (tests/golden/gen_from_reference.py
executes ``field_rendering.py``, ``derive_properties``, the ``ngp.py`` quantisers and SG evaluation,
``FeatureCompression.compress / get_features_from_texture_map``, ``sampling_raytrace_numpy``, ``sampling_indexing``,
``generate_splits`` in the build container).  tests/test_golden.py pins the ORACLE to them on the CPU; here the same
arrays go to the device and the kernels' results are compared with the reference's outputs themselves -- no oracle in
the chain.  Bars: integer / uint8 / bool outputs bit exact; fp32 2e-6 + 2e-5|x| (libm, summation order); uint8 codes
produced by device transcendentals (atan2 / acos / log of ``compress`` and ``discretize``) may land one code away from
the host's on a small fraction of elements -- bounded below, stated per test.
"""
import numpy as np
import pytest
import torch

from oracle import fields as ofields
from tests.test_golden import close, load

pytestmark = pytest.mark.gpu


def _dev(z, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in z.items()}


def test_hip_volrend_vs_reference(device):
    from quadraturefields_amd import field_rendering as fr
    z = _dev(load("volrend_ref.npz"), device)
    fn = lambda a, b, r: (z["rgbs"], z["sigmas"])
    c, o, d, ex = fr.rendering(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"], rgb_sigma_fn=fn, render_bkgd=z["bkgd"])
    close(c, z["colors"]); close(o, z["opacities"]); close(d, z["depths"], 1e-5, 1e-4)
    for k in ("weights", "trans", "alphas"):
        close(ex[k], z[k])
    c, o, d, ex = fr.rendering(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"],
                               rgb_alpha_fn=lambda a, b, r: (z["rgbs"], z["alphas_in"]))
    close(c, z["colors_alpha"], 1e-6, 3e-5); close(ex["weights"], z["weights_alpha"], 1e-6, 3e-5)
    f = fr.rendering_field(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"], rgb_sigma_fn=fn)
    for got, key in zip(f, ("field_colors", "field_opacities", "field_depths", "field_weights", "field_weights_rev")):
        close(got, z[key], 1e-5, 1e-4)
    vis = fr.render_visibility_from_density(z["t_starts"], z["t_ends"], z["sigmas"], ray_indices=z["ray_indices"],
                                            n_rays=z["n_rays"], early_stop_eps=0.05, alpha_thre=0.02)
    assert torch.equal(vis, z["visibility"])
    close(fr.accumulate_along_rays(z["weights"], z["rgbs"], z["ray_indices"], z["n_rays"]), z["accumulated"])


def test_hip_derive_properties_vs_reference(device):
    from quadraturefields_amd import spc_render, utils
    z = _dev(load("derive_properties_ref.npz"), device)
    b = spc_render.mark_pack_boundaries(z["index_ray"])
    for bg in ("white", "black", "random"):
        rgb, alpha, hit, dep, w = utils.derive_properties(z["color"], z["density"], z["depth"], z["deltas"], b, z["index_ray"],
                                                          render_bkgd=z["bkgd"], bg_color=bg, N=z["n_rays"])
        assert torch.equal(hit, z[f"hit_{bg}"])
        close(rgb, z[f"rgb_{bg}"], 2e-6, 2e-5); close(alpha, z[f"alpha_{bg}"], 2e-6, 2e-5)
        close(dep, z[f"depth_{bg}"], 2e-6, 2e-5); close(w, z[f"weights_{bg}"], 2e-6, 2e-5)


def test_hip_sg_and_texture_vs_reference(device):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    from quadraturefields_amd.texture_utils import FeatureCompression
    z = load("sg_ref.npz")
    for lobes in (3, 6):
        f = NGPRadianceFieldSGNew(aabb=[-1] * 3 + [1] * 3, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=8).to(device)
        close(f.features_to_rgb(z[f"features_{lobes}"].to(device), z[f"dirs_{lobes}"].to(device)), z[f"rgb_{lobes}"], 2e-6, 2e-5)
        close(f.spherical_gaussian_mixture(z[f"features_{lobes}"][:, 3:].to(device), z[f"dirs_{lobes}"].to(device)),
              ofields.spherical_gaussian_mixture(z[f"features_{lobes}"][:, 3:], z[f"dirs_{lobes}"], lobes), 2e-6, 2e-5)
        # discretize=True: device atan2 / acos / log may land one uint8 code away from the host's on a handful of
        # elements (a code step moves rgb by up to ~0.05), everything else agrees to rounding
        fd = NGPRadianceFieldSGNew(aabb=[-1] * 3 + [1] * 3, use_viewdirs=False, num_g_lobes=lobes, log2_hashmap_size=8,
                                   discretize=True).to(device)
        got = fd.features_to_rgb(z[f"features_{lobes}"].to(device), z[f"dirs_{lobes}"].to(device)).cpu()
        err = (got - z[f"rgb_disc_{lobes}"]).abs().max(dim=1).values
        assert (err > 2e-5).float().mean() < 0.02 and err.max() < 0.1, (float((err > 2e-5).float().mean()), float(err.max()))
    z = load("texture_ref.npz")
    for codec, thres in (("sigmoid", 7.5), ("linear", 5.0)):
        comp = FeatureCompression.from_arrays(z[f"{codec}_alpha"], z[f"{codec}_diffuse"], [z[f"{codec}_colors{i}"] for i in range(3)],
                                              [z[f"{codec}_lambdas{i}"] for i in range(3)], compression_type=codec, lambda_thres=thres)
        close(comp.get_features_from_texture_map(z[f"{codec}_indices"].to(device)), z[f"{codec}_features"], 2e-6, 2e-5)


def test_hip_sampling_vs_reference(device):
    from quadraturefields_amd.mesh_io import TriMesh
    from quadraturefields_amd.mesh_utils import MeshIntersection
    z = load("sampling_ref.npz")
    mi = MeshIntersection(TriMesh(z["vertices"].numpy(), z["faces"].numpy()), simplify_mesh=False, scale=1.0,
                          num_intersections=25, min_hit_separation=0.0)      # the fixture's stand-in intersector returns every hit
    data = mi.sampling_raytrace_device(z["viewdirs"], z["origins"])
    for got, key in zip(data, ("xyzs", "dirs", "index_ray", "ts", "index_tri", "origins_s")):
        assert torch.equal(got.cpu(), z[key]), key
    dev = lambda t: t.to(device)
    out = mi.sampling_indexing(data[0], data[5], data[1], data[2], dev(z["ts_perturbed"]), data[4])
    for got, key in zip(out, ("s_points", "s_deltas", "s_boundary", "s_dirs", "s_index_ray", "s_depth", "s_index_tri", "s_origins")):
        assert torch.equal(got.cpu(), z[key]), key


def test_hip_volrend_pieces_vs_reference(device):
    """The remaining public pieces of field_rendering.py on the fixture: pack_info / exclusive_sum / exclusive_prod
    (through the transmittance helpers), render_weight_from_density, render_visibility_from_alpha,
    accumulate_along_rays_ (in place)."""
    from quadraturefields_amd import field_rendering as fr
    z = _dev(load("volrend_ref.npz"), device)
    w, t, a = fr.render_weight_from_density(z["t_starts"], z["t_ends"], z["sigmas"], ray_indices=z["ray_indices"], n_rays=z["n_rays"])
    close(w, z["weights"]); close(t, z["trans"]); close(a, z["alphas"])
    w2, t2 = fr.render_weight_from_alpha(z["alphas_in"], ray_indices=z["ray_indices"], n_rays=z["n_rays"])
    close(w2, z["weights_alpha"], 1e-6, 3e-5)
    info = fr.pack_info(z["ray_indices"], z["n_rays"])
    counts = torch.bincount(z["ray_indices"], minlength=z["n_rays"])
    assert torch.equal(info[:, 1], counts) and torch.equal(info[:, 0], torch.cumsum(counts, 0) - counts)
    out = torch.zeros((z["n_rays"], 3), device=device)
    fr.accumulate_along_rays_(z["weights"], z["rgbs"], z["ray_indices"], out)
    close(out, z["accumulated"])
    close(fr.accumulate_along_rays(z["weights"], None, z["ray_indices"], z["n_rays"]), z["opacities"])


def test_hip_quantisers_all_256_codes_vs_reference(device):
    """Every uint8 code point through the DEVICE decoders (the texel kernels' code tables): a 256 x 256 texture set
    whose texel (i, j) carries code i / j in every plane, fetched with qf_texture_fetch, against the reference's
    ``inverse_of_*`` outputs for all 256 (colours, lambdas) and all 65 536 (azimuth, elevation) code points."""
    from quadraturefields_amd.texture_utils import FeatureCompression
    z = load("quantisers_ref.npz")
    codes = torch.arange(256, dtype=torch.uint8)
    ci = codes[:, None].expand(256, 256).contiguous()            # code = row
    cj = codes[None, :].expand(256, 256).contiguous()            # code = column
    ii, jj = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    idx = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)).long().to(device)
    for codec, thres, lam_key in (("sigmoid", 7.5, "inv_lambda_75"), ("linear", 5.0, "inv_lambda_50"), ("sigma", 7.5, "inv_lambda_75")):
        comp = FeatureCompression.from_arrays(
            ci, torch.stack([ci, cj, ci], -1), [torch.stack([cj, ci, cj], -1)], [torch.stack([ci, ci, cj], -1)],
            compression_type=codec, lambda_thres=thres, device=device)
        f = comp.get_features_from_texture_map(idx).cpu().reshape(256, 256, -1)      # [diffuse3 | axis3, lambda, colour3 | sigma]
        inv_c = z[f"inv_colors_{codec}"]
        # linear decoders are exact; log / exp / sin / cos of the device differ from the host's libm by an ulp
        eq = (lambda a, b: close(a, b, 2e-6, 2e-6)) if codec == "sigma" else (lambda a, b: close(a, b, 0, 0))
        eq(f[:, 0, 0], inv_c); eq(f[0, :, 1], inv_c); eq(f[:, 0, 2], inv_c)          # diffuse
        eq(f[0, :, 7], inv_c); eq(f[:, 0, 8], inv_c); eq(f[0, :, 9], inv_c)          # lobe colour
        close(f[:, 0, 6], z[lam_key], 1e-7, 2e-6)                                    # lambda (exp)
        # axis from (azimuth = row code, elevation = column code): the fixture enumerates az-major
        close(f[:, :, 3:6].reshape(-1, 3), z["inv_axis"], 3e-7, 0)
    # sigma: the texture decoder clips 1 - a/255 at 1e-6 (texture_utils.py:61-65, Appendix B-9); codes 0..254 equal
    # utils.inverse_of_compressed_sigma, code 255 is finite here and inf there
    close(f[:255, 0, -1], z["inv_sigma_utils"][:255], 1e-6, 2e-6)
    assert torch.isfinite(f[255, 0, -1]) and torch.isinf(z["inv_sigma_utils"][255])


def test_hip_compress_vs_reference(device):
    """FeatureCompression.compress on DEVICE tensors (texture_utils.py:67-90) against the reference's codes.  alpha,
    diffuse, colours and lambdas go through exp / log / clip only: exact.  azimuth / elevation come from device atan2 /
    acos: at most one code away from the host's, on < 2 % of the texels."""
    from quadraturefields_amd.texture_utils import FeatureCompression
    z = load("texture_ref.npz")
    for codec, thres in (("sigmoid", 7.5), ("linear", 5.0)):
        comp = FeatureCompression(3, initialize=True, texture_size=8, compression_type=codec, lambda_thres=thres, device=device)
        data = comp.compress(z[f"{codec}_raw"].to(device))
        d_alpha = (data["alpha"].cpu().int() - z[f"{codec}_c_alpha"].int()).abs()
        d_dif = (data["diffuse"].cpu().int() - z[f"{codec}_c_diffuse"].int()).abs()
        assert int(d_alpha.max()) <= 1 and float((d_alpha > 0).float().mean()) < 0.02
        assert int(d_dif.max()) <= 1 and float((d_dif > 0).float().mean()) < 0.02
        for i in range(3):
            dc = (data["colors"][i].cpu().int() - z[f"{codec}_c_colors{i}"].int()).abs()
            dl = (data["lambdas"][i].cpu().int() - z[f"{codec}_c_lambdas{i}"].int()).abs()
            dl = torch.minimum(dl, 256 - dl)                    # azimuth wraps mod 256 (Appendix B-8)
            assert int(dc.max()) <= 1 and float((dc > 0).float().mean()) < 0.02
            assert int(dl.max()) <= 1 and float((dl > 0).float().mean()) < 0.02


def test_hip_generate_splits_vs_reference(device):
    """generate_splits (train_finetune.py:419-439) on device arrays -- the searchsorted / view route -- reproduces the
    reference's chunking of the fixture."""
    from quadraturefields_amd import utils
    z = load("sampling_ref.npz")
    data = [z[k].to(device) for k in ("xyzs", "dirs")] + [z["split_ids"].to(device)] + \
           [z[k].to(device) for k in ("ts", "index_tri", "origins_s")]
    chunks = utils.generate_splits(data, int(z["split_ids"].max()) + 1)
    assert len(chunks) == z["n_chunks"] and [c[0].shape[0] for c in chunks] == z["chunk_sizes"].tolist()
    assert torch.equal(chunks[0][2].cpu(), z["chunk0_index_ray"]) and torch.equal(chunks[-1][0].cpu(), z["chunk_last_xyzs"])
