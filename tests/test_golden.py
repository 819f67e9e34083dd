"""Golden fixtures produced by the reference's own function bodies (tests/golden/gen_from_reference.py).

CPU (-m "not gpu"): the oracle reproduces the reference's outputs -- this is what pins the oracle.
The HIP path is run on the SAME fixtures, directly (no oracle in between), in tests/test_gpu_golden.py.
Tolerances: integer / uint8 / bool outputs bit exact; fp32 outputs 1e-6 + 1e-5*|x| (libm and summation order).
"""
import os

import numpy as np
import pytest
import torch

from oracle import fields as ofields
from oracle import meshpath as om
from oracle import quantize as oq
from oracle import volrend as ov

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def close(a, b, atol=1e-6, rtol=1e-5):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {err.max().item():.3e}"


# =============================================================================== CPU: oracle vs reference
def test_oracle_volrend_vs_reference():
    z = load("volrend_ref.npz")
    fn = lambda a, b, r: (z["rgbs"], z["sigmas"])
    c, o, d, ex = ov.rendering(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"], rgb_sigma_fn=fn, render_bkgd=z["bkgd"])
    close(c, z["colors"]); close(o, z["opacities"]); close(d, z["depths"], 1e-5, 1e-4)
    for k in ("weights", "trans", "alphas"):
        close(ex[k], z[k])
    c, o, d, ex = ov.rendering(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"],
                               rgb_alpha_fn=lambda a, b, r: (z["rgbs"], z["alphas_in"]))
    close(c, z["colors_alpha"]); close(o, z["opacities_alpha"]); close(ex["weights"], z["weights_alpha"])
    f = ov.rendering_field(z["t_starts"], z["t_ends"], z["ray_indices"], z["n_rays"], rgb_sigma_fn=fn)
    for got, key in zip(f, ("field_colors", "field_opacities", "field_depths", "field_weights", "field_weights_rev")):
        close(got, z[key], 1e-5, 1e-4)
    vis = ov.render_visibility_from_density(z["t_starts"], z["t_ends"], z["sigmas"], ray_indices=z["ray_indices"],
                                            n_rays=z["n_rays"], early_stop_eps=0.05, alpha_thre=0.02)
    assert torch.equal(vis, z["visibility"])
    close(ov.accumulate_along_rays(z["weights"], z["rgbs"], z["ray_indices"], z["n_rays"]), z["accumulated"])


def test_oracle_derive_properties_vs_reference():
    z = load("derive_properties_ref.npz")
    b = ov.mark_pack_boundaries(z["index_ray"])
    for bg in ("white", "black", "random"):
        rgb, alpha, hit, dep, w = ov.derive_properties(z["color"], z["density"], z["depth"], z["deltas"], b, z["index_ray"],
                                                       render_bkgd=z["bkgd"], bg_color=bg, N=z["n_rays"])
        assert torch.equal(hit, z[f"hit_{bg}"])
        close(rgb, z[f"rgb_{bg}"]); close(alpha, z[f"alpha_{bg}"]); close(dep, z[f"depth_{bg}"]); close(w, z[f"weights_{bg}"])


def test_oracle_quantisers_vs_reference():
    z = load("quantisers_ref.npz")
    u = z["codes"]
    for name in ("sigma", "sigmoid", "linear"):
        assert torch.equal(oq.inverse_of_compressed_colors(u, compress_type=name), z[f"inv_colors_{name}"])
    assert torch.equal(oq.inverse_of_compressed_lambda(u, 7.5), z["inv_lambda_75"])
    assert torch.equal(oq.inverse_of_compressed_lambda(u, 5.0), z["inv_lambda_50"])
    az = u[:, None].expand(256, 256).reshape(-1)
    el = u[None, :].expand(256, 256).reshape(-1)
    assert torch.equal(oq.inverse_of_azimuth_and_elevation(az, el), z["inv_axis"])
    assert torch.equal(oq.inverse_of_compressed_sigma_unclipped(u), z["inv_sigma_utils"])
    assert torch.equal(oq.compress_colors(z["x"], compress_type="sigma"), z["comp_colors_sigma"])
    assert torch.equal(oq.compress_colors(z["x"], compress_type="sigmoid"), z["comp_colors_sigmoid"])
    assert torch.equal(oq.compress_lambda(z["lam"], 7.5), z["comp_lambda"])
    a, e = oq.compress_polar_coordinates(z["x"])
    assert torch.equal(a, z["comp_azimuth"]) and torch.equal(e, z["comp_elevation"])
    assert torch.equal(oq.compress_sigma(z["sigma"]), z["comp_sigma"])


def test_oracle_sg_vs_reference():
    z = load("sg_ref.npz")
    for lobes in (3, 6):
        close(ofields.features_to_rgb(z[f"features_{lobes}"], z[f"dirs_{lobes}"], lobes), z[f"rgb_{lobes}"], 1e-7, 1e-6)
        # discretize=True (ngp.py:377-382,458-459): same torch ops on the same machine -> identical code points
        assert torch.equal(ofields.features_to_rgb(z[f"features_{lobes}"], z[f"dirs_{lobes}"], lobes, discretize=True),
                           z[f"rgb_disc_{lobes}"])
        assert torch.equal(ofields.spherical_gaussian_mixture(z[f"features_{lobes}"][:, 3:], z[f"dirs_{lobes}"], lobes,
                                                              discretize=True), z[f"mixture_disc_{lobes}"])


def test_oracle_texture_vs_reference():
    z = load("texture_ref.npz")
    for codec, thres in (("sigmoid", 7.5), ("linear", 5.0)):
        colors = [z[f"{codec}_colors{i}"] for i in range(3)]
        lambdas = [z[f"{codec}_lambdas{i}"] for i in range(3)]
        got = oq.features_from_texture_map(z[f"{codec}_indices"], z[f"{codec}_alpha"], z[f"{codec}_diffuse"], colors,
                                           lambdas, codec, thres)
        assert torch.equal(got, z[f"{codec}_features"])
        data = oq.compress_features(z[f"{codec}_raw"], 3, codec, thres)
        assert torch.equal(data["alpha"], z[f"{codec}_c_alpha"]) and torch.equal(data["diffuse"], z[f"{codec}_c_diffuse"])
        for i in range(3):
            assert torch.equal(data["colors"][i], z[f"{codec}_c_colors{i}"])
            assert torch.equal(data["lambdas"][i], z[f"{codec}_c_lambdas{i}"])


def test_oracle_sampling_vs_reference():
    z = load("sampling_ref.npz")
    bf = om.BruteForceIntersector(z["vertices"].numpy(), z["faces"].numpy(), min_separation=0.0)    # as the generator
    s = om.sampling_raytrace_numpy(bf, z["viewdirs"].numpy(), z["origins"].numpy(), 25)
    data = om.to_loader_tensors(s)
    for got, key in zip(data, ("xyzs", "dirs", "index_ray", "ts", "index_tri", "origins_s")):
        assert torch.equal(got, z[key]), key
    assert np.array_equal(s[3], z["depth64"].numpy())
    out = om.sampling_indexing(data[0], data[5], data[1], data[2], z["ts_perturbed"], data[4])
    for got, key in zip(out, ("s_points", "s_deltas", "s_boundary", "s_dirs", "s_index_ray", "s_depth", "s_index_tri", "s_origins")):
        assert torch.equal(got, z[key]), key
    chunks = om.generate_splits((data[0], data[1], z["split_ids"], data[3], data[4], data[5]), int(z["split_ids"].max()) + 1)
    assert len(chunks) == z["n_chunks"] and [c[0].shape[0] for c in chunks] == z["chunk_sizes"].tolist()
    assert torch.equal(chunks[0][2], z["chunk0_index_ray"]) and torch.equal(chunks[-1][0], z["chunk_last_xyzs"])


def test_oracle_and_product_ray_generation_vs_reference():
    from quadraturefields_amd import synthetic
    z = load("rays_ref.npz")
    for fn in (om.generate_rays, synthetic.camera_rays):
        o, d = fn(z["c2w"], z["focal"], z["width"], z["height"])
        assert torch.equal(o, z["origins"]) and torch.equal(d, z["viewdirs"])


def test_training_ray_batches_vs_reference():
    """SubjectLoader.fetch_data + preprocess in training mode (random pixels of random images, pixel jitter,
    background augmentation) against the reference's own bodies under the same seeded CPU generator: the loader makes
    the same draws in the same order, so the batches are identical.  (Host tensors here are the reference's own
    configuration -- its loaders live on the CPU, train_finetune.py:301 -- not a fallback of a device kernel.)"""
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    z = load("train_rays_ref.npz")
    for tag, over, noise, up, aug in (("a", True, False, 1, "white"), ("b", False, True, 2, "random"), ("c", True, True, 2, "black")):
        ds = SubjectLoader.from_arrays(z["images"].numpy(), z["c2w_all"].numpy(), float(z["focal"]), split="train",
                                       num_rays=int(z["num_rays"]), batch_over_images=over, add_ray_direction_noise=noise,
                                       upsample=up, color_bkgd_aug=aug, device="cpu")
        assert ds.training and len(ds) == 10000000 and len(ds.images) == 3
        torch.manual_seed(int(z["seed"]))
        item = ds[2]
        assert torch.equal(item["rays"].origins, z[f"{tag}_origins"])
        assert torch.equal(item["rays"].viewdirs, z[f"{tag}_viewdirs"])
        assert torch.equal(item["pixels"], z[f"{tag}_pixels"]) and torch.equal(item["color_bkgd"], z[f"{tag}_color_bkgd"])
    ds.update_num_rays(7)
    assert ds[0]["rays"].origins.shape == (7, 3)
    # a split outside train / trainval never trains, whatever num_rays says (the scripts' "whole" loader)
    assert not SubjectLoader.from_arrays(z["images"].numpy(), z["c2w_all"].numpy(), 13.7, split="whole", num_rays=64,
                                         device="cpu").training


def test_product_host_quantisers_vs_reference():
    """The product's torch-level codecs (encode side of the texture format) against the reference outputs."""
    from quadraturefields_amd.radiance_fields import ngp
    from quadraturefields_amd import utils
    z = load("quantisers_ref.npz")
    u = z["codes"]
    assert torch.equal(ngp.inverse_of_compressed_colors(u, compress_type="sigmoid"), z["inv_colors_sigmoid"])
    assert torch.equal(ngp.torch_invserse_of_compressed_lambda(u, 7.5), z["inv_lambda_75"])
    assert torch.equal(ngp.compress_colors(z["x"], compress_type="sigmoid"), z["comp_colors_sigmoid"])
    assert torch.equal(ngp.compress_lambda_torch(z["lam"], 7.5), z["comp_lambda"])
    a, e = ngp.compress_polar_coordinates_torch(z["x"])
    assert torch.equal(a, z["comp_azimuth"]) and torch.equal(e, z["comp_elevation"])
    assert torch.equal(utils.compress_sigma(z["sigma"]), z["comp_sigma"])
    assert torch.equal(utils.inverse_of_compressed_sigma(u), z["inv_sigma_utils"])
    assert torch.equal(ngp.continuous_color(u), z["continuous_color"]) and torch.equal(ngp.continuous_axis(u), z["continuous_axis"])
    assert torch.equal(ngp.discretize_color(z["x"]), z["discretize_color"])
    assert torch.equal(ngp.discretize_axis(torch.tanh(z["x"])), z["discretize_axis"])
