"""Shared test plumbing: build oracle weight bundles from the product modules' state dicts."""
import numpy as np
import torch

from oracle import fields as ofields


def oracle_ngp_weights(module) -> ofields.NGPWeights:
    """NGPRadianceField / NGPRadianceFieldSGNew (product module) -> oracle.fields.NGPWeights on the CPU."""
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    lv = ofields.grid_levels(module.n_levels, module.log2_hashmap_size, module.base_resolution,
                             module.per_level_scale)
    p = sd["mlp_base.params"].float()
    base, off = ofields.split_tcnn_mlp(p, 32, 1, 64, 16)
    table = p[off:].reshape(-1, 2)
    assert table.shape[0] == lv.n_entries
    w = ofields.NGPWeights(aabb=sd["aabb"].float(), levels=lv, table=table, base=base)
    if "mlp_head.params" in sd:
        w.head_tcnn, _ = ofields.split_tcnn_mlp(sd["mlp_head.params"].float(), 32, 2, 64, 16)
    else:
        w.head_layers = [(sd["mlp_head.layers.0.weight"], sd["mlp_head.layers.0.bias"]),
                         (sd["mlp_head.layers.1.weight"], sd["mlp_head.layers.1.bias"]),
                         (sd["mlp_head.lout.weight"], sd["mlp_head.lout.bias"])]
        w.n_lobes = module.num_g_lobes
    return w


def oracle_deform_weights(module) -> ofields.DeformWeights:
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    g = module.xyz_encoder.grid
    lv = ofields.grid_levels(g.n_levels, g.log2_hashmap_size, g.base_resolution, g.per_level_scale)
    layers = [(sd["decoder_field.layers.0.weight"], sd["decoder_field.layers.0.bias"]),
              (sd["decoder_field.layers.1.weight"], sd["decoder_field.layers.1.bias"]),
              (sd["decoder_field.lout.weight"], sd["decoder_field.lout.bias"])]
    return ofields.DeformWeights(scale=float(module.scale), levels=lv, table=sd["xyz_encoder.params"].reshape(-1, 2),
                                 layers=layers)


def random_points(n, aabb_half=1.5, seed=0, outside_frac=0.05):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, 3, generator=g) * 2 - 1) * aabb_half
    k = int(n * outside_frac)
    if k:
        x[:k] *= 1.2      # some points outside the aabb (selector = 0)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    return x, d


def packed_segments(n_rays, max_per_ray, seed=0, empty_frac=0.3):
    """Random per-ray sample counts (with empty rays) -> sorted ray_indices."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, max_per_ray + 1, size=n_rays)
    counts[rng.random(n_rays) < empty_frac] = 0
    return torch.from_numpy(np.repeat(np.arange(n_rays), counts)).long(), torch.from_numpy(counts)
