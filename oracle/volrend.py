"""Oracle: packed volume-rendering arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, on torch-CPU:

* nerfacc 0.5.3 ``pack_info`` / ``exclusive_sum`` / ``exclusive_prod`` semantics
  (un-vendored) as used by
  ``examples/field_rendering.py:201-203,257-261``;
* the nerfacc ``volrend`` family vendored at ``examples/field_rendering.py:14-573``
  and the reference's own ``rendering_field`` (``field_rendering.py:575-733``);
* kaolin 0.14 ``render.spc`` ``mark_pack_boundaries`` / ``cumsum`` / ``sum_reduce`` /
  ``exponential_integration`` (un-vendored; Appendix A.4) as called at
  ``examples/utils.py:869-879`` and ``examples/mesh_utils.py:407``;
* ``derive_properties`` (``examples/utils.py:863-898``) including its background
  quirks (Appendix B-1, B-2).

Segmented scans are evaluated in float64 and rounded once to the input dtype, so the
oracle is the correctly-rounded answer; device kernels accumulate sequentially in
fp32 and are compared at a stated tolerance.
"""
from typing import Callable, Optional

import torch
from torch import Tensor


# --------------------------------------------------------------------------- nerfacc
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """(start, count) per ray for samples grouped by ray.  nerfacc.pack.pack_info."""
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    counts = torch.zeros(n_rays, dtype=ray_indices.dtype)
    counts.index_add_(0, ray_indices, torch.ones_like(ray_indices))
    starts = counts.cumsum(0) - counts
    return torch.stack([starts, counts], dim=-1)


def _segment_ids(packed_info: Tensor, n: int) -> Tensor:
    counts = packed_info[:, 1].long()
    ids = torch.repeat_interleave(torch.arange(len(counts)), counts)
    assert ids.numel() == n, "packed_info does not cover the flattened input"
    return ids


def exclusive_sum(x: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """Per-chunk exclusive prefix sum (first element of every chunk is 0)."""
    if packed_info is None:
        shifted = torch.cat([torch.zeros_like(x[..., :1]), x[..., :-1]], dim=-1)
        return shifted.double().cumsum(-1).to(x.dtype)
    if x.numel() == 0:
        return x.clone()
    seg = _segment_ids(packed_info, x.shape[0])
    incl = x.double().cumsum(0)
    excl = incl - x.double()
    starts = packed_info[:, 0].long()
    base = excl[starts.clamp(max=x.shape[0] - 1)]  # value of the running sum where each chunk starts
    return (excl - base[seg]).to(x.dtype)


def exclusive_prod(x: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    """Per-chunk exclusive prefix product (first element of every chunk is 1)."""
    if packed_info is None:
        shifted = torch.cat([torch.ones_like(x[..., :1]), x[..., :-1]], dim=-1)
        return shifted.double().cumprod(-1).to(x.dtype)
    out = torch.empty_like(x)
    xd = x.double()
    for start, count in packed_info.tolist():
        if count == 0:
            continue
        seg = xd[start:start + count]
        shifted = torch.cat([torch.ones(1, dtype=torch.float64), seg[:-1]])
        out[start:start + count] = shifted.cumprod(0).to(x.dtype)
    return out


def render_transmittance_from_alpha(alphas, packed_info=None, ray_indices=None,
                                    n_rays=None, prefix_trans=None):
    """T_i = prod_{j<i} (1 - alpha_j).  field_rendering.py:161-206."""
    if ray_indices is not None and packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    trans = exclusive_prod(1 - alphas, packed_info)
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=None,
                                      ray_indices=None, n_rays=None, prefix_trans=None):
    """T_i = exp(-sum_{j<i} sigma_j dt_j), alpha_i = 1 - exp(-sigma_i dt_i).  :209-264."""
    if ray_indices is not None and packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, packed_info))
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans, alphas


def render_weight_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None,
                             prefix_trans=None):
    """w_i = T_i alpha_i.  field_rendering.py:267-309."""
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None,
                               ray_indices=None, n_rays=None, prefix_trans=None):
    """w_i = T_i (1 - exp(-sigma_i dt_i)).  field_rendering.py:312-362."""
    trans, alphas = render_transmittance_from_density(
        t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans, alphas


def render_visibility_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None,
                                 early_stop_eps=1e-4, alpha_thre=0.0, prefix_trans=None):
    """field_rendering.py:365-418."""
    trans = render_transmittance_from_alpha(alphas, packed_info, ray_indices, n_rays, prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info=None,
                                   ray_indices=None, n_rays=None, early_stop_eps=1e-4,
                                   alpha_thre=0.0, prefix_trans=None):
    """field_rendering.py:421-480."""
    trans, alphas = render_transmittance_from_density(
        t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays, prefix_trans)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    """sum_i w_i v_i per ray.  field_rendering.py:483-547."""
    if values is None:
        src = weights[..., None]
    else:
        assert values.dim() == weights.dim() + 1
        assert weights.shape == values.shape[:-1]
        src = weights[..., None] * values
    if ray_indices is None:
        return src.sum(dim=-2)
    assert n_rays is not None, "n_rays must be provided"
    assert weights.dim() == 1, "weights must be flattened"
    out = torch.zeros((n_rays, src.shape[-1]), dtype=torch.float64)
    out.index_add_(0, ray_indices, src.double())
    return out.to(src.dtype)


def rendering(t_starts, t_ends, ray_indices=None, n_rays=None,
              rgb_sigma_fn: Optional[Callable] = None,
              rgb_alpha_fn: Optional[Callable] = None,
              render_bkgd: Optional[Tensor] = None):
    """nerfacc-style packed rendering.  field_rendering.py:14-158."""
    if ray_indices is not None:
        assert t_starts.shape == t_ends.shape == ray_indices.shape
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")
    empty = t_starts.shape[0] == 0
    if rgb_sigma_fn is not None:
        if empty:
            rgbs, sigmas = torch.empty((0, 3)), torch.empty((0,))
        else:
            rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
        assert rgbs.shape[-1] == 3
        assert sigmas.shape == t_starts.shape
        weights, trans, alphas = render_weight_from_density(
            t_starts, t_ends, sigmas, ray_indices=ray_indices, n_rays=n_rays)
        extras = dict(weights=weights, alphas=alphas, trans=trans, sigmas=sigmas, rgbs=rgbs)
    else:
        if empty:
            rgbs, alphas = torch.empty((0, 3)), torch.empty((0,))
        else:
            rgbs, alphas = rgb_alpha_fn(t_starts, t_ends, ray_indices)
        assert rgbs.shape[-1] == 3
        assert alphas.shape == t_starts.shape
        weights, trans = render_weight_from_alpha(alphas, ray_indices=ray_indices, n_rays=n_rays)
        extras = dict(weights=weights, trans=trans, rgbs=rgbs, alphas=alphas)
    colors = accumulate_along_rays(weights, rgbs, ray_indices, n_rays)
    opacities = accumulate_along_rays(weights, None, ray_indices, n_rays)
    mids = (t_starts + t_ends)[..., None] / 2.0
    depths = accumulate_along_rays(weights, mids, ray_indices, n_rays)
    depths = depths / opacities.clamp_min(torch.finfo(rgbs.dtype).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opacities)
    return colors, opacities, depths, extras


def rendering_field(t_starts, t_ends, ray_indices=None, n_rays=None,
                    rgb_sigma_fn: Optional[Callable] = None,
                    render_bkgd: Optional[Tensor] = None):
    """``rendering`` + the reversed-ray weights.  field_rendering.py:575-733 (density branch).

    Quirks kept (:719-731): the mirrored (t_ends, t_starts) are passed swapped so dt stays
    positive; samples and ray ids are flipped, but ``pack_info`` still lays its chunks out
    in ascending ray-id order (start = cumsum(count) - count), so on the flipped
    (descending) data the chunks only coincide with rays when the per-ray counts are
    palindromic.  The reference inherits this from nerfacc's chunk-based scan; we restate
    it as is.  Only stage 2 (train_field.py) consumes the result.
    """
    colors, opacities, depths, extras = rendering(
        t_starts, t_ends, ray_indices, n_rays, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=render_bkgd)
    weights, sigmas = extras["weights"], extras["sigmas"]
    max_val = t_starts.max() + t_ends.max()
    ts_r = torch.flip(max_val - t_starts, dims=[0])
    te_r = torch.flip(max_val - t_ends, dims=[0])
    w_rev, _, _ = render_weight_from_density(
        te_r, ts_r, torch.flip(sigmas, dims=[0]),
        ray_indices=torch.flip(ray_indices, dims=[0]), n_rays=n_rays)
    return colors, opacities, depths, weights, torch.flip(w_rev, dims=[0])


# ---------------------------------------------------------------------------- kaolin
def mark_pack_boundaries(ridx: Tensor) -> Tensor:
    """True at i == 0 and wherever ridx[i] != ridx[i-1].  kaolin render.spc (A.4)."""
    b = torch.ones(ridx.shape[0], dtype=torch.bool)
    if ridx.shape[0] > 1:
        b[1:] = ridx[1:] != ridx[:-1]
    return b


def _boundary_segments(boundary: Tensor) -> Tensor:
    return boundary.long().cumsum(0) - 1


def seg_cumsum(feats: Tensor, boundary: Tensor, exclusive: bool = False) -> Tensor:
    """Segmented cumulative sum restarting at each boundary.  kaolin ``cumsum``."""
    if feats.shape[0] == 0:
        return feats.clone()
    seg = _boundary_segments(boundary)
    f = feats.double()
    incl = f.cumsum(0)
    starts = torch.nonzero(boundary).flatten()
    base = (incl - f)[starts]
    incl = incl - base[seg]
    return (incl - f if exclusive else incl).to(feats.dtype)


def sum_reduce(feats: Tensor, boundary: Tensor) -> Tensor:
    """Per-segment sum, one row per segment in segment order.  kaolin ``sum_reduce``."""
    nseg = int(boundary.sum().item())
    out = torch.zeros((nseg,) + tuple(feats.shape[1:]), dtype=torch.float64)
    if feats.shape[0]:
        out.index_add_(0, _boundary_segments(boundary), feats.double())
    return out.to(feats.dtype)


def exponential_integration(feats: Tensor, tau: Tensor, boundary: Tensor, exclusive: bool = True):
    """kaolin ``exponential_integration`` (A.4): returns (sum_seg w*feats, w) with
    w = exp(-cumsum(tau)) * (1 - exp(-tau)).  NB the second value is the weight, not T."""
    alpha = 1.0 - torch.exp(-tau)
    trans = torch.exp(-seg_cumsum(tau, boundary, exclusive=exclusive))
    w = trans * alpha
    return sum_reduce(w * feats, boundary), w


def derive_properties(color, density, depths, deltas, boundary, index_ray,
                      render_bkgd=None, bg_color="white", N=0):
    """Per-ray colour / alpha / depth from packed samples.  examples/utils.py:863-898.

    Returns (rgb[N,3], alpha[N,1], hit_ray_ids, depth[N,1], weights[S,1]).
    Background quirks reproduced: the already-weighted colour sum is multiplied by alpha
    again (B-1); rays without samples are white unless bg_color == "black" (B-2).
    """
    color = color.reshape(-1, 3).contiguous()
    tau = (density * deltas).reshape(-1, 1)
    ray_colors, weights = exponential_integration(color, tau, boundary, exclusive=True)
    ray_depth, _ = exponential_integration(depths.reshape(-1, 1), tau, boundary, exclusive=True)
    alpha = sum_reduce(weights, boundary)
    out_alpha = torch.zeros(N, 1, dtype=torch.float32)
    out_depth = torch.zeros(N, 1, dtype=torch.float32)
    if bg_color == "white":
        rgb = torch.ones(N, 3, dtype=torch.float32)
        blended = (1.0 - alpha) + alpha * ray_colors
    elif bg_color == "black":
        rgb = torch.zeros(N, 3, dtype=torch.float32)
        blended = alpha * ray_colors
    else:
        rgb = torch.ones(N, 3, dtype=torch.float32)
        blended = alpha * ray_colors + (1.0 - alpha) * render_bkgd
    hit = index_ray[boundary]
    out_depth[hit] = ray_depth.float()
    rgb[hit] = blended.float()
    out_alpha[hit] = alpha.float()
    return rgb, out_alpha, hit, out_depth, weights
