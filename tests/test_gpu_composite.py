"""GPU parity: packed compositing kernels vs the CPU oracle, incl. the reference's docstring vectors.

Tolerance: fp32 sequential scans on the device vs float64-then-rounded scans in the oracle:
|a-b| <= 1e-6 + 1e-5*|b| (<= 25 samples per ray; up to 400 in the long-ray cases).
"""
import numpy as np
import pytest
import torch

from oracle import volrend as ov
from tests import helpers

pytestmark = pytest.mark.gpu


def _close(a, b, atol=1e-6, rtol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {err.max().item():.3e}"


def test_docstring_vectors_on_device(device):
    """field_rendering.py:192-195,246-253,298-302,347-355,403-409,461-471."""
    from quadraturefields_amd import field_rendering as fr
    alphas = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], device=device)
    ridx = torch.tensor([0, 0, 0, 1, 1, 2, 2], device=device)
    t = fr.render_transmittance_from_alpha(alphas, ray_indices=ridx, n_rays=3)
    _close(t, torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0]))
    w, t = fr.render_weight_from_alpha(alphas, ray_indices=ridx, n_rays=3)
    _close(w, torch.tensor([0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9]))
    ts = torch.arange(7.0, device=device)
    te = ts + 1
    sig = alphas.clone()
    w, t, a = fr.render_weight_from_density(ts, te, sig, ray_indices=ridx, n_rays=3)
    assert torch.allclose(t.cpu(), torch.tensor([1.00, 0.67, 0.30, 1.00, 0.45, 1.00, 1.00]), atol=5e-3)
    assert torch.allclose(a.cpu(), torch.tensor([0.33, 0.55, 0.095, 0.55, 0.095, 0.00, 0.59]), atol=5e-3)
    assert torch.allclose(w.cpu(), torch.tensor([0.33, 0.37, 0.03, 0.55, 0.04, 0.00, 0.59]), atol=6e-3)
    v = fr.render_visibility_from_alpha(alphas, ray_indices=ridx, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2)
    assert v.cpu().tolist() == [True, True, False, True, False, False, True]
    v = fr.render_visibility_from_density(ts, te, sig, ray_indices=ridx, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2)
    assert v.cpu().tolist() == [True, True, False, True, False, False, True]


@pytest.mark.parametrize("n_rays,max_per", [(1, 1), (7, 3), (1000, 25), (50, 400)])
def test_pack_scan_accumulate(device, n_rays, max_per):
    from quadraturefields_amd import field_rendering as fr
    ridx, counts = helpers.packed_segments(n_rays, max_per, seed=n_rays)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(1)
    x = torch.rand(n, generator=g)
    info_o = ov.pack_info(ridx, n_rays)
    info = fr.pack_info(ridx.to(device), n_rays)
    assert torch.equal(info.cpu(), info_o)                       # integer: bit exact
    _close(fr.exclusive_sum(x.to(device), info), ov.exclusive_sum(x, info_o))
    _close(fr.exclusive_prod(0.2 + x.to(device), info), ov.exclusive_prod(0.2 + x, info_o), rtol=2e-5)
    vals = torch.rand(n, 4, generator=g)
    _close(fr.accumulate_along_rays(x.to(device), vals.to(device), ridx.to(device), n_rays),
           ov.accumulate_along_rays(x, vals, ridx, n_rays))
    _close(fr.accumulate_along_rays(x.to(device), None, ridx.to(device), n_rays),
           ov.accumulate_along_rays(x, None, ridx, n_rays))
    # batched form
    xb = torch.rand(5, 9, generator=g)
    _close(fr.exclusive_sum(xb.to(device)), ov.exclusive_sum(xb))
    _close(fr.accumulate_along_rays(xb.to(device), torch.ones(5, 9, 2, device=device)),
           ov.accumulate_along_rays(xb, torch.ones(5, 9, 2)))


@pytest.mark.parametrize("bkgd", [None, [0.2, 0.5, 0.9]])
def test_rendering_matches_oracle(device, bkgd):
    from quadraturefields_amd import field_rendering as fr
    ridx, _ = helpers.packed_segments(500, 30, seed=5)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(2)
    ts = torch.rand(n, generator=g)
    te = ts + 0.01 + 0.05 * torch.rand(n, generator=g)
    rgbs = torch.rand(n, 3, generator=g)
    sig = torch.rand(n, generator=g) * 60
    sig[::7] = 0.0
    bk = None if bkgd is None else torch.tensor(bkgd)
    c_o, o_o, d_o, ex_o = ov.rendering(ts, te, ridx, 500, rgb_sigma_fn=lambda a, b, c: (rgbs, sig), render_bkgd=bk)
    dev = lambda t: None if t is None else t.to(device)
    c, o, d, ex = fr.rendering(dev(ts), dev(te), dev(ridx), 500, rgb_sigma_fn=lambda a, b, c: (dev(rgbs), dev(sig)),
                               render_bkgd=dev(bk))
    assert c.shape == (500, 3) and o.shape == (500, 1) and d.shape == (500, 1)
    _close(c, c_o)
    _close(o, o_o)
    _close(d, d_o, atol=1e-5, rtol=1e-4)          # depth is a quotient of two sums
    for k in ("weights", "trans", "alphas"):
        _close(ex[k], ex_o[k])
    # alpha branch
    al = torch.rand(n, generator=g)
    c_o, o_o, d_o, _ = ov.rendering(ts, te, ridx, 500, rgb_alpha_fn=lambda a, b, c: (rgbs, al))
    c, o, d, _ = fr.rendering(dev(ts), dev(te), dev(ridx), 500, rgb_alpha_fn=lambda a, b, c: (dev(rgbs), dev(al)))
    _close(c, c_o, rtol=3e-5)
    _close(o, o_o, rtol=3e-5)
    with pytest.raises(ValueError):
        fr.rendering(dev(ts), dev(te), dev(ridx), 500)
    # empty input
    e = torch.zeros(0, device=device)
    c, o, d, _ = fr.rendering(e, e, torch.zeros(0, dtype=torch.long, device=device), 3,
                              rgb_sigma_fn=lambda a, b, c: (None, None))
    assert c.shape == (3, 3) and float(o.abs().sum()) == 0.0


def test_rendering_field_matches_oracle(device):
    from quadraturefields_amd import field_rendering as fr
    ridx, _ = helpers.packed_segments(64, 12, seed=8, empty_frac=0.2)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(3)
    ts = torch.rand(n, generator=g)
    te = ts + 0.02
    rgbs, sig = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 40
    want = ov.rendering_field(ts, te, ridx, 64, rgb_sigma_fn=lambda a, b, c: (rgbs, sig))
    dev = lambda t: t.to(device)
    got = fr.rendering_field(dev(ts), dev(te), dev(ridx), 64, rgb_sigma_fn=lambda a, b, c: (dev(rgbs), dev(sig)))
    for a, b in zip(got, want):
        _close(a, b, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("bg", ["white", "black", "random"])
def test_derive_properties_matches_oracle(device, bg):
    from quadraturefields_amd import spc_render, utils
    n_rays = 777
    ridx, _ = helpers.packed_segments(n_rays, 25, seed=11)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(4)
    color = torch.rand(n, 3, generator=g)
    density = torch.rand(n, generator=g) * 300
    density[::5] = 0.0
    density[3::11] = 5e4                          # saturated samples
    depth = torch.rand(n, generator=g) * 5
    deltas = torch.full((n,), 0.005)
    bk = torch.tensor([0.1, 0.6, 0.3])
    boundary_o = ov.mark_pack_boundaries(ridx)
    want = ov.derive_properties(color, density, depth, deltas, boundary_o, ridx, render_bkgd=bk, bg_color=bg, N=n_rays)
    dev = lambda t: t.to(device)
    boundary = spc_render.mark_pack_boundaries(dev(ridx))
    assert torch.equal(boundary.cpu(), boundary_o)
    got = utils.derive_properties(dev(color), dev(density), dev(depth), dev(deltas), boundary, dev(ridx),
                                  render_bkgd=dev(bk), bg_color=bg, N=n_rays)
    assert torch.equal(got[2].cpu(), want[2])     # ids of rays that have samples: bit exact
    for k in (0, 1, 3, 4):
        _close(got[k], want[k], atol=2e-6, rtol=2e-5)
    # kaolin duck types, piecewise
    tau = (density * deltas)[:, None]
    rc, w = spc_render.exponential_integration(dev(color), dev(tau), boundary, exclusive=True)
    rc_o, w_o = ov.exponential_integration(color, tau, boundary_o, exclusive=True)
    _close(rc, rc_o, atol=2e-6, rtol=2e-5)
    _close(w, w_o, atol=2e-6, rtol=2e-5)
    _close(spc_render.sum_reduce(w, boundary), ov.sum_reduce(w_o, boundary_o), atol=2e-6, rtol=2e-5)


def test_derive_properties_closed_forms(device):
    """single-sample ray, zero density, saturated density, empty ray."""
    from quadraturefields_amd import utils
    ridx = torch.tensor([0, 2, 2, 3], device=device)
    color = torch.tensor([[0.2, 0.4, 0.6], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.5, 0.5, 0.5]], device=device)
    density = torch.tensor([100.0, 0.0, 1e9, 0.0], device=device)
    depth = torch.tensor([2.0, 1.0, 3.0, 4.0], device=device)
    rgb, alpha, hit, dep, w = utils.derive_properties(color, density, depth, 0.005, None, ridx, N=5)
    a0 = 1 - np.exp(-0.5)
    assert abs(float(alpha[0]) - a0) < 1e-6
    # white background with the reference's double alpha: (1-a) + a*(a*c)
    assert torch.allclose(rgb[0].cpu(), torch.tensor([(1 - a0) + a0 * a0 * c for c in (0.2, 0.4, 0.6)], dtype=torch.float32), atol=1e-6)
    assert abs(float(dep[0]) - a0 * 2.0) < 1e-6
    assert torch.equal(rgb[1].cpu(), torch.ones(3)) and float(alpha[1]) == 0.0            # empty ray
    assert torch.allclose(rgb[2].cpu(), torch.tensor([0.0, 1.0, 0.0]), atol=1e-6)           # opaque second sample
    assert abs(float(alpha[2]) - 1.0) < 1e-6 and abs(float(dep[2]) - 3.0) < 1e-6
    assert torch.equal(rgb[3].cpu(), torch.ones(3)) and float(alpha[3]) == 0.0            # zero density
    assert torch.equal(rgb[4].cpu(), torch.ones(3))
    assert w.shape == (4, 1)


def test_derive_properties_chunking_long_rays_and_sample_index(device):
    """The compositing kernel stages 1024-sample chunks (+64 halo) in LDS: rays that straddle chunks, rays longer than
    the staged window (occupancy-grid marching: hundreds of samples, finished from global memory), a sample count that
    is not a multiple of the chunk; against the oracle, and the ``sample_index`` form (colour / density stored in another order) bit-identical to the direct one."""
    from quadraturefields_amd import utils
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 40, size=3000)
    counts[rng.random(3000) < 0.02] = rng.integers(200, 2500, size=int((rng.random(3000) < 0.02).sum()) or 1)[0]
    counts[17] = 1500                                   # longer than chunk + halo
    counts[18] = 0
    ridx = torch.from_numpy(np.repeat(np.arange(3000), counts)).long()
    n = ridx.shape[0]
    assert n % 1024 != 0
    g = torch.Generator().manual_seed(8)
    color, density = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 30
    depth, deltas = torch.rand(n, generator=g) * 5, torch.rand(n, generator=g) * 0.01
    want = ov.derive_properties(color, density, depth, deltas, ov.mark_pack_boundaries(ridx), ridx, bg_color="white", N=3000)
    dev = lambda t: t.to(device)
    got = utils.derive_properties(dev(color), dev(density), dev(depth), dev(deltas), None, dev(ridx), N=3000)
    for k in (0, 1, 3, 4):
        _close(got[k], want[k], atol=5e-6, rtol=5e-5)
    perm = torch.randperm(n, generator=g)
    inv = torch.empty(n, dtype=torch.int32)
    inv[perm] = torch.arange(n, dtype=torch.int32)
    got2 = utils.derive_properties(dev(color[perm]), dev(density[perm]), dev(depth), dev(deltas), None, dev(ridx), N=3000,
                                   sample_index=dev(inv))
    for k in (0, 1, 3, 4):
        assert torch.equal(got2[k], got[k])


@pytest.mark.parametrize("bg", ["white", "black", "random"])
@pytest.mark.parametrize("w,h", [(50, 37), (64, 8), (8, 8), (3, 2)])
def test_tile_compositor_equals_the_chunked_kernel(device, bg, w, h):
    """qf_composite_tiles (colours / densities / depths in the coherent tile-rank-pixel order) against
    qf_derive_properties on the ray-major samples: pixels bit for bit -- with and without samples, image sizes that
    are not multiples of the 8x8 tile -- and the weights, through the inverse map."""
    import ctypes
    from types import SimpleNamespace
    from quadraturefields_amd import _C, utils
    rng = np.random.default_rng(w * 100 + h)
    n_rays, k = w * h, 25
    counts = rng.integers(0, 40, size=n_rays).astype(np.int32)       # some above K: clamped like the pack does
    counts[rng.random(n_rays) < 0.3] = 0
    counts[0] = 31
    counts[-1] = 0
    used = np.minimum(counts, k)
    ridx = torch.from_numpy(np.repeat(np.arange(n_rays), used)).long()
    offsets = torch.from_numpy(np.concatenate([[0], np.cumsum(used)])).long().to(device)
    n = ridx.shape[0]
    hit_count = torch.from_numpy(counts).to(device)
    clamped = torch.from_numpy(used.astype(np.int32)).to(device)
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    totals = torch.empty((tiles,), dtype=torch.int64, device=device)
    _C.check(_C.lib().qf_tile_totals(_C.ptr(clamped), w, h, _C.ptr(totals), _C.stream()), "qf_tile_totals")
    tile_base = (torch.cumsum(totals, 0) - totals).contiguous()
    inverse = torch.empty((n,), dtype=torch.int32, device=device)
    _C.check(_C.lib().qf_coherent_layout(_C.ptr(clamped), _C.ptr(offsets), _C.ptr(tile_base), w, h, None, _C.ptr(inverse),
                                         0, _C.stream()), "qf_coherent_layout")
    g = torch.Generator().manual_seed(4)
    color, density = torch.rand(n, 3, generator=g).to(device), (torch.rand(n, generator=g) * 300).to(device)
    depth = (torch.rand(n, generator=g) * 5).to(device)
    bk = torch.rand(3, generator=g).to(device)
    inv = inverse.long()
    color_c, density_c, depth_c = torch.empty_like(color), torch.empty_like(density), torch.empty_like(depth)
    color_c[inv], density_c[inv], depth_c[inv] = color, density, depth
    frame = SimpleNamespace(depth_c=depth_c, hit_count=hit_count, max_hits=k, tile_base=tile_base, width=w, height=h)
    a = utils.derive_properties(color, density, depth, 5e-3, None, ridx.to(device), render_bkgd=bk, bg_color=bg, N=n_rays)
    rgb, alpha, dep, wts = utils.composite_frame(color_c, density_c, frame, 5e-3, render_bkgd=bk, bg_color=bg, want_weights=True)
    assert torch.equal(rgb, a[0]) and torch.equal(alpha, a[1]) and torch.equal(dep, a[3])
    assert torch.equal(wts[inv], a[4])
    assert utils.composite_frame(color_c, density_c, frame, 5e-3, render_bkgd=bk, bg_color=bg)[3] is None
    with pytest.raises(ValueError):
        utils.composite_frame(color_c[:-1], density_c, frame, 5e-3)


def test_derive_properties_ray_ids_outside_the_image(device):
    """N = 0 (the reference signature's default) with samples raises like the reference's index error instead of
    writing through a NULL / too small buffer, and a ray id >= N contributes to no pixel (forward and backward)."""
    from quadraturefields_amd import utils
    g = torch.Generator().manual_seed(0)
    n = 40
    ridx = torch.sort(torch.randint(0, 6, (n,), generator=g)).values
    col, sig, dep = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 50, torch.rand(n, generator=g)
    dev = lambda t: t.to(device)
    with pytest.raises(IndexError):
        utils.derive_properties(dev(col), dev(sig), dev(dep), 5e-3, None, dev(ridx), N=0)
    rgb6, a6, _, d6, w6 = utils.derive_properties(dev(col), dev(sig), dev(dep), 5e-3, None, dev(ridx), N=6)
    rgb4, a4, _, d4, w4 = utils.derive_properties(dev(col), dev(sig), dev(dep), 5e-3, None, dev(ridx), N=4)   # rays 4, 5 outside
    torch.cuda.synchronize()
    assert torch.equal(rgb4, rgb6[:4]) and torch.equal(a4, a6[:4]) and torch.equal(d4, d6[:4]) and torch.equal(w4, w6)
    with torch.enable_grad():
        c = dev(col).requires_grad_(True)
        s = dev(sig).requires_grad_(True)
        out = utils.derive_properties(c, s, dev(dep), 5e-3, None, dev(ridx), N=4)
        (out[0].sum() + out[1].sum()).backward()
    outside = dev(ridx) >= 4
    assert torch.isfinite(c.grad).all() and torch.isfinite(s.grad).all()
    assert float(c.grad[outside].abs().max()) == 0.0 and float(s.grad[outside].abs().max()) == 0.0
    assert float(s.grad[~outside].abs().max()) > 0.0
