"""GPU parity: fused field kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Tolerance: every operand and accumulator is fp32 on both sides; the kernels differ from the oracle only in
summation order (MFMA k-order vs MKL sgemm, fma vs mul+add) and libm (expf), so outputs agree to a few 1e-6
relative.  Stated bound: |a-b| <= 2e-5 + 2e-5*|b| on rgb (values in [0,1]) and geo features, 5e-5 relative on
densities (exp amplifies the raw-output error by |raw|).
"""
import numpy as np
import pytest
import torch

from oracle import fields as ofields
from tests import helpers

pytestmark = pytest.mark.gpu


def _make(cls, device, log2_T=12, seed=42, **kw):
    from quadraturefields_amd import synthetic
    torch.manual_seed(0)
    f = cls(aabb=[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], log2_hashmap_size=log2_T, **kw)
    lobes = kw.get("num_g_lobes", 0)
    st = synthetic.seeded_ngp_state(log2_T, f.mlp_base.grid.n_rows, seed=seed, sg_lobes=lobes)
    missing = f.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys
    return f.to(device)


def _close(a, b, atol, rtol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    assert bool((err <= bound).all()), f"max err {err.max().item():.3e}, worst ratio {(err / bound).max().item():.2f}"


@pytest.mark.parametrize("log2_T", [8, 14, 19])
def test_grid_encode_matches_oracle(device, log2_T):
    from quadraturefields_amd import tinycudann as tcnn
    torch.manual_seed(1)
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                            "log2_hashmap_size": log2_T, "base_resolution": 16,
                            "per_level_scale": ofields.ngp_per_level_scale(4096, 16, 16)})
    with torch.no_grad():
        enc.params.copy_((torch.rand_like(enc.params) * 2 - 1))
    enc = enc.to(device)
    x = torch.rand(4099, 3)
    x[:7] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.999999, 0.5, 0.25], [1e-7, 0.5, 1.0],
                          [-0.25, 0.5, 0.5], [1.3, -0.1, 0.5], [0.5, 0.5, 0.5]])
    lv = ofields.grid_levels(16, log2_T, 16, ofields.ngp_per_level_scale(4096, 16, 16))
    want = ofields.hash_encode(x, enc.params.detach().cpu().reshape(-1, 2), lv)
    got = enc(x.to(device))
    _close(got, want, 2e-6, 2e-6)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4096 + 5])
def test_ngp_forward_matches_oracle(device, n):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device)
    x, d = helpers.random_points(n, seed=n)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.ngp_forward(x, d, w)
    rgb, den = f(x.to(device), d.to(device))
    assert rgb.shape == (n, 3) and den.shape == (n, 1)
    _close(rgb, rgb_o, 2e-5, 2e-5)
    _close(den, den_o, 1e-7, 5e-5)
    # selector: strictly inside the aabb, zero density outside (ngp.py:763)
    sel, _ = ofields.normalize_to_aabb(x, w.aabb)
    assert bool((den.cpu()[~sel] == 0).all())


def test_ngp_query_density_and_feat(device):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device, log2_T=19)
    x, _ = helpers.random_points(2048, seed=3)
    w = helpers.oracle_ngp_weights(f)
    den_o, feat_o = ofields.query_density(x, w, return_feat=True)
    den, feat = f.query_density(x.to(device), return_feat=True)
    assert feat.shape == (2048, 15)
    _close(den, den_o, 1e-7, 5e-5)
    _close(feat, feat_o, 2e-5, 2e-5)
    den2 = f.query_density(x.to(device).reshape(32, 64, 3))
    assert den2.shape == (32, 64, 1)
    assert torch.equal(den2.reshape(-1), den.reshape(-1))
    # the tcnn duck type returns the raw 16 outputs
    sel, x01 = f.normalize(x.to(device))
    raw = f.mlp_base(x01)
    _close(raw[:, 1:], feat_o, 2e-5, 2e-5)


@pytest.mark.parametrize("lobes", [1, 3, 6, 8])
def test_sg_forward_and_features_match_oracle(device, lobes):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    f = _make(NGPRadianceFieldSGNew, device, use_viewdirs=False, num_g_lobes=lobes)
    x, d = helpers.random_points(1000 + lobes, seed=lobes)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.sg_forward(x, d, w)
    feats_o = ofields.sg_features(x, w)
    rgb, den = f(x.to(device), d.to(device))
    feats = f.features(x.to(device))
    assert feats.shape == (x.shape[0], 3 + 7 * lobes + 1)
    _close(den, den_o, 1e-7, 5e-5)
    _close(feats[:, :-1], feats_o[:, :-1], 5e-5, 5e-5)
    _close(feats[:, -1], feats_o[:, -1], 1e-7, 5e-5)
    _close(rgb, rgb_o, 5e-5, 5e-5)
    # features_to_rgb on the oracle's features reproduces the oracle rgb
    rgb2 = f.features_to_rgb(feats_o[:, :-1].contiguous().to(device), d.to(device))
    _close(rgb2, ofields.features_to_rgb(feats_o[:, :-1], d, lobes), 5e-6, 5e-6)


def test_two_call_form_matches_forward(device):
    """query_density(return_feat=True) + _query_rgb (ngp.py:781-796, 428-443) is forward in two launches."""
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    x, d = helpers.random_points(3000, seed=9)
    xd, dd = x.to(device), d.to(device)
    f = _make(NGPRadianceField, device)
    rgb, den = f(xd, dd)
    den2, emb = f.query_density(xd, return_feat=True)
    assert torch.equal(den2, den)
    _close(f._query_rgb(dd, emb), rgb, 5e-6, 5e-6)
    raw = f._query_rgb(dd, emb, apply_act=False)
    _close(torch.sigmoid(raw), rgb, 5e-6, 5e-6)
    w = helpers.oracle_ngp_weights(f)
    _close(f._query_rgb(dd, emb), ofields.ngp_forward(x, d, w)[0], 2e-5, 2e-5)
    g = _make(NGPRadianceFieldSGNew, device, use_viewdirs=False, num_g_lobes=3)
    rgb, den = g(xd, dd)
    den2, emb = g.query_density(xd, return_feat=True)
    _close(g._query_rgb(dd, emb), rgb, 2e-5, 2e-5)
    # per-lobe form (ngp.py:371-393)
    feats = g.features(xd)
    mix = g.spherical_gaussian_mixture(feats[:, 3:-1], dd)
    one = sum(g.spherical_gaussian(c, dd) for c in torch.chunk(feats[:, 3:-1], 3, dim=-1))
    _close(mix, one, 1e-7, 1e-6)
    _close(torch.sigmoid(feats[:, :3] + mix), rgb, 2e-5, 2e-5)


def test_sg_discretized_forward(device):
    """discretize=True (ngp.py:377-382): forward = sigmoid(diffuse + mixture of the quantise-dequantised lobes); the
    diffuse colour is only quantised by features_to_rgb (ngp.py:458-459)."""
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    f = _make(NGPRadianceFieldSGNew, device, use_viewdirs=False, num_g_lobes=3, discretize=True)
    x, d = helpers.random_points(2000, seed=5)
    w = helpers.oracle_ngp_weights(f)
    feats_o = ofields.sg_features(x, w)
    want = torch.sigmoid(feats_o[:, :3] + ofields.spherical_gaussian_mixture(feats_o[:, 3:-1], d, 3, discretize=True))
    rgb, den = f(x.to(device), d.to(device))
    _close(den, feats_o[:, -1:], 1e-7, 5e-5)
    err = (rgb.cpu() - want).abs().max(dim=1).values       # a code point may flip where a value sits on a step
    assert (err > 5e-5).float().mean() < 0.02 and err.max() < 0.1
    plain = _make(NGPRadianceFieldSGNew, device, use_viewdirs=False, num_g_lobes=3)
    assert (plain(x.to(device), d.to(device))[0] - rgb).abs().max() > 1e-3      # the codecs do change the colour


def test_deform_field_matches_oracle(device):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    torch.manual_seed(0)
    f = Field(scale=1.5, precision=16, log2_T=16, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
              num_features=2, back_prop=False, nl="relu")
    f.load_state_dict(synthetic.seeded_deform_state(f.xyz_encoder.grid.n_params), strict=False)
    f = f.to(device)
    x, _ = helpers.random_points(3001, seed=9, outside_frac=0.0)
    want = ofields.deform_field(x, helpers.oracle_deform_weights(f))
    got, grad = f(x.to(device), return_grad=False)
    assert grad is None and got.shape == (3001, 1)
    _close(got, want, 2e-5, 2e-5)
    # a processing permutation only changes the order in which points are visited
    order = torch.randperm(3001, generator=torch.Generator().manual_seed(1)).to(torch.int32).to(device)
    assert torch.equal(f(x.to(device), return_grad=False, order=order)[0], got)
    big, _ = helpers.random_points(70001, seed=10, outside_frac=0.0)     # enough groups for the XCD-contiguous mapping
    want_big = ofields.deform_field(big, helpers.oracle_deform_weights(f))
    _close(f(big.to(device), return_grad=False)[0], want_big, 2e-5, 2e-5)


def test_deform_field_at_the_reference_table_size(device):
    """The deformation field as the reference builds it (train_finetune.py:387-399: ``Field(log2_T=24)``, 101.6 M rows,
    0.8 GB fp32 -- four times the Infinity Cache, levels 0-10 dense and 11-15 hashed): ``deform_kernel`` against the
    oracle on a few thousand points (the oracle only gathers from the host copy of the table), ray-major and through a
    processing permutation."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    f = Field(scale=1.5, precision=16, log2_T=24, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
              num_features=2, back_prop=False, nl="relu")
    g = f.xyz_encoder.grid
    lv = ofields.grid_levels(g.n_levels, g.log2_hashmap_size, g.base_resolution, g.per_level_scale)
    assert g.n_rows == lv.n_entries and 101_000_000 < g.n_rows < 102_000_000
    assert lv.hashed == [False] * 11 + [True] * 5                     # dense levels 0-10 at T = 2^24
    f.load_state_dict(synthetic.seeded_deform_state(g.n_params), strict=False)
    wts = helpers.oracle_deform_weights(f)
    f = f.to(device)
    x, _ = helpers.random_points(4001, seed=24, outside_frac=0.0)
    want = ofields.deform_field(x, wts)
    got = f(x.to(device), return_grad=False)[0]
    _close(got, want, 2e-5, 2e-5)
    order = torch.randperm(4001, generator=torch.Generator().manual_seed(2)).to(torch.int32).to(device)
    assert torch.equal(f(x.to(device), return_grad=False, order=order)[0], got)
    # points on a surface patch a pixel apart (what a coherent wave pass holds): neighbours share cells on every level
    base = torch.tensor([0.31, -0.42, 0.77])
    patch = base + 1e-3 * torch.randn(2048, 3, generator=torch.Generator().manual_seed(3))
    _close(f(patch.to(device), return_grad=False)[0], ofields.deform_field(patch, wts), 2e-5, 2e-5)
    del f, wts
    torch.cuda.empty_cache()


def test_linearity_of_grid_in_table(device):
    """Size-independent property at the full T=2^19 table: the encoding is linear in the table."""
    from quadraturefields_amd import tinycudann as tcnn
    cfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
           "base_resolution": 16, "per_level_scale": ofields.ngp_per_level_scale(4096, 16, 16)}
    a, b = tcnn.Encoding(3, cfg).to(device), tcnn.Encoding(3, cfg).to(device)
    with torch.no_grad():
        a.params.uniform_(-1, 1)
        b.params.uniform_(-1, 1)
    x = torch.rand(200000, 3, device=device)
    ea, eb = a(x), b(x)
    with torch.no_grad():
        a.params.add_(b.params)
    assert torch.allclose(a(x), ea + eb, atol=1e-5, rtol=1e-5)
    # constant table -> constant features (the trilinear weights sum to one)
    with torch.no_grad():
        a.params.fill_(0.75)
    assert torch.allclose(a(x), torch.full((200000, 32), 0.75, device=device), atol=1e-6)


def test_errors(device):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device)
    with pytest.raises(RuntimeError):
        f(torch.zeros(4, 3), torch.zeros(4, 3))          # host tensors: no CPU fallback
    with pytest.raises(AssertionError):
        f(torch.zeros(4, 3, device=device), torch.zeros(5, 3, device=device))
    rgb, den = f(torch.zeros(0, 3, device=device), torch.zeros(0, 3, device=device))
    assert rgb.shape == (0, 3) and den.shape == (0, 1)


@pytest.mark.parametrize("n", [1, 17, 5000])
def test_bf16_ngp_matches_bf16_oracle(device, n):
    """BASELINE config 3 (bf16 tables + MLPs, fp32 accumulate on v_mfma_f32_16x16x32_bf16).  The oracle rounds the
    same quantities to bf16 at the same places, so the two differ only by fp32 summation order -- except where a
    value sits on a bf16 rounding boundary and the 1-ulp-of-fp32 difference flips it (one bf16 step = 2^-8 relative).
    Stated tolerance: 2e-3 + 2e-3*|x| on rgb / features, 1e-2 relative on densities; fp32-vs-bf16 itself is ~1e-2."""
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device, log2_T=14)
    f.compute_dtype = "bf16"
    x, d = helpers.random_points(n, seed=n + 100)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.ngp_forward_bf16(x, d, w)
    rgb, den = f(x.to(device), d.to(device))
    _close(rgb, rgb_o, 2e-3, 2e-3)
    _close(den, den_o, 1e-6, 1e-2)
    den2, feat = f.query_density(x.to(device), return_feat=True)
    _close(feat, ofields.query_density_bf16(x, w)[1], 2e-3, 2e-3)
    assert torch.equal(den2, den)
    # and it is close to (but not the same as) the fp32 evaluation
    f.compute_dtype = "fp32"
    rgb32, _ = f(x.to(device), d.to(device))
    assert float((rgb32 - rgb).abs().max()) < 0.08
    if n > 100:
        assert float((rgb32 - rgb).abs().max()) > 1e-5


@pytest.mark.parametrize("lobes", [3, 6])
def test_bf16_sg_matches_bf16_oracle(device, lobes):
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    f = _make(NGPRadianceFieldSGNew, device, log2_T=14, use_viewdirs=False, num_g_lobes=lobes)
    f.compute_dtype = "bf16"
    x, d = helpers.random_points(3000, seed=lobes + 50)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.sg_forward_bf16(x, d, w)
    rgb, den = f(x.to(device), d.to(device))
    _close(den, den_o, 1e-6, 1e-2)
    _close(rgb, rgb_o, 5e-3, 5e-3)
    # order does not change bf16 results either
    order = torch.randperm(3000, device=device).to(torch.int32)
    rgb2, den2 = f(x.to(device), d.to(device), order=order)
    assert torch.equal(rgb, rgb2) and torch.equal(den, den2)


def test_stale_order_is_refused(device):
    """radiance_field(points, dirs, order=...) validates the permutation instead of indexing out of bounds."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=12)
    field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    x, d = helpers.random_points(256, seed=3)
    x, d = x.to(device), d.to(device)
    good = torch.randperm(256, device=device).to(torch.int32)
    a = field(x, d, order=good)
    b = field(x, d)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for bad in (good[:100], good.long(), good.cpu()):
        with pytest.raises(ValueError):
            field(x, d, order=bad)


def test_ngp_rgb_and_density_at_the_baseline_table_size(device):
    """BASELINE configs[1]: T = 2^19 (6 299 960 rows), the size bench.py runs -- colour AND density against the oracle,
    in ray-major and in a permuted processing order."""
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device, log2_T=19)
    assert f.mlp_base.grid.n_rows == 6299960
    x, d = helpers.random_points(6000, seed=19)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.ngp_forward(x, d, w)
    rgb, den = f(x.to(device), d.to(device))
    _close(rgb, rgb_o, 2e-5, 2e-5)
    _close(den, den_o, 1e-7, 5e-5)
    order = torch.randperm(6000, device=device).to(torch.int32)
    rgb2, den2 = f(x.to(device), d.to(device), order=order)
    assert torch.equal(rgb, rgb2) and torch.equal(den, den2)


def test_bf16_field_at_the_config3_table_size(device):
    """BASELINE configs[2]: T = 2^21 (22 565 520 rows), bf16 tables + MLPs: field_kernel_bf16 against the oracle that
    rounds to bf16 at the same places (tolerances of test_bf16_ngp_matches_bf16_oracle)."""
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    f = _make(NGPRadianceField, device, log2_T=21)
    assert f.mlp_base.grid.n_rows == 22565520
    f.compute_dtype = "bf16"
    x, d = helpers.random_points(4000, seed=21)
    w = helpers.oracle_ngp_weights(f)
    rgb_o, den_o = ofields.ngp_forward_bf16(x, d, w)
    rgb, den = f(x.to(device), d.to(device))
    _close(rgb, rgb_o, 2e-3, 2e-3)
    _close(den, den_o, 1e-6, 1e-2)
