"""CPU: the oracle against the reference's own known-answer vectors (docstrings of
examples/field_rendering.py) and closed forms."""
import os

import numpy as np
import torch

from oracle import fields as ofields
from oracle import meshpath as om
from oracle import quantize as oq
from oracle import volrend as ov

ALPHAS = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9])
RIDX = torch.tensor([0, 0, 0, 1, 1, 2, 2])
TS = torch.arange(7.0)
TE = TS + 1.0


def test_transmittance_from_alpha():          # field_rendering.py:192-195
    t = ov.render_transmittance_from_alpha(ALPHAS, ray_indices=RIDX, n_rays=3)
    assert torch.allclose(t, torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0]), atol=1e-6)


def test_transmittance_from_density():        # :246-253 (two printed digits)
    t, a = ov.render_transmittance_from_density(TS, TE, ALPHAS, ray_indices=RIDX, n_rays=3)
    assert torch.allclose(t, torch.tensor([1.00, 0.67, 0.30, 1.00, 0.45, 1.00, 1.00]), atol=5e-3)
    assert torch.allclose(a, torch.tensor([0.33, 0.55, 0.095, 0.55, 0.095, 0.00, 0.59]), atol=5e-3)


def test_weight_from_alpha():                 # :298-302
    w, t = ov.render_weight_from_alpha(ALPHAS, ray_indices=RIDX, n_rays=3)
    assert torch.allclose(w, torch.tensor([0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9]), atol=1e-6)
    assert torch.allclose(t, torch.tensor([1.00, 0.60, 0.12, 1.00, 0.20, 1.00, 1.00]), atol=1e-6)


def test_weight_from_density():               # :347-355
    w, t, a = ov.render_weight_from_density(TS, TE, ALPHAS, ray_indices=RIDX, n_rays=3)
    assert torch.allclose(w, torch.tensor([0.33, 0.37, 0.03, 0.55, 0.04, 0.00, 0.59]), atol=6e-3)


def test_visibility():                        # :403-409, :461-471
    want = [True, True, False, True, False, False, True]
    assert ov.render_visibility_from_alpha(ALPHAS, ray_indices=RIDX, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2).tolist() == want
    assert ov.render_visibility_from_density(TS, TE, ALPHAS, ray_indices=RIDX, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2).tolist() == want


def test_rendering_shapes():                  # :60-73
    ts = torch.tensor([0.1, 0.2, 0.1, 0.2, 0.3])
    te = ts + 0.1
    ridx = torch.tensor([0, 0, 1, 1, 1])
    c, o, d, ex = ov.rendering(ts, te, ridx, n_rays=2, rgb_sigma_fn=lambda a, b, r: (torch.rand(5, 3), torch.rand(5)))
    assert c.shape == (2, 3) and o.shape == (2, 1) and d.shape == (2, 1)
    assert {"weights", "alphas", "trans"} <= set(ex)


def test_pack_info_and_scans():
    info = ov.pack_info(torch.tensor([0, 0, 2, 2, 2, 5]), 6)
    assert info.tolist() == [[0, 2], [2, 0], [2, 3], [5, 0], [5, 0], [5, 1]]
    x = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    assert ov.exclusive_sum(x, info).tolist() == [0, 1, 0, 3, 7, 0]
    assert ov.exclusive_prod(x, info).tolist() == [1, 1, 1, 3, 12, 1]
    assert ov.exclusive_sum(torch.tensor([[1.0, 2.0, 3.0]])).tolist() == [[0, 1, 3]]


def test_kaolin_semantics_and_derive_properties_quirks():
    ridx = torch.tensor([1, 1, 4])
    b = ov.mark_pack_boundaries(ridx)
    assert b.tolist() == [True, False, True]
    tau = torch.tensor([[0.5], [1.0], [2.0]])
    feats = torch.tensor([[1.0], [2.0], [3.0]])
    out, w = ov.exponential_integration(feats, tau, b, exclusive=True)
    w0, w1, w2 = 1 - np.exp(-0.5), np.exp(-0.5) * (1 - np.exp(-1.0)), 1 - np.exp(-2.0)
    assert torch.allclose(w.flatten(), torch.tensor([w0, w1, w2], dtype=torch.float32))
    assert torch.allclose(out.flatten(), torch.tensor([w0 + 2 * w1, 3 * w2], dtype=torch.float32))
    color = torch.tensor([[0.5, 0.5, 0.5]] * 3)
    rgb, alpha, hit, depth, ww = ov.derive_properties(color, torch.tensor([100.0, 200.0, 400.0]), torch.tensor([1.0, 2.0, 3.0]),
                                                      torch.full((3,), 0.005), b, ridx, N=6)
    a = w0 + w1
    # double alpha (B-1): (1 - a) + a * (sum w c)
    assert torch.allclose(rgb[1], torch.full((3,), float((1 - a) + a * (a * 0.5))), atol=1e-6)
    assert rgb[0].tolist() == [1, 1, 1] and alpha[0].item() == 0 and hit.tolist() == [1, 4]
    rgb_k = ov.derive_properties(color, torch.tensor([100.0, 200.0, 400.0]), torch.tensor([1.0, 2.0, 3.0]),
                                 torch.full((3,), 0.005), b, ridx, bg_color="black", N=6)[0]
    assert rgb_k[0].tolist() == [0, 0, 0]
    assert torch.allclose(rgb_k[1], torch.full((3,), float(a * a * 0.5)), atol=1e-6)


def test_grid_level_rule_matches_survey_counts():
    b = ofields.ngp_per_level_scale(4096, 16, 16)
    assert abs(b - 1.4472692) < 1e-6
    lv = ofields.grid_levels(16, 19, 16, b)
    assert lv.n_entries == 6299960 and lv.hashed == [False] * 5 + [True] * 11
    assert ofields.grid_levels(16, 21, 16, b).n_entries == 22565520
    lf = ofields.grid_levels(16, 24, 16, ofields.field_per_level_scale(512, 1.5, 16, 16))
    assert round(lf.n_entries / 1e6, 1) == 101.6 and sum(lf.hashed) == 5


def test_hash_encode_closed_forms():
    lv = ofields.grid_levels(16, 12, 16, ofields.ngp_per_level_scale(4096, 16, 16))
    const = torch.full((lv.n_entries, 2), 0.25)
    x = torch.rand(100, 3)
    assert torch.allclose(ofields.hash_encode(x, const, lv), torch.full((100, 32), 0.25), atol=1e-6)
    assert not lv.hashed[0] and lv.offset[1] == 4096   # level 0 is dense 16^3: check against a manual trilinear blend
    table = torch.rand(lv.n_entries, 2)
    x = torch.tensor([[0.213, 0.347, 0.481]])
    pos = (x.double() * 15.0 + 0.5).float()
    gi = torch.floor(pos).long()
    w = pos - gi
    man = torch.zeros(2)
    for c in range(8):
        cc = [gi[0, d] + ((c >> d) & 1) for d in range(3)]
        ww = 1.0
        for d in range(3):
            ww = ww * (w[0, d] if (c >> d) & 1 else 1 - w[0, d])
        man = man + ww * table[(cc[0] + cc[1] * 16 + cc[2] * 256) % 4096]
    assert torch.allclose(ofields.hash_encode(x, table, lv)[0, :2], man, atol=1e-6)


def test_sh4_values():
    d = torch.tensor([[0.0, 0.0, 1.0]])
    sh = ofields.sh4(d)[0]
    assert abs(sh[0] - 0.28209479) < 1e-7 and abs(sh[2] - 0.48860251) < 1e-7
    assert abs(sh[6] - (0.94617470 - 0.31539157)) < 1e-6 and abs(sh[12] - 0.37317633 * 2) < 1e-6
    # orthonormality on the sphere (Monte-Carlo)
    g = torch.Generator().manual_seed(0)
    v = torch.randn(200000, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    gram = (ofields.sh4(v).T @ ofields.sh4(v)) * (4 * np.pi / v.shape[0])
    assert torch.allclose(gram, torch.eye(16), atol=3e-2)


def test_quantiser_quirks():
    u = torch.arange(256, dtype=torch.uint8)
    # B-7: "sigmoid" and "linear" both decode linearly to [-12, 12]
    assert torch.equal(oq.inverse_of_compressed_colors(u, compress_type="sigmoid"), oq.inverse_of_compressed_colors(u, compress_type="linear"))
    assert oq.inverse_of_compressed_colors(u, compress_type="linear")[[0, 255]].tolist() == [-12.0, 12.0]
    # B-9: clipped vs unclipped sigma decode at a = 255
    assert abs(oq.inverse_of_compressed_sigma(u)[255].item() - 2763.102) < 1e-2
    assert torch.isinf(oq.inverse_of_compressed_sigma_unclipped(u)[255])
    # B-8: uint8 wrap of (azimuth - 128)
    ax = oq.inverse_of_azimuth_and_elevation(torch.tensor([0, 128], dtype=torch.uint8), torch.tensor([128, 128], dtype=torch.uint8))
    assert torch.allclose(ax[0], torch.tensor([-1.0, 0.0, 0.0]), atol=1e-6)   # (0-128) wraps to 128 -> azimuth pi
    assert torch.allclose(ax[1], torch.tensor([1.0, 0.0, 0.0]), atol=1e-6)
    # round trip within one step
    s = torch.linspace(0, 500, 100)
    back = oq.inverse_of_compressed_sigma(oq.compress_sigma(s))
    assert torch.all((1 - torch.exp(-back * 0.005)) <= (1 - torch.exp(-s * 0.005)) + 1e-6)


def test_bruteforce_intersector_closed_form():
    om.build()
    v = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [0, 0, 2], [1, 0, 2], [0, 1, 2]], dtype=np.float64)
    f = np.array([[0, 1, 2], [3, 4, 5]])
    bf = om.BruteForceIntersector(v, f)
    o = np.array([[0.2, 0.2, 0.0], [0.2, 0.2, 3.0], [5, 5, 0]], dtype=np.float32)
    d = np.array([[0, 0, 1], [0, 0, -1], [0, 0, 1]], dtype=np.float32)
    tri, t, cnt = bf.hits(o, d, 4)
    assert cnt.tolist() == [2, 2, 0]
    assert tri[0, :2].tolist() == [0, 1] and tri[1, :2].tolist() == [1, 0]      # both faces count, front to back
    assert np.allclose(t[0, :2], [1, 2]) and np.isinf(t[0, 2]) and tri[0, 2] == -1
    tri1, _, cnt1 = bf.hits(o, d, 1)
    assert cnt1.tolist() == [1, 1, 0] and tri1[:, 0].tolist() == [0, 1, -1]      # K-nearest truncation
    s = om.sampling_raytrace_numpy(bf, d, o, 4)
    assert s[2].tolist() == [0, 0, 1, 1] and s[4].tolist() == [0, 1, 1, 0]
    assert np.allclose(s[3], [1, 2, 1, 2]) and s[5] == 0
    assert om.sampling_raytrace_numpy(bf, d[2:], o[2:], 4) is None


def test_ray_generation_conventions():
    c2w = torch.eye(4)[:3]
    o, d = om.generate_rays(c2w, focal=50.0, width=4, height=2)
    assert o.shape == (8, 3) and torch.allclose(d.norm(dim=-1), torch.ones(8))
    # pixel (0,0): x = (0 - 2 + 0.5)/50, y = -(0 - 1 + 0.5)/50, z = -1 (OpenGL)
    want = torch.tensor([-1.5 / 50, 0.5 / 50, -1.0])
    assert torch.allclose(d[0], want / want.norm(), atol=1e-7)


def test_config1_crop_render_on_the_cpu_matches_the_committed_image():
    """BASELINE configs[0] (SURVEY.md 8c fixture 6): the 100x100 crop of the 800x800 synthetic camera rendered end to
    end on the CPU by the oracle -- brute-force multi-hit intersection (re-origin rule on), sampling_raytrace_numpy,
    the loader's casts, hash grid + MLPs, derive_properties -- against the committed tests/golden/crop_ref.npz
    (written by tests/golden/gen_crop.py).  The same crop through the host BVH walk must give the same samples.  No GPU,
    no product code on the data path (the product module only supplies the seeded state dict)."""
    import numpy as np
    import torch
    from oracle import meshpath as om
    from tests.golden import gen_crop
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop_ref.npz"))
    mesh, field, idx, o, d = gen_crop.scene()
    assert np.array_equal(idx.numpy(), z["ray_ids"])
    (rgb, alpha, depth, n, _, _, iray, itri), data = gen_crop.render_oracle(mesh, field, o, d)
    assert n == int(z["n_samples"])
    assert np.array_equal(iray.numpy(), z["index_ray"]) and np.array_equal(itri.numpy(), z["index_tri"])      # bit-exact ids
    assert np.abs(rgb.numpy() - z["rgb"]).max() <= 2e-6          # fp32 summation order of the CPU GEMMs may differ
    assert np.abs(alpha.numpy() - z["alpha"]).max() <= 2e-6 and np.abs(depth.numpy() - z["depth"]).max() <= 2e-5
    frac = float((z["alpha"] > 0).mean())
    assert 0.2 < frac < 0.95                                      # object rim + background in the crop
    (_, _, _, n2, _, _, iray2, itri2), data2 = gen_crop.render_oracle(
        mesh, field, o, d, om.BVHIntersector(mesh.vertices, mesh.faces))
    assert n2 == n and torch.equal(iray2, iray) and torch.equal(itri2, itri)
    for a, b in zip(data, data2):
        assert torch.equal(a, b)
