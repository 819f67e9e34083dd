"""Training side: gradients of the differentiable route -- HIP hash-grid
forward/backward, library-GEMM MLPs, HIP compositing backward -- against torch autograd through the CPU oracle on
the same seeded inputs.  Tolerances are fp32 rounding (the table gradient is an atomic sum: order-dependent)."""
import numpy as np
import pytest
import torch

from oracle import fields as ofields
from oracle import volrend as ov
from tests import helpers


def _leaf(t):
    return t.detach().clone().requires_grad_(True)


def _close(a, b, rtol=2e-4, atol=None):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    atol = atol if atol is not None else rtol * float(b.abs().max())
    return float((a - b).abs().max()) <= atol, float((a - b).abs().max()), float(b.abs().max())


def test_oracle_grid_input_gradient_matches_finite_differences():
    """CPU: pins the autograd of the oracle's hash_encode (the checker of the HIP backward) against central
    differences at points away from cell faces."""
    lv = ofields.grid_levels(16, 12, 16, 1.3819)
    g = torch.Generator().manual_seed(3)
    table = (torch.rand(lv.n_entries, 2, generator=g) * 2 - 1).float()
    x = (torch.rand(64, 3, generator=g) * 0.9 + 0.05).float()
    proj = torch.randn(32, generator=g).double()
    with torch.enable_grad():
        xr = _leaf(x)
        (ofields.hash_encode(xr, table, lv)[:, :8].double() @ proj[:8]).sum().backward()   # coarse levels: h << cell
    h = 1e-4
    for d in range(3):
        e = torch.zeros(3)
        e[d] = h
        fd = ((ofields.hash_encode(x + e, table, lv)[:, :8].double() - ofields.hash_encode(x - e, table, lv)[:, :8].double())
              @ proj[:8]) / (2 * h)
        ok = (fd - xr.grad[:, d].double()).abs() <= 2e-2 * fd.abs().max()
        assert ok.float().mean() > 0.9        # the few points whose +-h straddles a cell face are excluded


@pytest.mark.gpu
@pytest.mark.parametrize("log2_t,n", [(12, 3000), (16, 3000), (19, 40000), (21, 40000)])
def test_grid_encode_backward_vs_oracle_autograd(device, log2_t, n):
    """n = 3000: the quad-atomic scatter.  n = 40 000 (>= 2^15): the partitioned scatter -- at T = 2^19 every level takes
    the LDS walk (<= 27 partitions), at T = 2^21 (22.6 M rows) the dense levels 0-5 take it and the hashed levels 6-15
    (105 partitions each, above kMaxLdsParts) go to the quad atomics: both routes in one gradient."""
    from quadraturefields_amd import tinycudann as tcnn
    cfg = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": log2_t,
           "base_resolution": 16, "per_level_scale": 1.4472692012786865}
    enc = tcnn.Encoding(3, cfg)
    g = torch.Generator().manual_seed(log2_t)
    with torch.no_grad():
        enc.params.copy_((torch.rand(enc.params.shape, generator=g) * 2 - 1) * 0.5)
    lv = ofields.grid_levels(16, log2_t, 16, cfg["per_level_scale"])
    x = torch.rand(n, 3, generator=g).float()
    x[:8] = torch.tensor([0.0, 0.5, 1.0])[None]                       # cell faces and the box boundary
    gout = torch.randn(n, 32, generator=g).float()
    gout[10:20] = 0.0                                                 # rows the scatter may skip
    with torch.enable_grad():
        xo, to = _leaf(x), _leaf(enc.params.reshape(-1, 2))
        fo = ofields.hash_encode(xo, to, lv)
        fo.backward(gout)
        enc = enc.to(device)
        xd = _leaf(x.to(device))
        fd = enc(xd)
        assert fd.requires_grad
        fd.backward(gout.to(device))
    assert torch.equal(fd.detach().cpu(), fo.detach()) or _close(fd, fo, 1e-6)[0]
    ok, err, ref = _close(enc.params.grad.reshape(-1, 2), to.grad, 1e-5)
    assert ok, (err, ref)
    ok, err, ref = _close(xd.grad, xo.grad, 1e-5)
    assert ok, (err, ref)
    # input-only / table-only requests
    with torch.enable_grad():
        enc.params.requires_grad_(False)
        x2 = _leaf(x.to(device))
        enc(x2).backward(gout.to(device))
        assert _close(x2.grad, xo.grad, 1e-5)[0] and enc.params.grad is not None
        enc.params.requires_grad_(True)
        enc.params.grad = None
        enc(x.to(device)).backward(gout.to(device))
        assert _close(enc.params.grad.reshape(-1, 2), to.grad, 1e-5)[0]


def _oracle_leaves(wts):
    wts.table = _leaf(wts.table)
    wts.base = [_leaf(w) for w in wts.base]
    if wts.head_tcnn is not None:
        wts.head_tcnn = [_leaf(w) for w in wts.head_tcnn]
    if getattr(wts, "head_layers", None):
        wts.head_layers = [(_leaf(w), _leaf(b)) for w, b in wts.head_layers]
    return wts


@pytest.mark.gpu
def test_ngp_field_training_route_matches_fused_forward_and_oracle_gradients(device):
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=14)
    field.load_state_dict(synthetic.seeded_ngp_state(14, field.mlp_base.grid.n_rows), strict=False)
    wts = _oracle_leaves(helpers.oracle_ngp_weights(field))
    field = field.to(device)
    x, d = helpers.random_points(4000, seed=5)
    g = torch.Generator().manual_seed(9)
    t_rgb, t_sig = torch.rand(4000, 3, generator=g), torch.rand(4000, 1, generator=g)

    def loss_fn(rgb, sigma, t_rgb, t_sig):
        return ((rgb - t_rgb) ** 2).mean() + 1e-2 * ((torch.log1p(sigma) - t_sig) ** 2).mean()

    rgb_f, sig_f = field(x.to(device), d.to(device))                  # fused kernel (autograd off by default)
    with torch.enable_grad():
        rgb_t, sig_t = field(x.to(device), d.to(device))
        assert rgb_t.requires_grad and sig_t.requires_grad
        loss_fn(rgb_t, sig_t, t_rgb.to(device), t_sig.to(device)).backward()
        rgb_o, sig_o = ofields.ngp_forward(x, d, wts)
        loss_fn(rgb_o, sig_o, t_rgb, t_sig).backward()
    assert _close(rgb_t, rgb_f, atol=2e-6, rtol=0)[0]
    assert torch.allclose(sig_t, sig_f, rtol=2e-5, atol=1e-6)
    assert _close(rgb_t, rgb_o, atol=2e-6, rtol=0)[0]
    gb = field.mlp_base.params.grad
    n_net = field.mlp_base.n_network_params
    for name, got, want in [
        ("base mlp", gb[:n_net], torch.cat([w.grad.reshape(-1) for w in wts.base])),
        ("table", gb[n_net:], wts.table.grad.reshape(-1)),
        ("head mlp", field.mlp_head.params.grad, torch.cat([w.grad.reshape(-1) for w in wts.head_tcnn])),
    ]:
        ok, err, ref = _close(got, want, 3e-4)
        assert ok and ref > 0, (name, err, ref)
    # one optimiser step on the flat parameters lowers the loss (the reference trains these with Adam)
    opt = torch.optim.Adam(field.parameters(), lr=1e-2)
    with torch.enable_grad():
        losses = []
        for _ in range(5):
            opt.zero_grad()
            rgb_t, sig_t = field(x.to(device), d.to(device))
            loss = loss_fn(rgb_t, sig_t, t_rgb.to(device), t_sig.to(device))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize("bg", ["white", "black", "custom"])
def test_sg_field_through_compositing_backward_vs_oracle(device, bg):
    """The finetune loss path (train_finetune.py:489-533): SG field -> derive_properties -> image loss."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceFieldSGNew
    field = NGPRadianceFieldSGNew(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=13, use_viewdirs=False)
    field.load_state_dict(synthetic.seeded_ngp_state(13, field.mlp_base.grid.n_rows, sg_lobes=field.num_g_lobes), strict=False)
    wts = _oracle_leaves(helpers.oracle_ngp_weights(field))
    field = field.to(device)
    n_rays = 300
    ridx, _ = helpers.packed_segments(n_rays, 12, seed=4)
    n = ridx.shape[0]
    x, d = helpers.random_points(n, seed=6, outside_frac=0.02)
    g = torch.Generator().manual_seed(2)
    depth = torch.rand(n, generator=g).float() * 4
    target = torch.rand(n_rays, 3, generator=g)
    bk = torch.tensor([0.2, 0.5, 0.7])
    boundary = ov.mark_pack_boundaries(ridx)
    scale = 30.0                                                     # sigma * 0.005 of order 1

    def image_loss(rgb, alpha, dep, tgt):
        return ((rgb - tgt) ** 2).mean() + 0.1 * (alpha ** 2).mean() + 0.05 * dep.mean()

    with torch.enable_grad():
        rgb_s, sig = field(x.to(device), d.to(device))
        dep_d = _leaf(depth.to(device))
        rgb, alpha, _, dep, w = utils.derive_properties(rgb_s, sig * scale, dep_d, 0.005, boundary.to(device),
                                                        ridx.to(device), render_bkgd=bk.to(device), bg_color=bg, N=n_rays)
        assert rgb.requires_grad and not w.requires_grad
        image_loss(rgb, alpha, dep, target.to(device)).backward()
        rgb_so, sig_o = ofields.sg_forward(x, d, wts)
        dep_o = _leaf(depth)
        rgb_o, alpha_o, _, depo, _ = ov.derive_properties(rgb_so, sig_o.reshape(-1) * scale, dep_o, 0.005, boundary, ridx,
                                                          render_bkgd=bk, bg_color=bg, N=n_rays)
        image_loss(rgb_o, alpha_o, depo, target).backward()
    assert _close(rgb, rgb_o, atol=3e-6, rtol=0)[0]
    ok, err, ref = _close(dep_d.grad, dep_o.grad, 1e-4)
    assert ok and ref > 0, (err, ref)
    n_net = field.mlp_base.n_network_params
    gb = field.mlp_base.params.grad
    pairs = [("table", gb[n_net:], wts.table.grad.reshape(-1)),
             ("base mlp", gb[:n_net], torch.cat([w_.grad.reshape(-1) for w_ in wts.base]))]
    for k, (wo, bo) in zip(("layers.0", "layers.1", "lout"), wts.head_layers):
        mod = dict(field.mlp_head.named_modules())[k]
        pairs += [(k + ".weight", mod.weight.grad, wo.grad), (k + ".bias", mod.bias.grad, bo.grad)]
    for name, got, want in pairs:
        ok, err, ref = _close(got, want, 5e-4)
        assert ok and ref > 0, (name, err, ref)


@pytest.mark.gpu
def test_derive_properties_backward_direct(device):
    """Compositing backward alone, with per-sample deltas, empty rays and a zero-density ray."""
    from quadraturefields_amd import utils
    n_rays = 500
    ridx, counts = helpers.packed_segments(n_rays, 25, seed=11)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(1)
    color, sigma = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 60
    first = int(torch.nonzero(counts)[0])
    sigma[ridx == first] = 0.0
    depth, deltas = torch.rand(n, generator=g) * 5, torch.rand(n, generator=g) * 0.02
    boundary = ov.mark_pack_boundaries(ridx)
    g_rgb, g_a, g_d = torch.randn(n_rays, 3, generator=g), torch.randn(n_rays, 1, generator=g), torch.randn(n_rays, 1, generator=g)
    with torch.enable_grad():
        co, so, do = _leaf(color), _leaf(sigma), _leaf(depth)
        r, a, _, dd, _ = ov.derive_properties(co, so, do, deltas, boundary, ridx, bg_color="white", N=n_rays)
        torch.autograd.backward([r, a, dd], [g_rgb, g_a, g_d])
        cd, sd, dd_ = _leaf(color.to(device)), _leaf(sigma.to(device)), _leaf(depth.to(device))
        r2, a2, _, d2, _ = utils.derive_properties(cd, sd, dd_, deltas.to(device), boundary.to(device), ridx.to(device),
                                                   bg_color="white", N=n_rays)
        torch.autograd.backward([r2, a2, d2], [g_rgb.to(device), g_a.to(device), g_d.to(device)])
    for name, got, want in (("color", cd.grad, co.grad), ("sigma", sd.grad, so.grad), ("depth", dd_.grad, do.grad)):
        ok, err, ref = _close(got, want, 2e-5)
        assert ok and ref > 0, (name, err, ref)


@pytest.mark.gpu
def test_deformation_field_gradients(device):
    """Field.forward(return_grad=True) (examples/field.py:206-238): d field / d x and parameter gradients."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    f = Field(scale=1.5, back_prop=1, log2_T=13, L=16, max_res=512, hidden_size=32, nl="relu")
    f.load_state_dict(synthetic.seeded_deform_state(f.xyz_encoder.grid.n_params), strict=False)
    wts = helpers.oracle_deform_weights(f)
    wts.table = _leaf(wts.table)
    wts.layers = [(_leaf(w), _leaf(b)) for w, b in wts.layers]
    f = f.to(device)
    x, _ = helpers.random_points(2000, aabb_half=1.4, seed=8, outside_frac=0.0)
    fused = f(x.to(device), return_grad=False)[0]
    with torch.enable_grad():
        xd = x.to(device)
        val, grad = f(xd, return_grad=True)
        (val.sum() + (grad ** 2).sum().detach()).backward()
        xo = _leaf(x)
        vo = ofields.deform_field(xo, wts)
        go = torch.autograd.grad(vo.sum(), xo, retain_graph=True)[0]
        vo.sum().backward()
    assert _close(val, fused, atol=2e-6, rtol=0)[0] and _close(val, vo, atol=2e-6, rtol=0)[0]
    ok, err, ref = _close(grad, go, 1e-4)
    assert ok and ref > 0, (err, ref)
    ok, err, ref = _close(f.xyz_encoder.params.grad.reshape(-1, 2), wts.table.grad, 1e-4)
    assert ok and ref > 0, (err, ref)
    ok, err, ref = _close(f.decoder_field.lout.weight.grad, wts.layers[2][0].grad, 1e-4)
    assert ok and ref > 0, (err, ref)


@pytest.mark.gpu
def test_finetune_training_step_matches_oracle(device):
    """One optimisation step of the finetune stage (train_finetune.py:465-533): random barycentric samples,
    deformation along the rays, re-sort, SH field, compositing, image loss + regulariser; gradients of both
    networks against autograd through the oracle's restatement; MeshFinetune.update_d bookkeeping."""
    from oracle import meshpath as om
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune, MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=13)
    field.load_state_dict(synthetic.seeded_ngp_state(13, field.mlp_base.grid.n_rows), strict=False)
    net = Field(scale=1.5, precision=16, log2_T=13, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                num_features=2, back_prop=False, nl="relu")
    net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)
    wts = _oracle_leaves(helpers.oracle_ngp_weights(field))
    dwts = helpers.oracle_deform_weights(net)
    dwts.table = _leaf(dwts.table)
    dwts.layers = [(_leaf(w_), _leaf(b_)) for w_, b_ in dwts.layers]
    field, net = field.to(device), net.to(device)
    w = h = 48
    c2w = synthetic.orbit_cameras(1, seed=3)[0]
    o, d = synthetic.camera_rays(c2w, synthetic.lego_focal(800) * w / 800.0, w, h)
    data = om.to_loader_tensors(mi.sampling_raytrace_numpy(d.numpy(), o.numpy(), 0))
    n = data[0].shape[0]
    scaling = 0.0434
    target = torch.rand(w * h, 3, generator=torch.Generator().manual_seed(1))
    finetune = MeshFinetune(mesh.vertices, mesh.faces, scaling, device=device)
    rays = Rays(origins=o, viewdirs=d)
    with torch.enable_grad():
        torch.manual_seed(123)
        out = utils.render_image_finetune_with_occgrid(field, net, None, rays, data, render_step_size=5e-3,
                                                       mesh_intersect=mi, mesh_finetune=finetune, scaling=scaling)
        rgb, loss_reg = out[0].reshape(-1, 3), out[7]
        assert rgb.requires_grad and loss_reg.requires_grad and loss_reg.shape == (1,)
        (torch.nn.functional.smooth_l1_loss(rgb, target.to(device)) + loss_reg.sum()).backward()
        # the same uniform draws the product made (same device generator, same seed, same shape)
        torch.manual_seed(123)
        bw = torch.rand((n, 3), device=device).cpu()
        xyzs, dirs, index_ray, ts, index_tri, origins = data
        tri_v = torch.from_numpy(mesh.vertices.astype(np.float32))[torch.from_numpy(mesh.faces.astype(np.int64))[index_tri]]
        verts = om.barycentric_vertex_samples(tri_v, bw)
        reg_o = om.finetune_regulariser(dwts, xyzs, verts, scaling)
        rgb_o = om.render_image_finetune(wts, dwts, data, w * h, scaling=scaling)[0]
        (torch.nn.functional.smooth_l1_loss(rgb_o, target) + reg_o.sum()).backward()
    assert (rgb.detach().cpu() - rgb_o.detach()).abs().max().item() <= 3e-4
    assert abs(float(loss_reg.detach()) - float(reg_o.detach())) <= 1e-6 * max(1.0, abs(float(reg_o.detach())))
    n_net = field.mlp_base.n_network_params
    pairs = [("ngp table", field.mlp_base.params.grad[n_net:], wts.table.grad.reshape(-1)),
             ("ngp base", field.mlp_base.params.grad[:n_net], torch.cat([w_.grad.reshape(-1) for w_ in wts.base])),
             ("ngp head", field.mlp_head.params.grad, torch.cat([w_.grad.reshape(-1) for w_ in wts.head_tcnn])),
             ("deform table", net.xyz_encoder.params.grad.reshape(-1, 2), dwts.table.grad),
             ("deform l0", net.decoder_field.layers[0].weight.grad, dwts.layers[0][0].grad),
             ("deform out bias", net.decoder_field.lout.bias.grad, dwts.layers[2][1].grad)]
    # The deformed positions differ by an ulp between the two sides, so a handful of points next to a cell face
    # scatter into neighbouring rows: compare in the L2 norm (and the maximum loosely), not entry by entry.
    for name, got, want in pairs:
        got, want = got.detach().cpu().double(), want.detach().double()
        rel = float((got - want).norm() / want.norm())
        assert rel <= 1e-2 and float((got - want).abs().max()) <= 2e-2 * float(want.abs().max()), (name, rel)
    # update_d accumulated |dh| * w per triangle (mesh_utils.py:126-134)
    assert float(finetune.cache_w.sum()) > 1.0 and float(finetune.cache_d.abs().sum()) > 0.0
    # inference through the same entry point is unchanged by the training step's bookkeeping
    img = utils.render_image_finetune_with_occgrid(field, net, None, rays, data, render_step_size=5e-3,
                                                   mesh_intersect=mi, scaling=scaling)
    assert not img[0].requires_grad and (img[0].reshape(-1, 3).cpu() - rgb_o.detach()).abs().max().item() <= 3e-4


@pytest.mark.gpu
def test_finetune_loop_end_to_end(device):
    """A miniature train_finetune.py: target images from one field, a perturbed copy trained against them through the
    reference-named entry points (intersect -> render_image_finetune_with_occgrid -> loss -> backward -> Adam, with
    MeshFinetune.update_d every step), then update_faces + BVH refit (train_finetune.py:708-718) and evaluation.
    The image error must drop and the refitted intersector must agree with a freshly built one."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune, MeshIntersection, RayIntersector
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer, psnr
    torch.manual_seed(0)
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    aabb = [-1.5] * 3 + [1.5] * 3
    target_field = NGPRadianceField(aabb=aabb, log2_hashmap_size=13)
    state = synthetic.seeded_ngp_state(13, target_field.mlp_base.grid.n_rows)
    target_field.load_state_dict(state, strict=False)
    target_field = target_field.to(device)
    field = NGPRadianceField(aabb=aabb, log2_hashmap_size=13)
    g = torch.Generator().manual_seed(1)
    noisy = {k: v + 0.15 * v.abs().mean() * torch.randn(v.shape, generator=g) for k, v in state.items()}
    field.load_state_dict(noisy, strict=False)
    field = field.to(device)
    net = Field(scale=1.5, precision=16, log2_T=13, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                num_features=2, back_prop=False, nl="relu").to(device)
    scaling = 0.02
    w = h = 40
    cams = synthetic.orbit_cameras(4, seed=9)
    focal = synthetic.lego_focal(800) * w / 800.0
    views = []
    with torch.no_grad():
        for c2w in cams:
            o, d = synthetic.camera_rays(c2w, focal, w, h, device=device)
            views.append((o, d, FrameRenderer(mi, target_field).render(o, d, image_width=w)[0]))
    finetune = MeshFinetune(mesh.vertices, mesh.faces, scaling, device=device)
    params = list(field.parameters()) + list(net.parameters())
    opt = torch.optim.Adam(params, lr=2e-3, eps=1e-15)

    def evaluate():
        with torch.no_grad():
            return float(np.mean([psnr(FrameRenderer(mi, field, field_net=net).render(o, d, image_width=w, scaling=scaling)[0], t)
                                  for o, d, t in views]))

    before = evaluate()
    losses = []
    with torch.enable_grad():
        for step in range(40):
            o, d, tgt = views[step % len(views)]
            pick = torch.randperm(w * h, device=device)[:600]               # a random ray batch: BVH intersector
            data = mi.sampling_raytrace_device(d[pick], o[pick])
            rays = Rays(origins=o[pick], viewdirs=d[pick])
            out = utils.render_image_finetune_with_occgrid(field, net, None, rays, data, render_step_size=5e-3,
                                                           mesh_intersect=mi, mesh_finetune=finetune, scaling=scaling)
            loss = torch.nn.functional.smooth_l1_loss(out[0], tgt[pick]) + out[7].sum()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    after = evaluate()
    assert np.mean(losses[-8:]) < 0.7 * np.mean(losses[:8]), (losses[:8], losses[-8:])
    assert after > before + 1.0, (before, after)
    # mesh update and refit (train_finetune.py:708-718)
    assert float(finetune.cache_w.sum()) > 1.0
    finetune.update_faces()
    finetune.reset_d()
    assert np.isfinite(finetune.vertices).all() and np.abs(finetune.vertices - mesh.vertices).max() <= scaling + 1e-6
    mi.mesh.vertices = finetune.vertices
    mi.vertices = torch.from_numpy(finetune.vertices).to(device)
    mi.rayintersector.update_intersector(finetune.vertices)
    o, d, _ = views[0]
    a = mi.rayintersector.hits(o, d, image_width=w)
    fresh = RayIntersector(mi.mesh, max_hits=25, min_separation=mi.rayintersector.min_separation)
    b = fresh.hits(o, d, image_width=w)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert np.isfinite(evaluate())


@pytest.mark.gpu
@pytest.mark.parametrize("log2_t,n_pts", [(13, 1500), (21, 36000)])
def test_field_second_order_gradients(device, log2_t, n_pts):
    """Field.forward(x, return_grad=True) keeps field_grad in the graph (create_graph=True, field.py:206-238); a loss
    on it (field.py:253-264) back-propagates through the second-order grid kernel.  Against double autograd through
    the oracle: the table, the decoder and the input.  36 000 points at T = 2^21: the second-order table scatter on its
    partitioned route (LDS walk for the levels with <= 64 partitions, quad atomics for the rest)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    f = Field(scale=1.5, back_prop=1, log2_T=log2_t, L=16, max_res=512, hidden_size=32, nl="relu")
    f.load_state_dict(synthetic.seeded_deform_state(f.xyz_encoder.grid.n_params), strict=False)
    wts = helpers.oracle_deform_weights(f)
    wts.table = _leaf(wts.table)
    wts.layers = [(_leaf(w), _leaf(b)) for w, b in wts.layers]
    f = f.to(device)
    x, d = helpers.random_points(n_pts, aabb_half=1.4, seed=12, outside_frac=0.0)
    g = torch.Generator().manual_seed(4)
    w1, w2 = torch.rand(n_pts, generator=g), torch.rand(n_pts, generator=g)

    def loss_fn(field, grad, mod, w1, w2, d):
        return mod.compute_field_loss(w1, w2, grad, d) + 0.1 * mod.compute_abs_loss(grad) + 0.01 * field.sum()

    with torch.enable_grad():
        xd = x.to(device)
        val, grad = f(xd, return_grad=True)
        assert grad.requires_grad
        loss = loss_fn(val, grad, f, w1.to(device), w2.to(device), d.to(device))
        loss.backward()
        xo = _leaf(x)
        vo = ofields.deform_field(xo, wts)
        go = torch.autograd.grad(vo.sum(), xo, create_graph=True)[0]
        loss_o = loss_fn(vo, go, f, w1, w2, d)
        loss_o.backward()
    assert abs(float(loss.detach()) - float(loss_o.detach())) <= 1e-4 * abs(float(loss_o.detach()))
    pairs = [("table", f.xyz_encoder.params.grad.reshape(-1, 2), wts.table.grad),
             ("l0.weight", f.decoder_field.layers[0].weight.grad, wts.layers[0][0].grad),
             ("l1.weight", f.decoder_field.layers[1].weight.grad, wts.layers[1][0].grad),
             ("lout.weight", f.decoder_field.lout.weight.grad, wts.layers[2][0].grad),
             ("x", xd.grad, xo.grad)]
    for name, got, want in pairs:
        got, want = got.detach().cpu().double(), want.detach().double()
        rel = float((got - want).norm() / want.norm())
        assert float(want.norm()) > 0 and rel <= 2e-3, (name, rel)


@pytest.mark.gpu
def test_rendering_differentiable_route_vs_oracle(device):
    """nerfacc-style ``rendering`` (density branch, packed samples) with autograd recording: values and the gradients
    w.r.t. per-sample colour and density against autograd through the oracle's restatement of field_rendering.py."""
    from quadraturefields_amd import field_rendering as fr
    n_rays = 400
    ridx, _ = helpers.packed_segments(n_rays, 60, seed=21)
    n = ridx.shape[0]
    g = torch.Generator().manual_seed(5)
    ts = torch.sort(torch.rand(n, generator=g) * 4)[0]
    te = ts + 0.01 + torch.rand(n, generator=g) * 0.01
    rgbs, sig = torch.rand(n, 3, generator=g), torch.rand(n, generator=g) * 40
    bk = torch.tensor([0.3, 0.6, 0.9])
    tgt = torch.rand(n_rays, 3, generator=g)

    def loss_fn(c, o, d, tgt):
        return ((c - tgt) ** 2).mean() + 0.1 * (o ** 2).mean() + 0.01 * d.mean()

    with torch.enable_grad():
        ro, so = _leaf(rgbs), _leaf(sig)
        c_o, o_o, d_o, _ = ov.rendering(ts, te, ridx, n_rays, rgb_sigma_fn=lambda a, b, c: (ro, so), render_bkgd=bk)
        loss_fn(c_o, o_o, d_o, tgt).backward()
        rd, sd = _leaf(rgbs.to(device)), _leaf(sig.to(device))
        c, o, d, extras = fr.rendering(ts.to(device), te.to(device), ridx.to(device), n_rays,
                                       rgb_sigma_fn=lambda a, b, c: (rd, sd), render_bkgd=bk.to(device))
        assert c.requires_grad and extras["weights"].shape == (n,)
        loss_fn(c, o, d, tgt.to(device)).backward()
    assert _close(c, c_o, atol=5e-6, rtol=0)[0] and _close(o, o_o, atol=5e-6, rtol=0)[0] and _close(d, d_o, atol=5e-5, rtol=0)[0]
    for name, got, want in (("rgbs", rd.grad, ro.grad), ("sigmas", sd.grad, so.grad)):
        ok, err, ref = _close(got, want, 1e-4)
        assert ok and ref > 0, (name, err, ref)
    # inference gives the same image through the fused kernel
    c2 = fr.rendering(ts.to(device), te.to(device), ridx.to(device), n_rays,
                      rgb_sigma_fn=lambda a, b, c: (rgbs.to(device), sig.to(device)), render_bkgd=bk.to(device))[0]
    assert _close(c2, c, atol=5e-6, rtol=0)[0]


@pytest.mark.gpu
def test_occgrid_training_step_and_updates(device):
    """The rgb_full half of the finetune step (train_finetune.py:476-533): occupancy refresh with
    update_every_n_steps, stratified marching, differentiable field + rendering, one optimiser step."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.estimators import OccGridEstimator
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    torch.manual_seed(3)
    aabb = [-1.5] * 3 + [1.5] * 3
    field = NGPRadianceField(aabb=aabb, log2_hashmap_size=12)
    field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(device)
    est = OccGridEstimator(roi_aabb=aabb, resolution=32, levels=1).to(device)
    step_size = 0.02
    blob = lambda x: torch.exp(-(x ** 2).sum(-1) / 0.5)       # a smooth occupancy: one sample per cell is representative
    est.eval()
    est.update_every_n_steps(step=0, occ_eval_fn=blob, occ_thre=0.3)
    assert not bool(est.binaries.any())                      # eval mode: no update (nerfacc)
    est.train()
    est.update_every_n_steps(step=7, occ_eval_fn=blob, occ_thre=0.3)
    assert not bool(est.binaries.any())                      # not a multiple of n = 16
    for step in range(0, 64, 16):
        est.update_every_n_steps(step=step, occ_eval_fn=blob, occ_thre=0.3)
    ref = OccGridEstimator(roi_aabb=aabb, resolution=32, levels=1).to(device)
    ref.set_occupancy_from_density(blob, threshold=float(torch.clamp(est.occs.mean(), max=0.3)))
    agree = float((est.binaries == ref.binaries).float().mean())
    assert 0.02 < float(est.binaries.float().mean()) < 0.5 and agree > 0.95, agree
    assert torch.equal(est.binaries.flatten(), est.occs > torch.clamp(est.occs.mean(), max=0.3))
    before = est.occs.clone()
    est.update_every_n_steps(step=512, occ_eval_fn=lambda x: torch.zeros(x.shape[0], device=device), occ_thre=0.3)
    decayed = (est.occs < before).float().mean()             # past the warm-up only a subset of the cells is refreshed
    assert 0.2 < float(decayed) < 0.9 and bool((est.occs[est.occs < before] >= 0.95 * before[est.occs < before] - 1e-7).all())
    occ_fn = lambda x: field.query_density(x) * step_size
    for step in range(0, 64, 16):                            # now the occupancy of the field that is trained below
        est.update_every_n_steps(step=step, occ_eval_fn=occ_fn, occ_thre=0.05)
    assert 0.02 < float(est.binaries.float().mean()) < 0.98
    est.update_every_n_steps(step=512, occ_eval_fn=occ_fn, occ_thre=0.05)     # past the warm-up: subset refresh
    w = h = 16
    c2w = synthetic.orbit_cameras(1, seed=2)[0]
    o, d = synthetic.camera_rays(c2w, synthetic.lego_focal(800) * w / 800.0, w, h, device=device)
    rays = Rays(origins=o, viewdirs=d)
    target = torch.rand(w * h, 3, device=device)
    opt = torch.optim.Adam(field.parameters(), lr=1e-2, eps=1e-15)
    field.train()
    with torch.enable_grad():
        losses = []
        for _ in range(6):
            rgb, acc, depth, n_samples, extras = utils.render_image_with_occgrid(
                field, est, rays, render_step_size=step_size, render_bkgd=torch.ones(3, device=device))
            assert rgb.requires_grad and n_samples > 0
            loss = torch.nn.functional.smooth_l1_loss(rgb, target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    field.eval()
    rgb = utils.render_image_with_occgrid(field, est, rays, render_step_size=step_size,
                                          render_bkgd=torch.ones(3, device=device))[0]
    assert not rgb.requires_grad and bool(torch.isfinite(rgb).all())


@pytest.mark.gpu
def test_fit_sg_renderer_values_and_gradients(device):
    """render_image_fit_sg_with_occgrid (utils.py:610-730): SG colours over the frozen field's densities, inference
    and the training step of train_fit_sg.py:439-461, against the oracle."""
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    from oracle import meshpath as om
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3)
    mi = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25)
    aabb = [-1.5] * 3 + [1.5] * 3
    base = NGPRadianceField(aabb=aabb, log2_hashmap_size=12)
    base.load_state_dict(synthetic.seeded_ngp_state(12, base.mlp_base.grid.n_rows), strict=False)
    sg = NGPRadianceFieldSGNew(aabb=aabb, log2_hashmap_size=12, use_viewdirs=False, num_g_lobes=6)
    sg.load_state_dict(synthetic.seeded_ngp_state(12, sg.mlp_base.grid.n_rows, seed=7, sg_lobes=6), strict=False)
    w_base = helpers.oracle_ngp_weights(base)
    w_sg = _oracle_leaves(helpers.oracle_ngp_weights(sg))
    base, sg = base.to(device), sg.to(device)
    for p in base.parameters():
        p.requires_grad = False
    w = h = 40
    c2w = synthetic.orbit_cameras(1, seed=6)[0]
    o, d = synthetic.camera_rays(c2w, synthetic.lego_focal(800) * w / 800.0, w, h)
    data = om.to_loader_tensors(mi.sampling_raytrace_numpy(d.numpy(), o.numpy(), 0))
    xyzs, dirs, index_ray, ts, index_tri, origins = data
    rays = Rays(origins=o, viewdirs=d)
    target = torch.rand(w * h, 3, generator=torch.Generator().manual_seed(2))
    with torch.enable_grad():
        out = utils.render_image_fit_sg_with_occgrid(base, sg, None, rays, data, render_step_size=5e-3, mesh_intersect=mi)
        assert len(out) == 8 and out[0].requires_grad
        torch.nn.functional.smooth_l1_loss(out[0], target.to(device)).backward()
        rgbs_o, _ = ofields.sg_forward(xyzs, d[index_ray], w_sg)
        sig_o = ofields.query_density(xyzs, w_base).reshape(-1).detach()
        rgb_o = ov.derive_properties(rgbs_o, sig_o, ts, torch.full_like(ts, 5e-3), ov.mark_pack_boundaries(index_ray),
                                     index_ray, bg_color="white", N=w * h)[0]
        torch.nn.functional.smooth_l1_loss(rgb_o, target).backward()
    assert (out[0].detach().cpu() - rgb_o.detach()).abs().max().item() <= 2e-4
    assert base.mlp_base.params.grad is None                 # the density field is frozen and evaluated without gradient
    n_net = sg.mlp_base.n_network_params
    for name, got, want in (("sg table", sg.mlp_base.params.grad[n_net:], w_sg.table.grad.reshape(-1)),
                            ("sg lout", sg.mlp_head.lout.weight.grad, w_sg.head_layers[2][0].grad)):
        got, want = got.detach().cpu().double(), want.detach().double()
        assert float(want.norm()) > 0 and float((got - want).norm() / want.norm()) <= 2e-3, name
    img = utils.render_image_fit_sg_with_occgrid(base, sg, None, rays, data, render_step_size=5e-3, mesh_intersect=mi)[0]
    assert not img.requires_grad and (img.cpu() - rgb_o.detach()).abs().max().item() <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 17, 1000])
def test_fused_backward_equals_library_route(device, n):
    """The fused MLP backward kernels (qf_ngp_mlp_backward, qf_sg_mlp_backward, qf_deform_mlp_backward) against the
    hash-grid-Function + library-GEMM route of the same modules, including ragged groups (n not a multiple of 16)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField, NGPRadianceFieldSGNew
    aabb = [-1.5] * 3 + [1.5] * 3
    x, d = helpers.random_points(n, seed=n, outside_frac=0.2 if n > 4 else 0.0)
    x, d = x.to(device), d.to(device)
    g = torch.Generator().manual_seed(n)
    t_rgb, t_sig = torch.rand(n, 3, generator=g).to(device), torch.rand(n, 1, generator=g).to(device)

    def grads(module, fused, call):
        module.fused_backward = fused
        for p_ in module.parameters():
            p_.grad = None
        with torch.enable_grad():
            call(module).backward()
        return {k: p_.grad.clone() for k, p_ in module.named_parameters() if p_.numel() and p_.grad is not None}

    ngp = NGPRadianceField(aabb=aabb, log2_hashmap_size=12)
    ngp.load_state_dict(synthetic.seeded_ngp_state(12, ngp.mlp_base.grid.n_rows), strict=False)
    sg = NGPRadianceFieldSGNew(aabb=aabb, log2_hashmap_size=12, use_viewdirs=False, num_g_lobes=3)
    sg.load_state_dict(synthetic.seeded_ngp_state(12, sg.mlp_base.grid.n_rows, sg_lobes=3), strict=False)
    net = Field(scale=1.5, back_prop=0, log2_T=12, L=16, max_res=512, hidden_size=32, nl="relu")
    net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)

    def field_loss(m):
        rgb, sig = m(x, d)
        return ((rgb - t_rgb) ** 2).sum() + 1e-2 * ((torch.log1p(sig) - t_sig) ** 2).sum()

    def deform_loss(m):
        return (torch.tanh(m(x, return_grad=False)[0]) * t_sig).sum()

    for module, call in ((ngp.to(device), field_loss), (sg.to(device), field_loss), (net.to(device), deform_loss)):
        a = grads(module, True, call)
        b = grads(module, False, call)
        assert set(a) == set(b) and len(a) >= 2
        for k in a:
            scale = float(b[k].abs().max())
            assert float((a[k] - b[k]).abs().max()) <= 2e-4 * max(scale, 1e-6), (type(module).__name__, k, n)


@pytest.mark.gpu
def test_lds_partitioned_table_scatter_equals_the_atomic_one(device):
    """qf_grid_encode_backward_ws (LDS-partitioned scatter, from 2^15 points on) gives the table gradient of the
    quad-atomic kernel up to fp32 summation order, for a hashed-table config and a tiny all-dense one, and falls back
    to it without a workspace or for a small batch."""
    from quadraturefields_amd import _C
    from quadraturefields_amd import tinycudann as tcnn
    from oracle import fields as ofields
    lib = _C.lib()
    g = torch.Generator().manual_seed(0)
    for log2_t, n in ((19, 70000), (8, 40000), (14, 1000)):
        desc = _C.make_grid_desc(16, log2_t, 16, ofields.ngp_per_level_scale(4096, 16, 16))
        rows = int(desc.offset[16])
        table = (torch.rand(rows * 2, generator=g) - 0.5).to(device)
        x01 = torch.rand(n, 3, generator=g).to(device)
        dfeat = torch.randn(n, 32, generator=g).to(device)
        dfeat[::7] = 0.0
        ga, gb = torch.zeros_like(table), torch.zeros_like(table)
        _C.check(lib.qf_grid_encode_backward(desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), n, _C.ptr(ga), None,
                                             _C.stream()), "bwd")
        _C.grid_encode_backward(desc, table, x01, dfeat, n, gb, None)
        scale = float(ga.abs().max())
        assert scale > 0 and float((ga - gb).abs().max()) <= 2e-5 * max(scale, 1.0)
        assert int((ga != 0).sum()) == int((gb != 0).sum()) or n < (1 << 15)


@pytest.mark.gpu
def test_fused_adam_is_torch_adam(device):
    """quadraturefields_amd.optim.Adam against torch.optim.Adam (the reference's optimiser, train_finetune.py:402-417):
    same parameters after six steps -- incl. a step with an all-zero gradient (dense Adam keeps moving along the
    momentum) and a parameter that received no gradient -- to rounding; the state dicts are interchangeable."""
    from quadraturefields_amd.optim import Adam
    g = torch.Generator().manual_seed(0)
    shapes = [(1000003,), (64, 35), (7,)]
    for kwargs in (dict(lr=1e-2, eps=1e-15), dict(lr=3e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)):
        init = [torch.randn(s, generator=g) for s in shapes]
        pa = [torch.nn.Parameter(t.clone().to(device)) for t in init]
        pb = [torch.nn.Parameter(t.clone().to(device)) for t in init]
        oa, ob = Adam(pa, **kwargs), torch.optim.Adam(pb, **kwargs)
        for it in range(6):
            for i, (a, b) in enumerate(zip(pa, pb)):
                if i == 2 and it % 2:                      # no gradient this step: both optimisers skip the tensor
                    a.grad = b.grad = None
                    continue
                gr = torch.randn(a.shape, generator=g).to(device) * (0.0 if it == 3 else 1.0)
                gr[::7] = 0.0                              # untouched rows
                a.grad, b.grad = gr.clone(), gr.clone()
            oa.step()
            ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
        sa, sb = oa.state_dict(), ob.state_dict()
        assert sa["state"].keys() == sb["state"].keys()
        for k in sa["state"]:
            assert set(sa["state"][k]) == {"step", "exp_avg", "exp_avg_sq"} == set(sb["state"][k])
            assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
            assert torch.allclose(sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"], rtol=2e-6, atol=1e-6)
            assert torch.allclose(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"], rtol=2e-6, atol=1e-8)
        import copy                                        # (load_state_dict does not copy tensors that already have the
        ob.load_state_dict(copy.deepcopy(sa))              #  right dtype / device: without the deepcopy both optimisers
        oa.load_state_dict(copy.deepcopy(ob.state_dict())) #  would share -- and update twice -- one set of moments)
        for a, b in zip(pa, pb):
            a.grad = b.grad = torch.ones_like(a)
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)
    with pytest.raises(NotImplementedError):
        Adam(pa, amsgrad=True)
