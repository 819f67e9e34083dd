"""Instant-NGP radiance fields on the fused gfx950 field kernel.

Same classes, constructor arguments, methods and state-dict keys as
``examples/radiance_fields/ngp.py`` of the reference (``NGPRadianceField`` :657-809,
``NGPRadianceFieldSGNew`` :284-470, ``BasicDecoder`` :35-143, the uint8 quantisers :210-281), so the
reference's checkpoints load and its renderers (``examples/utils.py``) can call them unchanged.  Each
``forward`` / ``query_density`` / ``features`` is ONE launch of ``qf_field_forward``: hash-grid gather,
both MLPs on the fp32 matrix cores and the SH / spherical-Gaussian head never leave registers.
Training: when autograd is recording and something wants a gradient the same methods
take a differentiable route -- HIP hash-grid forward/backward, library GEMMs for the MLPs -- with identical values
to fp32 rounding; ``torch.no_grad()`` / frozen parameters select the fused kernel.
"""
import ctypes
from typing import Callable, List, Union

import numpy as np
import torch
from torch import nn

from .. import _C
from .. import tinycudann as tcnn


def trunc_exp(x):
    """Forward of the reference's _TruncExp (ngp.py:146-153): a plain exp."""
    return torch.exp(x)


class BasicDecoder(nn.Module):
    """nn.Linear + activation stack with the reference's attribute and state-dict names (ngp.py:35-121)."""

    def __init__(self, input_dim, output_dim, activation, bias, layer=nn.Linear, num_layers=1,
                 hidden_dim=128, skip=[]):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.activation, self.bias, self.layer = activation, bias, layer
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        self.skip = [] if skip is None else skip
        layers = []
        for i in range(num_layers):
            if i == 0:
                layers.append(layer(input_dim, hidden_dim, bias=bias))
            elif i in self.skip:
                layers.append(layer(hidden_dim + input_dim, hidden_dim, bias=bias))
            else:
                layers.append(layer(hidden_dim, hidden_dim, bias=bias))
        self.layers = nn.ModuleList(layers)
        self.lout = layer(hidden_dim, output_dim, bias=bias)

    def forward(self, x, return_h=False):
        """Stand-alone evaluation (library GEMMs); the fields below run it inside the fused kernel."""
        h = x
        for i, l in enumerate(self.layers):
            if i != 0 and i in self.skip:
                h = torch.cat([x, h], dim=-1)
            h = self.activation(l(h))
        out = self.lout(h)
        return (out, h) if return_h else out

    def name(self) -> str:
        return "BasicDecoder"


# ------------------------------------------------------------------ quantisers (ngp.py:210-281)
def discretize_axis(axis):
    return ((axis + 1.0) * 255 / 2).to(torch.uint8)


def continuous_axis(axis):
    return axis.to(torch.float32) / 255.0 * 2 - 1


def discretize_color(color):
    return (torch.sigmoid(color) * 255).to(torch.uint8)


def continuous_color(color):
    color = color.to(torch.float32) / 255.0
    return torch.log(torch.clip(color / (1 - color), 1e-8, 1e37))


def compress_polar_coordinates_torch(vectors):
    vectors = vectors / (torch.norm(vectors, dim=-1, keepdim=True) + 1e-6)
    azimuth = (torch.atan2(vectors[..., 1], vectors[..., 0]) * 128 / np.pi + 128).to(torch.uint8)
    elevation = (torch.acos(vectors[..., 2]) * 256 / np.pi).to(torch.uint8)
    return azimuth, elevation


def inverse_of_azimuth_and_elevantion_torch(azimuth, elevation):
    azimuth = (azimuth - 128) / 128 * np.pi      # uint8 inputs wrap mod 256, as in the reference (B-8)
    elevation = elevation / 256 * np.pi
    return torch.stack([torch.cos(azimuth) * torch.sin(elevation),
                        torch.sin(azimuth) * torch.sin(elevation),
                        torch.cos(elevation)], dim=-1)


def compress_lambda_torch(lambdas, compress_threshold=7.5):
    log_lambda = torch.log(torch.clamp(lambdas, 1e-5, np.inf))
    return (255 * torch.clamp((log_lambda + 2.5) / compress_threshold, 0.0, 1.0)).to(torch.uint8)


def torch_invserse_of_compressed_lambda(compressed_lambda, compress_threshold=7.5):
    return torch.exp(compressed_lambda * compress_threshold / 255 - 2.5)


def compress_colors(colors, thres=12, compress_type="sigma"):
    if compress_type == "sigma":       # the scripts pass "sigmoid"/"linear": both take the linear branch (B-7)
        colors = torch.sigmoid(colors)
    else:
        colors = (torch.clip(colors, -thres, thres) + thres) / 2 / thres
    return (colors * 255).to(torch.uint8)


def inverse_of_compressed_colors(colors, thres=12, compress_type="sigma"):
    colors = colors.to(torch.float32) / 255.0
    if compress_type == "sigma":
        return torch.log(torch.clip(colors / (1 - colors), 1e-8, 1e37))
    return colors * 2 * thres - thres


# ------------------------------------------------------------------------------ fields
class _FusedFieldBase(nn.Module):
    """Shared plumbing: aabb buffer, mlp_base, descriptor cache, launch helper."""

    def _init_common(self, aabb, num_dim, use_viewdirs, density_activation, unbounded, base_resolution,
                     max_resolution, geo_feat_dim, n_levels, log2_hashmap_size):
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb.to(torch.float32))
        if unbounded:
            raise NotImplementedError("unbounded scenes (contract_to_unisphere) are outside the hot path")
        if num_dim != 3 or geo_feat_dim != 15:
            raise NotImplementedError("the fused kernel implements num_dim=3, geo_feat_dim=15")
        self.num_dim, self.use_viewdirs = num_dim, use_viewdirs
        self.density_activation, self.unbounded = density_activation, unbounded
        self.base_resolution, self.max_resolution = base_resolution, max_resolution
        self.geo_feat_dim, self.n_levels, self.log2_hashmap_size = geo_feat_dim, n_levels, log2_hashmap_size
        self.per_level_scale = np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1)).tolist()
        self.mlp_base = tcnn.NetworkWithInputEncoding(
            n_input_dims=num_dim,
            n_output_dims=1 + geo_feat_dim,
            encoding_config={"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": 2,
                             "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
                             "per_level_scale": self.per_level_scale},
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": 64, "n_hidden_layers": 1},
        )
        self._desc_cache = None

    def _field_desc(self, head: int, n_lobes: int = 0) -> _C.FieldDesc:
        key = (self.aabb.data_ptr(), self.aabb._version)
        if self._desc_cache is None or self._desc_cache[0] != key:
            aabb_host = self.aabb.detach().to("cpu", torch.float32).tolist()   # one sync per aabb change
            self._desc_cache = (key, aabb_host)
        d = _C.FieldDesc()
        ctypes.memmove(ctypes.byref(d.grid), ctypes.byref(self.mlp_base.grid.desc), ctypes.sizeof(_C.GridDesc))
        for k in range(6):
            d.aabb[k] = self._desc_cache[1][k]
        d.head, d.n_lobes = head, n_lobes
        return d

    #: "fp32" (default; parity with the fp32 oracle to ~1e-6) or "bf16" (bf16 tables + MLPs, fp32 accumulate --
    #: BASELINE config 3; tcnn runs the same networks in fp16).  ``features()`` always evaluates in fp32.
    compute_dtype = "fp32"

    def _bf16_copies(self, head_ngp, head_sg):
        """Round-to-nearest-even bf16 copies of the fp32 master parameters, refreshed when a parameter changes."""
        params = [self.mlp_base.params] + ([head_ngp] if head_ngp is not None else []) + list(head_sg or [])
        key = tuple((t.data_ptr(), t._version) for t in params)
        cache = getattr(self, "_bf16_cache", None)
        if cache is None or cache[0] != key:
            bf = lambda t: t.detach().to(torch.bfloat16).contiguous()
            c = {"base": bf(self.mlp_base.network_params()), "table": bf(self.mlp_base.grid_params())}
            if head_ngp is not None:
                c["head"] = bf(head_ngp)
            if head_sg is not None:
                c["sg"] = [bf(head_sg[0]), bf(head_sg[1]), bf(head_sg[2]), bf(head_sg[4])]
            cache = (key, c)
            self._bf16_cache = cache
        return cache[1]

    def normalize(self, x):
        """(selector, x01) -- ngp.py:748-755; elementwise, kept in torch."""
        aabb_min, aabb_max = torch.split(self.aabb, self.num_dim, dim=-1)
        x = (x - aabb_min) / (aabb_max - aabb_min)
        selector = ((x > 0.0) & (x < 1.0)).all(dim=-1)
        return selector, x

    def _launch(self, head, n_lobes, xyz, dirs, want_rgb=False, want_sigma=False, want_geo=False, want_features=0,
                head_ngp=None, head_sg=None, order=None, enc_out=None, n_device=None):
        xyz = _C.f32c(xyz.reshape(-1, 3))
        n = xyz.shape[0]
        dev = xyz.device
        if dirs is not None:
            dirs = _C.f32c(dirs.reshape(-1, 3))
            if dirs.shape[0] != n:
                raise ValueError(f"{tuple(xyz.shape)} v.s. {tuple(dirs.shape)}")
        if order is not None:     # a stale permutation would index the points out of bounds inside the kernel
            if order.dtype != torch.int32 or order.dim() != 1 or order.shape[0] != n or order.device != dev:
                raise ValueError(f"order must be an int32 [{n}] permutation on {dev}, got {order.dtype} "
                                 f"{tuple(order.shape)} on {order.device}")
            order = order.contiguous()
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_rgb else None
        sigma = torch.empty((n,), dtype=torch.float32, device=dev) if want_sigma else None
        geo = torch.empty((n, 15), dtype=torch.float32, device=dev) if want_geo else None
        feats = torch.empty((n, want_features), dtype=torch.float32, device=dev) if want_features else None
        desc = self._field_desc(head, n_lobes)
        sg = None
        if self.compute_dtype == "bf16" and head in (_C.HEAD_NONE, _C.HEAD_NGP, _C.HEAD_SG):
            c = self._bf16_copies(head_ngp, head_sg)
            if head_sg is not None:   # w1, b1, w2 bf16 | b2 fp32 | wout bf16 | bout fp32
                sg = _C.SGHead(_C.ptr(c["sg"][0]), _C.ptr(c["sg"][1]), _C.ptr(c["sg"][2]), _C.ptr(head_sg[3]),
                               _C.ptr(c["sg"][3]), _C.ptr(head_sg[5]))
            _C.check(_C.lib().qf_field_forward_bf16(
                ctypes.byref(desc), _C.ptr(c["table"]), _C.ptr(c["base"]), _C.ptr(c.get("head")),
                ctypes.byref(sg) if sg is not None else None, _C.ptr(xyz), _C.ptr(dirs), n, _C.ptr(n_device, torch.int64),
                _C.ptr(order, torch.int32), _C.ptr(rgb), _C.ptr(sigma), _C.ptr(geo), _C.stream()), "qf_field_forward_bf16")
            return rgb, sigma, geo, feats
        if head_sg is not None:
            sg = _C.SGHead(*[_C.ptr(t) for t in head_sg])
        _C.check(_C.lib().qf_field_forward(
            ctypes.byref(desc), _C.ptr(self.mlp_base.grid_params()), _C.ptr(self.mlp_base.network_params()),
            _C.ptr(head_ngp), ctypes.byref(sg) if sg is not None else None, _C.ptr(xyz), _C.ptr(dirs), n,
            _C.ptr(n_device, torch.int64), _C.ptr(order, torch.int32), _C.ptr(rgb), _C.ptr(sigma), _C.ptr(geo), _C.ptr(feats),
            _C.ptr(enc_out), _C.stream()),
                 "qf_field_forward")
        return rgb, sigma, geo, feats

    def _recording(self, *inputs) -> bool:
        """Autograd is recording and a parameter or input wants a gradient: take the differentiable route
        (HIP grid forward/backward + library GEMMs) instead of the fused inference kernel."""
        return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in inputs)
                                            or any(p.requires_grad for p in self.parameters()))

    def _query_density_train(self, x):
        selector, x01 = self.normalize(x)
        out = self.mlp_base(x01.reshape(-1, self.num_dim))
        raw, feat = out[:, :1], out[:, 1:1 + self.geo_feat_dim]
        density = torch.exp(raw - 1.0) * selector.reshape(-1, 1)
        return density, feat

    def query_density(self, x, return_feat: bool = False):
        """density = exp(raw - 1) * selector, [..,1] (+ the 15 geometry features).  ngp.py:757-779."""
        lead = list(x.shape[:-1])
        if self._recording(x):
            density, feat = self._query_density_train(x)
            density = density.reshape(lead + [1])
            return (density, feat.reshape(lead + [self.geo_feat_dim])) if return_feat else density
        _, sigma, geo, _ = self._launch(_C.HEAD_NONE, 0, x, None, want_sigma=True, want_geo=return_feat)
        density = sigma.reshape(lead + [1])
        if return_feat:
            return density, geo.reshape(lead + [self.geo_feat_dim])
        return density


class _SGMixtureFn(torch.autograd.Function):
    """rgb = sigmoid(diffuse + SG mixture) with HIP forward and backward (gradient w.r.t. the features only; the view
    directions are data).  Replaces ~20 elementwise torch kernels per lobe in the SG-fitting step."""

    @staticmethod
    def forward(ctx, features, dirs, n_lobes):
        features = _C.f32c(features.detach())
        dirs = _C.f32c(dirs.detach().reshape(-1, 3))
        n = features.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=features.device)
        _C.check(_C.lib().qf_sg_features_to_rgb(_C.ptr(features), features.shape[1], _C.ptr(dirs), n, n_lobes, _C.ptr(rgb),
                                                _C.stream()), "qf_sg_features_to_rgb")
        ctx.save_for_backward(features, dirs)
        ctx.n_lobes = n_lobes
        return rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb):
        features, dirs = ctx.saved_tensors
        n = features.shape[0]
        d_features = torch.zeros_like(features)           # columns past 3+7L (if any) get no gradient
        if n:
            _C.check(_C.lib().qf_sg_features_to_rgb_backward(
                _C.ptr(features), features.shape[1], _C.ptr(dirs), _C.ptr(_C.f32c(d_rgb)), n, ctx.n_lobes,
                _C.ptr(d_features), d_features.shape[1], _C.stream()), "qf_sg_features_to_rgb_backward")
        return d_features, None, None


class _NGPTrainFn(torch.autograd.Function):
    """(rgb, density) of NGPRadianceField with a fused backward: forward = the inference kernel (qf_field_forward),
    backward = grid encode + qf_ngp_mlp_backward (recompute, back-propagate and accumulate both MLPs' weight
    gradients on the matrix cores) + qf_grid_encode_backward (table scatter, input gradient).  First order only; the
    module falls back to the library-GEMM route when a graph of the backward is requested."""

    @staticmethod
    def forward(ctx, positions, directions, base_params, head_params, module):
        xyz = _C.f32c(positions.detach().reshape(-1, 3))
        dirs = _C.f32c(directions.detach().reshape(-1, 3))
        # the forward IS the inference kernel; it also leaves the hash-grid encoding behind for the backward (128 B per
        # point) -- gathering the table a second time there cost 0.64 ms per 0.8 M points
        enc = torch.empty((xyz.shape[0], 32), dtype=torch.float32, device=xyz.device)
        rgb, sigma, _, _ = module._launch(_C.HEAD_NGP, 0, xyz, dirs, want_rgb=True, want_sigma=True,
                                          head_ngp=head_params.detach(), enc_out=enc)
        ctx.save_for_backward(xyz, dirs, base_params, head_params, enc)
        ctx.module = module
        return rgb, sigma

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb, d_sigma):
        xyz, dirs, base_params, head_params, enc = ctx.saved_tensors
        m = ctx.module
        n = xyz.shape[0]
        dev = xyz.device
        base = base_params.detach()
        n_net = m.mlp_base.n_network_params
        net_w, table = base[:n_net].contiguous(), base[n_net:].contiguous()
        head_w = _C.f32c(head_params.detach())
        selector, x01 = m.normalize(xyz)
        x01 = _C.f32c(x01)
        sel = selector.to(torch.uint8).contiguous()
        lib = _C.lib()
        desc = m.mlp_base.grid.desc
        d_enc = torch.empty((n, 32), dtype=torch.float32, device=dev)
        g_net = torch.zeros_like(net_w)
        g_head = torch.zeros_like(head_w)
        need_x, need_base, need_head = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        g_table = torch.zeros_like(table) if need_base else None
        g_x01 = torch.empty_like(x01) if need_x else None
        if n:
            _C.check(lib.qf_ngp_mlp_backward(_C.ptr(enc), _C.ptr(dirs), _C.ptr(sel), _C.ptr(_C.f32c(d_rgb.reshape(-1, 3))),
                                             _C.ptr(_C.f32c(d_sigma.reshape(-1))), _C.ptr(net_w), _C.ptr(head_w), n,
                                             _C.ptr(d_enc), _C.ptr(g_net), _C.ptr(g_head), _C.stream()),
                     "qf_ngp_mlp_backward")
            if need_base or need_x:
                _C.grid_encode_backward(desc, table, x01, d_enc, n, g_table, g_x01)
        g_pos = None
        if need_x:
            lo, hi = torch.split(m.aabb, 3, dim=-1)
            g_pos = g_x01 / (hi - lo)
        g_base = torch.cat([g_net, g_table]) if need_base else None
        return g_pos, None, g_base, (g_head if need_head else None), None


class _SGTrainFn(torch.autograd.Function):
    """(rgb, density) of NGPRadianceFieldSGNew with a fused backward: forward = the inference kernels (features, then
    the SG mixture), backward = SG-mixture backward + grid encode + qf_sg_mlp_backward + qf_grid_encode_backward."""

    @staticmethod
    def forward(ctx, positions, directions, base_params, w1, b1, w2, b2, wout, bout, module):
        xyz = _C.f32c(positions.detach().reshape(-1, 3))
        dirs = _C.f32c(directions.detach().reshape(-1, 3))
        n = xyz.shape[0]
        head = [_C.f32c(t.detach()) for t in (w1, b1, w2, b2, wout, bout)]
        width = 3 + 7 * module.num_g_lobes + 1
        enc = torch.empty((n, 32), dtype=torch.float32, device=xyz.device)        # kept for the backward, see _NGPTrainFn
        _, _, _, feats = module._launch(_C.HEAD_SG_FEATURES, module.num_g_lobes, xyz, None, want_features=width,
                                        head_sg=head, enc_out=enc)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
        _C.check(_C.lib().qf_sg_features_to_rgb(_C.ptr(feats), width, _C.ptr(dirs), n, module.num_g_lobes, _C.ptr(rgb),
                                                _C.stream()), "qf_sg_features_to_rgb")
        ctx.save_for_backward(xyz, dirs, feats, base_params, enc, *head)
        ctx.module = module
        return rgb, feats[:, -1].contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_rgb, d_sigma):
        xyz, dirs, feats, base_params, enc, w1, b1, w2, b2, wout, bout = ctx.saved_tensors
        m = ctx.module
        n = xyz.shape[0]
        dev = xyz.device
        lib = _C.lib()
        L = m.num_g_lobes
        width = feats.shape[1]
        base = base_params.detach()
        n_net = m.mlp_base.n_network_params
        net_w, table = base[:n_net].contiguous(), base[n_net:].contiguous()
        selector, x01 = m.normalize(xyz)
        x01 = _C.f32c(x01)
        sel = selector.to(torch.uint8).contiguous()
        desc = m.mlp_base.grid.desc
        need_x, need_base = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        d_feat = torch.empty((n, 3 + 7 * L), dtype=torch.float32, device=dev)
        d_enc = torch.empty((n, 32), dtype=torch.float32, device=dev)
        g_net = torch.zeros_like(net_w)
        grads = [torch.zeros_like(t) for t in (w1, b1, w2, b2, wout, bout)]
        g_table = torch.zeros_like(table) if need_base else None
        g_x01 = torch.empty_like(x01) if need_x else None
        if n:
            _C.check(lib.qf_sg_features_to_rgb_backward(_C.ptr(feats), width, _C.ptr(dirs), _C.ptr(_C.f32c(d_rgb.reshape(-1, 3))),
                                                        n, L, _C.ptr(d_feat), d_feat.shape[1], _C.stream()),
                     "qf_sg_features_to_rgb_backward")
            head = _C.SGHead(*[_C.ptr(t) for t in (w1, b1, w2, b2, wout, bout)])
            ghead = _C.SGHead(*[_C.ptr(t) for t in grads])
            _C.check(lib.qf_sg_mlp_backward(_C.ptr(enc), _C.ptr(sel), _C.ptr(d_feat), d_feat.shape[1],
                                            _C.ptr(_C.f32c(d_sigma.reshape(-1))), _C.ptr(net_w), ctypes.byref(head), L, n,
                                            _C.ptr(d_enc), _C.ptr(g_net), ctypes.byref(ghead), _C.stream()),
                     "qf_sg_mlp_backward")
            if need_base or need_x:
                _C.grid_encode_backward(desc, table, x01, d_enc, n, g_table, g_x01)
        g_pos = None
        if need_x:
            lo, hi = torch.split(m.aabb, 3, dim=-1)
            g_pos = g_x01 / (hi - lo)
        g_base = torch.cat([g_net, g_table]) if need_base else None
        return (g_pos, None, g_base, *grads, None)


class NGPRadianceField(_FusedFieldBase):
    """Instant-NGP radiance field with the SH-degree-4 view-dependent head (ngp.py:657-809)."""

    #: Training route of ``forward``: True = fused HIP backward (``_NGPTrainFn``: first order, position and parameter
    #: gradients), False = hash-grid autograd Function + library GEMMs (also differentiable a second time).
    fused_backward = True

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 density_activation: Callable = lambda x: trunc_exp(x - 1), unbounded: bool = False,
                 base_resolution: int = 16, max_resolution: int = 4096, geo_feat_dim: int = 15,
                 n_levels: int = 16, log2_hashmap_size: int = 19, num_layers=2, hidden_size=64) -> None:
        super().__init__()
        self._init_common(aabb, num_dim, use_viewdirs, density_activation, unbounded, base_resolution,
                          max_resolution, geo_feat_dim, n_levels, log2_hashmap_size)
        if not use_viewdirs or hidden_size != 64:
            raise NotImplementedError("the fused kernel implements use_viewdirs=True, hidden_size=64")
        self.direction_encoding = tcnn.Encoding(
            n_input_dims=num_dim,
            encoding_config={"otype": "Composite",
                             "nested": [{"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}]})
        self.mlp_head = tcnn.Network(
            n_input_dims=self.direction_encoding.n_output_dims + geo_feat_dim, n_output_dims=3,
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": hidden_size, "n_hidden_layers": 2})

    def _query_rgb(self, dir, embedding, apply_act: bool = True):
        """SH4((dir + 1) / 2) ++ embedding -> mlp_head (-> sigmoid), ngp.py:781-796.  The two-call form of ``forward``
        (density and colour in separate launches); ``forward`` itself takes the fused kernel."""
        d = self.direction_encoding(((dir + 1.0) / 2.0).reshape(-1, dir.shape[-1]))
        h = torch.cat([d, embedding.reshape(-1, self.geo_feat_dim).to(d.dtype)], dim=-1)
        rgb = self.mlp_head(h).reshape(list(embedding.shape[:-1]) + [3]).to(embedding)
        return torch.sigmoid(rgb) if apply_act else rgb

    def forward(self, positions: torch.Tensor, directions: torch.Tensor = None, order: torch.Tensor = None,
                n_device: torch.Tensor = None):
        """(rgb [..,3], density [..,1]).  ngp.py:798-809.  ``order`` (int32 permutation, optional) only changes the
        order in which points are processed (cache locality), never the result.  ``n_device`` (extension, inference
        only): a device int64 scalar -- only the first ``min(n_device, len(positions))`` points are evaluated, the
        arrays are worst-case buffers (a render-only frame whose sample count never reaches the host)."""
        if directions is None:
            raise ValueError("NGPRadianceField.forward needs view directions")
        assert positions.shape == directions.shape, f"{positions.shape} v.s. {directions.shape}"
        lead = list(positions.shape[:-1])
        if self._recording(positions, directions):
            if self.fused_backward and self.compute_dtype == "fp32" and not directions.requires_grad:
                rgb, sigma = _NGPTrainFn.apply(positions, directions, self.mlp_base.params, self.mlp_head.params, self)
                return rgb.reshape(lead + [3]), sigma.reshape(lead + [1])
            density, feat = self._query_density_train(positions)
            sh = self.direction_encoding((directions.reshape(-1, 3) + 1.0) / 2.0)
            rgb = torch.sigmoid(self.mlp_head(torch.cat([sh, feat], dim=-1)))
            return rgb.reshape(lead + [3]), density.reshape(lead + [1])
        rgb, sigma, _, _ = self._launch(_C.HEAD_NGP, 0, positions, directions, want_rgb=True, want_sigma=True,
                                        head_ngp=self.mlp_head.params.detach(), order=order, n_device=n_device)
        return rgb.reshape(lead + [3]), sigma.reshape(lead + [1])


class NGPRadianceFieldSGNew(_FusedFieldBase):
    """Instant-NGP field with the spherical-Gaussian head (ngp.py:284-470)."""

    #: Training route of ``forward``: True = fused HIP backward (``_SGTrainFn``), False = library GEMMs.
    fused_backward = True

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 density_activation: Callable = lambda x: trunc_exp(x - 1), unbounded: bool = False,
                 base_resolution: int = 16, max_resolution: int = 4096, geo_feat_dim: int = 15,
                 n_levels: int = 16, log2_hashmap_size: int = 19, num_g_lobes=3, hidden_size=64, num_layers=2,
                 output_activation="sigmoid", discretize=False) -> None:
        super().__init__()
        self._init_common(aabb, num_dim, use_viewdirs, density_activation, unbounded, base_resolution,
                          max_resolution, geo_feat_dim, n_levels, log2_hashmap_size)
        if use_viewdirs:
            raise NotImplementedError("the reference scripts build the SG field with use_viewdirs=False "
                                      "(train_finetune.py:374-380); only that form is implemented")
        if hidden_size != 64 or num_layers != 2 or not (1 <= num_g_lobes <= _C.QF_MAX_LOBES):
            raise NotImplementedError("the fused kernel implements hidden_size=64, num_layers=2, 1..8 lobes")
        self.num_g_lobes = num_g_lobes
        self.output_activation = output_activation
        self.discretize = discretize
        self.mlp_head = BasicDecoder(input_dim=geo_feat_dim, output_dim=3 + num_g_lobes * 7, num_layers=num_layers,
                                     bias=True, activation=torch.nn.ReLU(), hidden_dim=hidden_size)

    def _sg_params(self):
        h = self.mlp_head
        return [_C.f32c(t.detach()) for t in (h.layers[0].weight, h.layers[0].bias, h.layers[1].weight,
                                              h.layers[1].bias, h.lout.weight, h.lout.bias)]

    def spherical_gaussian(self, x, direction):
        """One lobe [.., 7] = (axis 3, lambda 1, colour 3): c * exp(|lambda| (axis/|axis| . d - 1)), ngp.py:371-383.
        ``discretize``: axis, sharpness and colour go through their uint8 codecs and back first (what a baked texture
        would hold).  Plain tensor ops -- the per-lobe form the reference exposes; ``forward`` / ``features_to_rgb``
        with ``discretize=False`` take the fused kernels instead."""
        axis = x[..., :3]
        axis = axis / torch.linalg.norm(axis, dim=-1, keepdim=True)
        lambda_ = torch.abs(x[..., 3])
        c = x[..., 4:]
        if self.discretize:
            axis = inverse_of_azimuth_and_elevantion_torch(*compress_polar_coordinates_torch(axis))
            lambda_ = torch_invserse_of_compressed_lambda(compress_lambda_torch(lambda_))
            c = inverse_of_compressed_colors(compress_colors(c))
        return c * torch.exp(lambda_ * (torch.sum(axis * direction, -1) - 1))[..., None]

    def spherical_gaussian_mixture(self, x, direction):
        """Sum of the ``num_g_lobes`` lobes packed along the last axis of ``x`` [n, 7L], ngp.py:385-393."""
        rgb = torch.zeros((x.shape[0], 3), dtype=x.dtype, device=x.device)
        for x_ in torch.chunk(x, self.num_g_lobes, dim=-1):
            rgb = rgb + self.spherical_gaussian(x_, direction)
        return rgb

    def _query_rgb(self, dir, embedding, apply_act: bool = True):
        """mlp_head(embedding) -> sigmoid(diffuse + SG mixture), ngp.py:428-443 with use_viewdirs=False (``apply_act``
        is ignored there too).  The two-call form of ``forward``."""
        h = embedding.reshape(-1, self.geo_feat_dim)
        out = self.mlp_head(h).reshape(list(embedding.shape[:-1]) + [self.num_g_lobes * 7 + 3]).to(embedding)
        if self.discretize:
            return torch.sigmoid(out[:, :3] + self.spherical_gaussian_mixture(out[:, 3:], dir))
        return self.features_to_rgb(out, dir)

    def features(self, x):
        """[head(3+7L) | density], ngp.py:445-454."""
        width = 3 + 7 * self.num_g_lobes + 1
        if self._recording(x):
            density, feat = self._query_density_train(x)
            return torch.cat([self.mlp_head(feat), density], dim=-1)
        _, _, _, feats = self._launch(_C.HEAD_SG_FEATURES, self.num_g_lobes, x, None, want_features=width,
                                      head_sg=self._sg_params())
        return feats

    def features_to_rgb(self, features, dir):
        """sigmoid(diffuse + SG mixture), ngp.py:456-461.  features [n, >= 3+7L] (extra columns ignored).
        ``discretize``: diffuse colour and lobes take the uint8 round trip first (tensor ops, not the fused kernel --
        the reference's scripts all run with discretize False)."""
        if self.discretize:
            diffuse = inverse_of_compressed_colors(compress_colors(features[:, :3]))
            return torch.sigmoid(diffuse + self.spherical_gaussian_mixture(
                features[:, 3:3 + 7 * self.num_g_lobes], dir.reshape(-1, 3)))
        if torch.is_grad_enabled() and features.requires_grad and not dir.requires_grad:
            return _SGMixtureFn.apply(features, dir, self.num_g_lobes)
        if torch.is_grad_enabled() and (features.requires_grad or dir.requires_grad):
            dir = dir.reshape(-1, 3)
            rgb = features[:, :3]
            for lobe in torch.chunk(features[:, 3:3 + 7 * self.num_g_lobes], self.num_g_lobes, dim=-1):
                axis = lobe[:, :3] / torch.linalg.norm(lobe[:, :3], dim=-1, keepdim=True)
                sharp = torch.abs(lobe[:, 3])
                rgb = rgb + lobe[:, 4:7] * torch.exp(sharp * (torch.sum(axis * dir, -1) - 1.0))[:, None]
            return torch.sigmoid(rgb)
        features = _C.f32c(features)
        dir = _C.f32c(dir.reshape(-1, 3))
        n = features.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=features.device)
        _C.check(_C.lib().qf_sg_features_to_rgb(_C.ptr(features), features.shape[1], _C.ptr(dir), n,
                                                self.num_g_lobes, _C.ptr(rgb), _C.stream()), "qf_sg_features_to_rgb")
        return rgb

    def forward(self, positions: torch.Tensor, directions: torch.Tensor = None, order: torch.Tensor = None,
                n_device: torch.Tensor = None):
        """(rgb, density), ngp.py:463-470.  ``order`` / ``n_device``: see ``NGPRadianceField.forward``."""
        if directions is None:
            raise ValueError("NGPRadianceFieldSGNew.forward needs view directions")
        lead = list(positions.shape[:-1])
        if self.discretize:           # ngp.py:463-470 through _query_rgb: lobes quantised, diffuse colour not
            f = self.features(positions.reshape(-1, positions.shape[-1]))
            rgb = torch.sigmoid(f[:, :3] + self.spherical_gaussian_mixture(f[:, 3:-1], directions.reshape(-1, 3)))
            return rgb.reshape(lead + [3]), f[:, -1:].reshape(lead + [1])
        if self._recording(positions, directions):
            if self.fused_backward and self.compute_dtype == "fp32" and not directions.requires_grad:
                h = self.mlp_head
                rgb, sigma = _SGTrainFn.apply(positions, directions, self.mlp_base.params, h.layers[0].weight,
                                              h.layers[0].bias, h.layers[1].weight, h.layers[1].bias, h.lout.weight,
                                              h.lout.bias, self)
                return rgb.reshape(lead + [3]), sigma.reshape(lead + [1])
            f = self.features(positions)
            rgb = self.features_to_rgb(f[:, :-1], directions)
            return rgb.reshape(lead + [3]), f[:, -1:].reshape(lead + [1])
        rgb, sigma, _, _ = self._launch(_C.HEAD_SG, self.num_g_lobes, positions, directions, want_rgb=True,
                                        want_sigma=True, head_sg=self._sg_params(), order=order, n_device=n_device)
        return rgb.reshape(lead + [3]), sigma.reshape(lead + [1])
