"""Deformation / quadrature scalar field on the gfx950 kernels.

Mirrors ``Field`` of ``examples/field.py:130-270`` as the render path uses it
(``examples/utils.py:555-566``: ``field_net(x, return_grad=False)[0]``): hash grid (tcnn ``Encoding``)
followed by ``cat[x01, h] -> BasicDecoder``.  Inference is one fused launch; when autograd is recording the
differentiable route (HIP grid forward/backward + library GEMMs) is taken, which also serves ``field_grad`` --
including ``create_graph=True`` (the reference's default), through the second-order grid kernel.
"""
import numpy as np
import torch
from torch import nn

from . import _C
from . import tinycudann as tcnn


class BasicDecoder(nn.Module):
    """examples/field.py's BasicDecoder variant: like ngp.BasicDecoder plus ``bias_last``."""

    def __init__(self, input_dim, output_dim, activation, bias, layer=nn.Linear, num_layers=1, hidden_dim=128,
                 skip=[], bias_last=True):
        super().__init__()
        self.input_dim, self.output_dim, self.activation = input_dim, output_dim, activation
        self.num_layers, self.hidden_dim, self.skip = num_layers, hidden_dim, ([] if skip is None else skip)
        self.layers = nn.ModuleList(
            [layer(input_dim if i == 0 else hidden_dim, hidden_dim, bias=bias) for i in range(num_layers)])
        self.lout = layer(hidden_dim, output_dim, bias=bias_last)

    def forward(self, x):
        h = x
        for l in self.layers:
            h = self.activation(l(h))
        return self.lout(h)


class _DeformTrainFn(torch.autograd.Function):
    """Field.density with a fused backward (first order, parameter gradients only -- the input is data): forward = the
    inference kernel, backward = grid encode + qf_deform_mlp_backward + qf_grid_encode_backward."""

    @staticmethod
    def forward(ctx, x, table, w1, b1, w2, b2, wout, bout, module):
        x = _C.f32c(x.detach().reshape(-1, 3))
        enc = torch.empty((x.shape[0], 32), dtype=torch.float32, device=x.device)     # kept for the backward
        out = module._density_fused(x, None, enc_out=enc)
        ctx.save_for_backward(x, table, w1, b1, w2, b2, wout, bout, enc)
        ctx.module = module
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        x, table, w1, b1, w2, b2, wout, bout, enc = ctx.saved_tensors
        m = ctx.module
        n = x.shape[0]
        dev = x.device
        lib = _C.lib()
        table = _C.f32c(table.detach())
        ws = [_C.f32c(t.detach()) for t in (w1, b1, w2, b2, wout)]
        grads = [torch.zeros_like(t) for t in (w1, b1, w2, b2, wout, bout)]
        g_table = torch.zeros_like(table)
        if n:
            x01 = _C.f32c((x - m.xyz_min) / (m.xyz_max - m.xyz_min))
            desc = m.xyz_encoder.grid.desc
            d_enc = torch.empty((n, 32), dtype=torch.float32, device=dev)
            _C.check(lib.qf_deform_mlp_backward(_C.ptr(enc), _C.ptr(x01), _C.ptr(_C.f32c(d_out.reshape(-1))),
                                                *[_C.ptr(t) for t in ws], n, _C.ptr(d_enc), None,
                                                *[_C.ptr(t) for t in grads], _C.stream()), "qf_deform_mlp_backward")
            _C.grid_encode_backward(desc, table, x01, d_enc, n, g_table, None)
        return (None, g_table, *grads, None)


class Field(nn.Module):
    def __init__(self, scale, back_prop=0, precision=16, log2_T=19, L=16, max_res=512, output_dim=1, min_res=16,
                 hidden_size=32, num_features=2, nl="elu", bias=True, bias_last=True):
        super().__init__()
        if nl != "relu" or output_dim != 1 or hidden_size != 32 or not bias or not bias_last:
            raise NotImplementedError("the fused kernel implements the finetune configuration of "
                                      "train_finetune.py:387-399 (relu, hidden 32, output 1, biases)")
        self.output_dim = output_dim
        self.dtype = torch.float16 if precision == 16 else torch.float32   # kept for API parity; compute is fp32
        self.scale = scale
        self.register_buffer("center", torch.zeros(1, 3))
        self.register_buffer("xyz_min", -torch.ones(1, 3) * scale)
        self.register_buffer("xyz_max", torch.ones(1, 3) * scale)
        self.register_buffer("half_size", (self.xyz_max - self.xyz_min) / 2)
        self.back_prop = back_prop
        b = np.exp(np.log(max_res * scale / min_res) / (L - 1))
        self.xyz_encoder = tcnn.Encoding(
            n_input_dims=3,
            encoding_config={"otype": "Grid", "type": "Hash", "n_levels": L, "n_features_per_level": num_features,
                             "log2_hashmap_size": log2_T, "base_resolution": min_res, "per_level_scale": b,
                             "interpolation": "Linear"},
            dtype=self.dtype)
        self.decoder_field = BasicDecoder(input_dim=L * num_features + 3, output_dim=output_dim,
                                          activation=torch.nn.ReLU(), bias=bias, num_layers=2,
                                          hidden_dim=hidden_size, skip=[], bias_last=bias_last)

    def density(self, x, order=None, n_device=None):
        """[N,3] in [-scale, scale] -> [N,1].  field.py:186-203, one fused launch.  ``order`` (extension): int32
        processing permutation (``RayIntersector.coherent_order``), cache locality only."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if self.fused_backward and not x.requires_grad:
                d = self.decoder_field       # parameters train, the input is data: fused first-order backward
                return _DeformTrainFn.apply(x, self.xyz_encoder.params, d.layers[0].weight, d.layers[0].bias,
                                            d.layers[1].weight, d.layers[1].bias, d.lout.weight, d.lout.bias, self)
            x01 = (x.reshape(-1, 3) - self.xyz_min) / (self.xyz_max - self.xyz_min)
            h = self.xyz_encoder(x01 if self.back_prop else x01.detach())
            return self.decoder_field(torch.cat([x01, h], 1))
        return self._density_fused(x, order, n_device=n_device)

    #: Training route of ``density`` when only the parameters want gradients: True = fused HIP backward
    #: (``_DeformTrainFn``); the input-gradient / second-order route always goes through the hash-grid autograd Function.
    fused_backward = True

    def _density_fused(self, x, order=None, enc_out=None, n_device=None):
        x = _C.f32c(x.reshape(-1, 3))
        n = x.shape[0]
        out = torch.empty((n,), dtype=torch.float32, device=x.device)
        d = self.decoder_field
        w = [_C.f32c(t.detach()) for t in (d.layers[0].weight, d.layers[0].bias, d.layers[1].weight,
                                           d.layers[1].bias, d.lout.weight, d.lout.bias)]
        _C.check(_C.lib().qf_deform_field_forward(
            self.xyz_encoder.grid.desc, _C.ptr(self.xyz_encoder.params.detach()), float(self.scale), 32,
            *[_C.ptr(t) for t in w], _C.ptr(x), n, _C.ptr(n_device, torch.int64),
            _C.ptr(order, torch.int32) if order is not None and order.shape[0] == n else None,
            _C.ptr(out), _C.ptr(enc_out), _C.stream()), "qf_deform_field_forward")
        return out[:, None]

    def field(self, x, order=None, n_device=None):
        return self.density(x, order, n_device)[:, 0:self.output_dim]

    def forward(self, x, return_grad=True, order=None, n_device=None):
        """(field [N,1], field_grad [N,3] or None).  field.py:206-223.  ``order`` / ``n_device`` (extensions, inference
        only): see ``NGPRadianceField.forward``."""
        if not return_grad:
            return self.field(x, order, n_device), None
        if not x.requires_grad:
            x.requires_grad = True
        field = self.field(x)
        return field, self.field_grad(x, field, create_graph=True)

    def field_grad(self, coords, field, create_graph=True):
        """d field / d coords, field.py:229-238 (``create_graph=True`` keeps it differentiable for the losses of
        field.py:240-270)."""
        field = field.flatten()
        return torch.autograd.grad(field, [coords], grad_outputs=torch.ones_like(field), create_graph=create_graph,
                                   retain_graph=True)[0]

    def compute_field_loss(self, weights, weights_rev, field_norm, view_dirs):
        """field.py:253-259."""
        view_dirs = view_dirs / torch.norm(view_dirs, dim=1, keepdim=True)
        return torch.abs(torch.maximum(weights.detach(), weights_rev.detach())
                         - torch.abs(torch.sum(field_norm * view_dirs.detach(), 1))).mean()

    def compute_abs_loss(self, field_norm):
        """field.py:261-264."""
        return torch.linalg.norm(field_norm, ord=1, dim=1).mean()
