"""tiny-cuda-nn duck types backed by the gfx950 kernels.

The reference builds its fields from ``tcnn.Encoding``, ``tcnn.Network`` and
``tcnn.NetworkWithInputEncoding`` (``examples/radiance_fields/ngp.py:325-358,693-746``,
``examples/field.py:157-171``).  These classes take the same constructor arguments and JSON config keys
and keep ONE flat fp32 ``params`` Parameter in tcnn's order (network weights, then grid rows; SURVEY.md
A.1-A.2), so the reference's checkpoints load with ``load_state_dict``.  Parameters are stored and
evaluated in fp32 (tcnn computes in fp16); outputs are fp32.
"""
import math

import numpy as np
import torch
from torch import nn

from . import _C


def _next_multiple(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class _GridConfig:
    """Parses a tcnn grid encoding config and owns the level table."""

    def __init__(self, n_input_dims: int, cfg: dict):
        otype = cfg.get("otype", "HashGrid")
        if otype not in ("HashGrid", "Grid"):
            raise NotImplementedError(f"encoding otype {otype!r}")
        if otype == "Grid" and cfg.get("type", "Hash") != "Hash":
            raise NotImplementedError("only hash grids are implemented")
        if cfg.get("interpolation", "Linear") != "Linear":
            raise NotImplementedError("only linear interpolation is implemented")
        if n_input_dims != 3:
            raise NotImplementedError("only 3-D grids are implemented")
        self.n_levels = int(cfg.get("n_levels", 16))
        self.n_features = int(cfg.get("n_features_per_level", 2))
        self.log2_hashmap_size = int(cfg.get("log2_hashmap_size", 19))
        self.base_resolution = int(cfg.get("base_resolution", 16))
        self.per_level_scale = float(cfg.get("per_level_scale", 2.0))
        if self.n_levels != 16 or self.n_features != 2:
            raise NotImplementedError("the gfx950 field kernels are built for n_levels=16, n_features_per_level=2")
        self.desc = _C.make_grid_desc(self.n_levels, self.log2_hashmap_size, self.base_resolution,
                                      self.per_level_scale)
        self.n_rows = int(self.desc.offset[self.n_levels])
        self.n_params = self.n_rows * self.n_features
        self.n_output_dims = self.n_levels * self.n_features


def _mlp_dims(n_in: int, n_out: int, cfg: dict):
    if cfg.get("otype", "FullyFusedMLP") not in ("FullyFusedMLP", "CutlassMLP"):
        raise NotImplementedError(cfg.get("otype"))
    if cfg.get("activation", "ReLU") != "ReLU" or cfg.get("output_activation", "None") != "None":
        raise NotImplementedError("only ReLU hidden / linear output MLPs are implemented")
    n_neurons = int(cfg.get("n_neurons", 64))
    n_hidden = int(cfg.get("n_hidden_layers", 1))
    dims = [_next_multiple(n_in, 16)] + [n_neurons] * n_hidden + [_next_multiple(n_out, 16)]
    return dims


def _xavier_flat(dims, generator=None) -> torch.Tensor:
    chunks = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        chunks.append((torch.rand(fan_out * fan_in, generator=generator) * 2 - 1) * bound)
    return torch.cat(chunks)


class Encoding(nn.Module):
    """tcnn.Encoding: hash grid (field.py:157-171) or the SH-degree-4 composite (ngp.py:325-338)."""

    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=None, seed: int = 1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.encoding_config = encoding_config
        otype = encoding_config.get("otype")
        if otype == "Composite":
            nested = encoding_config.get("nested", [])
            if len(nested) != 1 or nested[0].get("otype") != "SphericalHarmonics" or nested[0].get("degree") != 4:
                raise NotImplementedError("only the single SphericalHarmonics(degree=4) composite is implemented")
            self.grid = None
            self.n_output_dims = 16
            self.params = nn.Parameter(torch.zeros(0, dtype=torch.float32))
        else:
            self.grid = _GridConfig(n_input_dims, encoding_config)
            self.n_output_dims = self.grid.n_output_dims
            # tcnn default: U(-1e-4, 1e-4)
            self.params = nn.Parameter((torch.rand(self.grid.n_params) * 2 - 1) * 1e-4)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.grid is None:
            raise NotImplementedError(
                "the SH encoding is evaluated inside the fused field kernel (NGPRadianceField.forward)")
        x = _C.f32c(x.reshape(-1, 3))
        n = x.shape[0]
        out = torch.empty((n, 32), dtype=torch.float32, device=x.device)
        _C.check(_C.lib().qf_grid_encode(self.grid.desc, _C.ptr(self.params.detach()), _C.ptr(x), n,
                                         _C.ptr(out), _C.stream()), "qf_grid_encode")
        return out


class Network(nn.Module):
    """tcnn.Network (FullyFusedMLP): parameter container in tcnn's flat row-major [out,in] layout (A.2).
    Its forward runs inside the fused field kernel; see NGPRadianceField."""

    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, seed: int = 1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = network_config
        self.dims = _mlp_dims(n_input_dims, n_output_dims, network_config)
        self.params = nn.Parameter(_xavier_flat(self.dims))

    def forward(self, x):
        raise NotImplementedError("tcnn.Network is evaluated inside the fused field kernel")


class NetworkWithInputEncoding(nn.Module):
    """tcnn.NetworkWithInputEncoding for hash grid -> 64-wide, 1-hidden-layer MLP (ngp.py:340-358).
    params = [network (3072) | grid rows * 2]."""

    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: dict, network_config: dict,
                 seed: int = 1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.grid = _GridConfig(n_input_dims, encoding_config)
        self.dims = _mlp_dims(self.grid.n_output_dims, n_output_dims, network_config)
        if self.dims != [32, 64, 16]:
            raise NotImplementedError(f"the fused kernel implements 32->64->16, got {self.dims}")
        self.n_network_params = sum(a * b for a, b in zip(self.dims[:-1], self.dims[1:]))
        net = _xavier_flat(self.dims)
        grid = (torch.rand(self.grid.n_params) * 2 - 1) * 1e-4
        self.params = nn.Parameter(torch.cat([net, grid]))

    # views into the flat parameter vector (no copies)
    def network_params(self) -> torch.Tensor:
        return self.params.detach()[: self.n_network_params]

    def grid_params(self) -> torch.Tensor:
        return self.params.detach()[self.n_network_params:]

    def forward(self, x01: torch.Tensor) -> torch.Tensor:
        x01 = _C.f32c(x01.reshape(-1, 3))
        n = x01.shape[0]
        out = torch.empty((n, 16), dtype=torch.float32, device=x01.device)
        _C.check(_C.lib().qf_grid_mlp_forward(self.grid.desc, _C.ptr(self.grid_params()),
                                              _C.ptr(self.network_params()), _C.ptr(x01), n, _C.ptr(out),
                                              _C.stream()), "qf_grid_mlp_forward")
        return out[:, : self.n_output_dims]
