"""tiny-cuda-nn duck types backed by the gfx950 kernels.

The reference builds its fields from ``tcnn.Encoding``, ``tcnn.Network`` and
``tcnn.NetworkWithInputEncoding`` (``examples/radiance_fields/ngp.py:325-358,693-746``,
``examples/field.py:157-171``).  These classes take the same constructor arguments and JSON config keys
and keep ONE flat fp32 ``params`` Parameter in tcnn's order (network weights, then grid rows; SURVEY.md
A.1-A.2), so the reference's checkpoints load with ``load_state_dict``.  Parameters are stored and
evaluated in fp32 (tcnn computes in fp16); outputs are fp32.

Training: when autograd is recording and a parameter or the input requires a
gradient, ``forward`` takes the differentiable route -- the hash grid through ``grid_encode`` (HIP forward +
HIP backward, ``qf_grid_encode_backward``) and the 64-wide MLPs as ``F.linear`` on views of the flat parameter
vector, so ``params.grad`` has tcnn's layout and the reference's optimiser / checkpoint code keeps working.
Inference (``torch.no_grad`` or frozen parameters) stays on the fused kernels.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd.function import once_differentiable

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


class _GridInputGradFn(torch.autograd.Function):
    """gx = J(x01; table)^T dfeat -- the input gradient of the grid, itself differentiable (second order:
    ``Field.field_grad(create_graph=True)``, examples/field.py:229-238).  Backward: qf_grid_encode_double_backward."""

    @staticmethod
    def forward(ctx, dfeat, x01, table, desc):
        dfeat, x01, table = _C.f32c(dfeat.detach()), _C.f32c(x01.detach()), _C.f32c(table.detach())
        n = x01.shape[0]
        gx = torch.empty_like(x01)
        if n:
            _C.check(_C.lib().qf_grid_encode_backward(desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), n, None,
                                                      _C.ptr(gx), _C.stream()), "qf_grid_encode_backward")
        ctx.save_for_backward(dfeat, x01, table)
        ctx.desc = desc
        return gx

    @staticmethod
    @once_differentiable
    def backward(ctx, v):
        dfeat, x01, table = ctx.saved_tensors
        need_d, need_x, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        v = _C.f32c(v)
        n = x01.shape[0]
        g_d = torch.empty_like(dfeat) if need_d else None
        g_x = torch.empty_like(x01) if need_x else None
        g_t = torch.zeros_like(table) if need_t else None
        if n and (need_d or need_x or need_t):
            ws, ws_bytes = None, 0
            if need_t and n >= (1 << 15):          # large batch: the partitioned table scatter (see _C.grid_encode_backward)
                ws_bytes = int(_C.lib().qf_grid_backward_workspace_bytes(n))
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x01.device)
            _C.check(_C.lib().qf_grid_encode_double_backward(
                ctx.desc, _C.ptr(table), _C.ptr(x01), _C.ptr(dfeat), _C.ptr(v), n, _C.ptr(g_d), _C.ptr(g_x), _C.ptr(g_t),
                _C.ptr(ws), ws_bytes, _C.stream()), "qf_grid_encode_double_backward")
        return g_d, g_x, g_t, None


class _GridEncodeFn(torch.autograd.Function):
    """feat = grid(x01; table).  Backward: table gradient by atomic scatter, input gradient by the analytic
    derivative of the trilinear blend (tcnn's kernel_grid_backward / kernel_grid_backward_input).  The input
    gradient is differentiable once more (``_GridInputGradFn``); the table gradient is not (nothing in the reference
    differentiates a parameter gradient)."""

    @staticmethod
    def forward(ctx, x01, table, desc):
        x01_c = _C.f32c(x01.detach().reshape(-1, 3))
        table_c = _C.f32c(table.detach())
        n = x01_c.shape[0]
        out = torch.empty((n, 32), dtype=torch.float32, device=x01_c.device)
        _C.check(_C.lib().qf_grid_encode(desc, _C.ptr(table_c), _C.ptr(x01_c), n, _C.ptr(out), _C.stream()), "qf_grid_encode")
        ctx.save_for_backward(x01, table)        # the graph tensors: the second-order path differentiates w.r.t. them
        ctx.desc = desc
        return out

    @staticmethod
    def backward(ctx, dfeat):
        x01, table = ctx.saved_tensors
        need_x, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        x01_c = _C.f32c(x01.detach().reshape(-1, 3))
        table_c = _C.f32c(table.detach())
        n = x01_c.shape[0]
        gx = gt = None
        if need_t:
            gt = torch.zeros_like(table_c)
            if n:
                _C.grid_encode_backward(ctx.desc, table_c, x01_c, _C.f32c(dfeat.detach()), n, gt, None)
        if need_x:
            if torch.is_grad_enabled() and (dfeat.requires_grad or x01.requires_grad or table.requires_grad):
                # create_graph=True: keep the input gradient in the graph
                gx = _GridInputGradFn.apply(dfeat, x01.reshape(-1, 3) if x01.dim() != 2 else x01, table, ctx.desc)
            else:
                gx = _GridInputGradFn.apply(dfeat.detach(), x01_c, table_c, ctx.desc)
            gx = gx.reshape(x01.shape)
        return gx, gt, None


def grid_encode(x01: torch.Tensor, table: torch.Tensor, desc) -> torch.Tensor:
    """Differentiable hash-grid encoding: x01 [n,3], flat table [rows*2] -> [n,32]."""
    return _GridEncodeFn.apply(x01, table, desc)


def recording(*tensors) -> bool:
    """True when autograd would record an op on these tensors (the switch between the fused inference kernels
    and the differentiable route)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _mlp_nobias(x: torch.Tensor, flat: torch.Tensor, dims) -> torch.Tensor:
    """FullyFusedMLP as library GEMMs on views of the flat row-major [out,in] parameter vector."""
    o, h = 0, x
    last = len(dims) - 2
    for k, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        h = F.linear(h, flat[o:o + a * b].view(b, a))
        if k != last:
            h = F.relu(h)
        o += a * b
    return h


def sh4(d: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics up to degree 4 (16 values) of unit vectors [n,3], tcnn's sign convention."""
    x, y, z = d.unbind(-1)
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * (x2 - y2), 0.59004358992664352 * y * (y2 - 3.0 * x2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (3.0 * y2 - x2)], dim=-1)


def _grid_rows_for(log2_t: int, cfg: "_GridConfig") -> int:
    return int(_C.make_grid_desc(cfg.n_levels, log2_t, cfg.base_resolution, cfg.per_level_scale).offset[cfg.n_levels])


def _explain_params_size(owner: str, got: int, n_network: int, grid: "_GridConfig") -> str:
    """The message a wrong-sized ``params`` gets: what the flat vector is made of here, and which hash-map size the
    checkpoint's length would fit (the usual cause: a checkpoint trained with another ``log2_hashmap_size``)."""
    want = n_network + (grid.n_params if grid is not None else 0)
    parts = [f"{owner}: checkpoint `params` has {got} entries, this module expects {want}"]
    if grid is not None:
        parts.append(f"= [{n_network} network weights | {grid.n_features} x {grid.n_rows} grid rows] "
                     f"(log2_hashmap_size={grid.log2_hashmap_size}, n_levels={grid.n_levels}, "
                     f"base_resolution={grid.base_resolution}, per_level_scale={grid.per_level_scale:.6g}; "
                     "tcnn's flat order, SURVEY.md A.1-A.2, restated from memory)")
        fits = [t for t in range(10, 25) if n_network + grid.n_features * _grid_rows_for(t, grid) == got]
        if fits:
            parts.append(f"-- the checkpoint's length fits log2_hashmap_size={fits[0]} with the same levels")
    else:
        parts.append(f"= {n_network} network weights (row-major [out,in] per layer, inputs padded to 16)")
    return " ".join(parts)


def _rms(t: torch.Tensor) -> float:
    return float(t.detach().double().pow(2).mean().sqrt()) if t.numel() else 0.0


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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        v = state_dict.get(prefix + "params")
        if v is not None and self.grid is not None and v.numel() != self.params.numel():
            error_msgs.append(_explain_params_size(f"{prefix}Encoding", v.numel(), 0, self.grid))
            state_dict = {k: t for k, t in state_dict.items() if k != prefix + "params"}       # (torch would only repeat it)
            strict = False
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.grid is None:
            return sh4(x.reshape(-1, 3) * 2.0 - 1.0)      # tcnn maps its [0,1] input back to [-1,1]
        if recording(x, self.params):
            return grid_encode(x.reshape(-1, 3), self.params, self.grid.desc)
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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        v = state_dict.get(prefix + "params")
        if v is not None and v.numel() != self.params.numel():
            error_msgs.append(_explain_params_size(f"{prefix}Network {self.dims}", v.numel(), self.params.numel(), None))
            state_dict = {k: t for k, t in state_dict.items() if k != prefix + "params"}
            strict = False
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, x):
        """Library-GEMM evaluation (training route); inference runs inside the fused field kernel.  tcnn pads the
        input to a multiple of 16 with ones and returns the first n_output_dims columns."""
        x = x.reshape(-1, self.n_input_dims).to(torch.float32)
        pad = self.dims[0] - self.n_input_dims
        if pad:
            x = torch.cat([x, torch.ones((x.shape[0], pad), dtype=x.dtype, device=x.device)], dim=-1)
        return _mlp_nobias(x, self.params, self.dims)[:, : self.n_output_dims]


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

    #: ``layout_check`` verdicts
    LAYOUT_OK, LAYOUT_SUSPECT = "network|grid", "grid|network?"

    def layout_check(self, flat: torch.Tensor):
        """Does a flat ``params`` vector look like ``[network | grid]`` -- the order this module assumes -- or like the
        transposed ``[grid | network]``?  The order is the one from-memory fact about tcnn whose failure
        would be silent and total on a real checkpoint: the sizes match either way and every output is garbage.  The two
        halves are told apart by magnitude: tcnn initialises the tables U(-1e-4, 1e-4) and after training the finest
        levels' rows are still orders of magnitude smaller in RMS than the 3 072 dense Xavier-scale MLP weights.  Under
        the assumed order rms(first 3 072) / rms(rest) is large; under the transposed one rms(LAST 3 072) / rms(the rest
        before it) is.  Returns (verdict, assumed_ratio, transposed_ratio)."""
        n = self.n_network_params
        flat = flat.detach().reshape(-1)
        if flat.numel() != self.params.numel() or flat.numel() <= 2 * n:
            return self.LAYOUT_OK, float("nan"), float("nan")
        eps = 1e-30
        assumed = _rms(flat[:n]) / (_rms(flat[n:]) + eps)
        transposed = _rms(flat[-n:]) / (_rms(flat[:-n]) + eps)
        suspect = transposed > 1.0 and transposed > 4.0 * assumed
        return (self.LAYOUT_SUSPECT if suspect else self.LAYOUT_OK), assumed, transposed

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Reference checkpoints (ngp.py:709-727; train_finetune.py:407-409 ``load_state_dict``): a wrong-sized ``params``
        gets an error that names the expected ``[3072 | 2 x rows]`` split; a right-sized one is checked for the transposed
        order and WARNED about (not refused: the heuristic cannot prove it)."""
        v = state_dict.get(prefix + "params")
        if v is not None and v.numel() != self.params.numel():
            error_msgs.append(_explain_params_size(f"{prefix}NetworkWithInputEncoding", v.numel(), self.n_network_params,
                                                   self.grid))
            state_dict = {k: t for k, t in state_dict.items() if k != prefix + "params"}
            strict = False
        elif v is not None:
            verdict, assumed, transposed = self.layout_check(v)
            if verdict != self.LAYOUT_OK:
                import warnings
                n = self.n_network_params
                warnings.warn(
                    f"{prefix}params: this checkpoint looks like [grid | network], not the [network ({n}) | grid "
                    f"({self.grid.n_features} x {self.grid.n_rows} rows)] order this module assumes for tcnn's flat "
                    f"parameter vector (SURVEY.md A.2, restated from memory): rms(first {n}) / rms(rest) = {assumed:.3g} "
                    f"but rms(last {n}) / rms(before) = {transposed:.3g} -- the dense MLP weights seem to sit at the END.  "
                    "Loaded as is; if the render is garbage, move the last "
                    f"{n} entries to the front (torch.cat([p[-{n}:], p[:-{n}]])).", stacklevel=2)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    # views into the flat parameter vector (no copies)
    def network_params(self) -> torch.Tensor:
        return self.params.detach()[: self.n_network_params]

    def grid_params(self) -> torch.Tensor:
        return self.params.detach()[self.n_network_params:]

    def forward(self, x01: torch.Tensor) -> torch.Tensor:
        if recording(x01, self.params):
            enc = grid_encode(x01.reshape(-1, 3), self.params[self.n_network_params:], self.grid.desc)
            return _mlp_nobias(enc, self.params, self.dims)[:, : self.n_output_dims]
        x01 = _C.f32c(x01.reshape(-1, 3))
        n = x01.shape[0]
        out = torch.empty((n, 16), dtype=torch.float32, device=x01.device)
        _C.check(_C.lib().qf_grid_mlp_forward(self.grid.desc, _C.ptr(self.grid_params()),
                                              _C.ptr(self.network_params()), _C.ptr(x01), n, _C.ptr(out),
                                              _C.stream()), "qf_grid_mlp_forward")
        return out[:, : self.n_output_dims]
