"""Oracle: instant-NGP style fields (TEST INFRASTRUCTURE, see oracle/__init__.py).

torch-CPU eager fp32 restatement of what the reference evaluates per quadrature point:

* tiny-cuda-nn ``HashGrid`` forward (un-vendored, unpinned git HEAD),
  configured at ``examples/radiance_fields/ngp.py:340-358,709-727`` and
  ``examples/field.py:157-171``;
* tiny-cuda-nn ``FullyFusedMLP`` (A.2) and ``SphericalHarmonics`` degree 4 (A.3),
  configured at ``ngp.py:325-338,693-746``;
* ``NGPRadianceField`` (``ngp.py:748-809``), ``NGPRadianceFieldSGNew``
  (``ngp.py:371-470``), ``BasicDecoder`` (``ngp.py:35-121``), and the deformation
  ``Field.density`` (``field.py:186-203``).

PARITY UNPINNED for the tiny-cuda-nn pieces: their source is not in the container and the
reference has no test vector for them.  tcnn computes in fp16 with fp16 accumulation; this
restatement (and the HIP kernels) use fp32 storage/accumulate, so the reference render itself
sits ~1e-3 relative away from both.
"""
import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

PRIME_Y = 2654435761
PRIME_Z = 805459861
U32 = 0xFFFFFFFF


@dataclass
class GridLevels:
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    scale: List[float]       # fp32 values, one per level
    resolution: List[int]
    offset: List[int]        # n_levels + 1 entries (in table rows, not floats)
    hashed: List[bool]

    @property
    def n_entries(self) -> int:
        return self.offset[-1]


def grid_levels(n_levels: int, log2_hashmap_size: int, base_resolution: int,
                per_level_scale: float, n_features: int = 2) -> GridLevels:
    """tcnn level rule (A.1): scale_l = exp2(l*log2(b))*N_min - 1 (fp32),
    res_l = ceil(scale_l)+1, rows = min(round_up(res^3, 8), 2^T)."""
    # every fp32 step evaluated in float64 and rounded once (libm-independent)
    log2_b = np.float32(np.log2(np.float64(np.float32(per_level_scale))))
    scales, ress, offs, hashed = [], [], [0], []
    for l in range(n_levels):
        arg = np.float32(np.float32(l) * log2_b)
        e = np.float32(np.exp2(np.float64(arg)))
        s = np.float32(np.float32(e * np.float32(base_resolution)) - np.float32(1.0))
        res = int(np.ceil(s)) + 1
        max_params = U32 // 2
        dense = max_params if float(res) ** 3 > max_params else res ** 3
        dense = (dense + 7) // 8 * 8
        rows = min(dense, 1 << log2_hashmap_size)
        # index rule: stride walks 1, res, res^2, res^3 while stride <= rows
        stride = 1
        for _ in range(3):
            if stride <= rows:
                stride = (stride * res) & U32
        scales.append(float(s))
        ress.append(res)
        hashed.append(rows < stride)
        offs.append(offs[-1] + rows)
    return GridLevels(n_levels, n_features, log2_hashmap_size, base_resolution,
                      float(per_level_scale), scales, ress, offs, hashed)


def _float_to_u32_grid(t: Tensor) -> Tensor:
    """floor()ed fp32 -> (uint32)(int32) with saturation, kept in an int64 container."""
    i = t.double().clamp(-2147483648.0, 2147483647.0).to(torch.int64)
    return i & U32


def hash_encode(x01: Tensor, table: Tensor, lv: GridLevels) -> Tensor:
    """x01 [N,3] fp32 (nominally in [0,1]), table [rows, F] fp32 -> [N, L*F] fp32, level-major."""
    n = x01.shape[0]
    out = torch.empty((n, lv.n_levels * lv.n_features), dtype=torch.float32)
    for l in range(lv.n_levels):
        scale, res = lv.scale[l], lv.resolution[l]
        rows = lv.offset[l + 1] - lv.offset[l]
        # pos = fmaf(scale, x, 0.5): product exact in float64, one rounding to fp32
        pos = (x01.double() * float(np.float32(scale)) + 0.5).float()
        g = torch.floor(pos)
        w = pos - g
        gi = _float_to_u32_grid(g)
        acc = torch.zeros((n, lv.n_features), dtype=torch.float32)
        for corner in range(8):
            weight = torch.ones(n, dtype=torch.float32)
            c = []
            for d in range(3):
                if corner & (1 << d):
                    weight = weight * w[:, d]
                    c.append((gi[:, d] + 1) & U32)
                else:
                    weight = weight * (1.0 - w[:, d])
                    c.append(gi[:, d])
            stride, idx = 1, torch.zeros(n, dtype=torch.int64)
            for d in range(3):
                if stride <= rows:
                    idx = (idx + c[d] * stride) & U32
                    stride = (stride * res) & U32
            if rows < stride:
                idx = (c[0] ^ ((c[1] * PRIME_Y) & U32) ^ ((c[2] * PRIME_Z) & U32)) & U32
            idx = idx % rows
            acc = acc + weight[:, None] * table[lv.offset[l] + idx]
        out[:, l * lv.n_features:(l + 1) * lv.n_features] = acc
    return out


def mlp_nobias(x: Tensor, weights: List[Tensor]) -> Tensor:
    """tcnn FullyFusedMLP shape: ReLU hidden layers, linear output, no biases (A.2)."""
    h = x
    for w in weights[:-1]:
        h = F.relu(F.linear(h, w))
    return F.linear(h, weights[-1])


def split_tcnn_mlp(params: Tensor, n_in_pad: int, n_hidden: int, n_neurons: int, n_out_pad: int):
    """Slice a flat tcnn network param vector into row-major [out,in] matrices (A.2)."""
    mats, o = [], 0
    dims = [n_in_pad] + [n_neurons] * n_hidden + [n_out_pad]
    for a, b in zip(dims[:-1], dims[1:]):
        mats.append(params[o:o + a * b].reshape(b, a))
        o += a * b
    return mats, o


def sh4(d: Tensor) -> Tensor:
    """Degree-4 real SH basis of a unit vector (x,y,z), 16 values (A.3)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ], dim=-1)


def normalize_to_aabb(x: Tensor, aabb: Tensor) -> Tuple[Tensor, Tensor]:
    """ngp.py:748-755 (bounded branch): returns (selector, x01); strict 0<x<1 (B-5)."""
    lo, hi = aabb[:3], aabb[3:]
    x01 = (x - lo) / (hi - lo)
    selector = ((x01 > 0.0) & (x01 < 1.0)).all(dim=-1)
    return selector, x01


@dataclass
class NGPWeights:
    """Everything the NGP-family fields need, in the reference's state-dict layout."""
    aabb: Tensor                      # [6]
    levels: GridLevels
    table: Tensor                     # [rows, 2] fp32
    base: List[Tensor]                # [64,32], [16,64]
    head_tcnn: Optional[List[Tensor]] = None   # NGPRadianceField: [64,32],[64,64],[16,64]
    head_layers: Optional[List[Tuple[Tensor, Tensor]]] = None  # SGNew BasicDecoder (W,b) incl. lout
    n_lobes: int = 0


def query_density(x: Tensor, wts: NGPWeights, return_feat: bool = False):
    """ngp.py:757-779 / 404-426.  density = exp(raw - 1) * selector (B-6)."""
    selector, x01 = normalize_to_aabb(x, wts.aabb)
    enc = hash_encode(x01.reshape(-1, 3), wts.table, wts.levels)
    out = mlp_nobias(enc, wts.base)
    raw, feat = out[:, :1], out[:, 1:16]
    density = torch.exp(raw - 1.0) * selector[:, None]
    return (density, feat) if return_feat else density


def ngp_forward(x: Tensor, d: Tensor, wts: NGPWeights):
    """NGPRadianceField.forward, ngp.py:781-809: SH4((d+1)/2 -> 2u-1) ++ geo15 ++ pad(1.0)."""
    density, feat = query_density(x, wts, return_feat=True)
    d01 = (d + 1.0) / 2.0
    sh = sh4(d01 * 2.0 - 1.0)
    h = torch.cat([sh, feat, torch.ones_like(feat[:, :1])], dim=-1)
    rgb = torch.sigmoid(mlp_nobias(h, wts.head_tcnn)[:, :3])
    return rgb, density


def basic_decoder(h: Tensor, layers: List[Tuple[Tensor, Tensor]]) -> Tensor:
    """ngp.py:93-121 with skip=[] and ReLU activation."""
    for w, b in layers[:-1]:
        h = F.relu(F.linear(h, w, b))
    w, b = layers[-1]
    return F.linear(h, w, b)


def spherical_gaussian_mixture(x: Tensor, direction: Tensor, n_lobes: int, discretize: bool = False) -> Tensor:
    """ngp.py:371-393: sum_l c_l * exp(|lambda_l| (a_l/|a_l| . d - 1)); ``discretize`` sends axis, sharpness and
    colour of every lobe through their uint8 codecs and back first (ngp.py:377-382)."""
    from . import quantize as q
    rgb = torch.zeros((x.shape[0], 3), dtype=x.dtype)
    for lobe in torch.chunk(x, n_lobes, dim=-1):
        axis = lobe[..., :3]
        axis = axis / torch.linalg.norm(axis, dim=-1, keepdim=True)
        lam = torch.abs(lobe[..., 3])
        c = lobe[..., 4:]
        if discretize:
            axis = q.inverse_of_azimuth_and_elevation(*q.compress_polar_coordinates(axis))
            lam = q.inverse_of_compressed_lambda(q.compress_lambda(lam))
            c = q.inverse_of_compressed_colors(q.compress_colors(c))
        rgb = rgb + c * torch.exp(lam * (torch.sum(axis * direction, -1) - 1))[..., None]
    return rgb


def features_to_rgb(features: Tensor, d: Tensor, n_lobes: int, discretize: bool = False) -> Tensor:
    """ngp.py:456-461; ``discretize`` also round-trips the diffuse colour (ngp.py:458-459)."""
    diffuse = features[:, :3]
    if discretize:
        from . import quantize as q
        diffuse = q.inverse_of_compressed_colors(q.compress_colors(diffuse))
    return torch.sigmoid(diffuse + spherical_gaussian_mixture(features[:, 3:], d, n_lobes, discretize))


def sg_features(x: Tensor, wts: NGPWeights) -> Tensor:
    """NGPRadianceFieldSGNew.features, ngp.py:445-454: [head(3+7L) | density]."""
    density, feat = query_density(x, wts, return_feat=True)
    return torch.cat([basic_decoder(feat, wts.head_layers), density], dim=-1)


def sg_forward(x: Tensor, d: Tensor, wts: NGPWeights):
    """NGPRadianceFieldSGNew.forward with use_viewdirs=False, ngp.py:428-443,463-470."""
    density, feat = query_density(x, wts, return_feat=True)
    f = basic_decoder(feat, wts.head_layers)
    return features_to_rgb(f, d, wts.n_lobes), density


@dataclass
class DeformWeights:
    """examples/field.py Field: grid + BasicDecoder(35 -> h -> h -> 1, ReLU, bias)."""
    scale: float
    levels: GridLevels
    table: Tensor
    layers: List[Tuple[Tensor, Tensor]]


def deform_field(x: Tensor, wts: DeformWeights) -> Tensor:
    """Field.density, field.py:186-203: cat[x01, grid(x01)] -> MLP -> [N,1]."""
    x01 = (x + wts.scale) / (2.0 * wts.scale)
    h = hash_encode(x01, wts.table, wts.levels)
    return basic_decoder(torch.cat([x01, h], dim=1), wts.layers)


def field_per_level_scale(max_res: int, scale: float, n_min: int, n_levels: int) -> float:
    """field.py:154."""
    return float(np.exp(np.log(max_res * scale / n_min) / (n_levels - 1)))


def ngp_per_level_scale(max_resolution: int, base_resolution: int, n_levels: int) -> float:
    """ngp.py:320-322 / 689-691."""
    return float(np.exp((np.log(max_resolution) - np.log(base_resolution)) / (n_levels - 1)))


# ---------------------------------------------------------------------------- bf16 mode (BASELINE config 3)
def bf16_round(t: Tensor) -> Tensor:
    """Round-to-nearest-even to bfloat16, returned as fp32 (what an fp32-accumulating bf16 MFMA consumes)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _mlp_nobias_bf16(x: Tensor, weights: List[Tensor]) -> Tensor:
    h = x
    for w in weights[:-1]:
        h = F.relu(F.linear(bf16_round(h), bf16_round(w)))
    return F.linear(bf16_round(h), bf16_round(weights[-1]))


def query_density_bf16(x: Tensor, wts: NGPWeights):
    """query_density with bf16 table / weights / inter-layer activations and fp32 accumulation."""
    selector, x01 = normalize_to_aabb(x, wts.aabb)
    enc = hash_encode(x01.reshape(-1, 3), bf16_round(wts.table), wts.levels)      # blend in fp32
    out = _mlp_nobias_bf16(enc, wts.base)
    raw, feat = out[:, :1], out[:, 1:16]
    return torch.exp(raw - 1.0) * selector[:, None], feat


def ngp_forward_bf16(x: Tensor, d: Tensor, wts: NGPWeights):
    density, feat = query_density_bf16(x, wts)
    sh = sh4(((d + 1.0) / 2.0) * 2.0 - 1.0)
    h = torch.cat([sh, feat, torch.ones_like(feat[:, :1])], dim=-1)
    return torch.sigmoid(_mlp_nobias_bf16(h, wts.head_tcnn)[:, :3]), density


def sg_forward_bf16(x: Tensor, d: Tensor, wts: NGPWeights):
    """SG head in bf16: w1/b1/w2/wout and the layer inputs are bf16, b2/bout stay fp32 (accumulator init)."""
    density, feat = query_density_bf16(x, wts)
    (w1, b1), (w2, b2), (wo, bo) = wts.head_layers
    h = F.relu(F.linear(bf16_round(feat), bf16_round(w1), bf16_round(b1)))
    h = F.relu(F.linear(bf16_round(h), bf16_round(w2), b2))
    f = F.linear(bf16_round(h), bf16_round(wo), bo)
    return features_to_rgb(f, d, wts.n_lobes), density
