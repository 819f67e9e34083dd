"""Oracle: uint8 texture codecs and baked-texture decode (TEST INFRASTRUCTURE).

torch-CPU restatement of

* the quantisers of ``examples/radiance_fields/ngp.py:210-281`` and
  ``examples/utils.py:54-63``;
* ``FeatureCompression.compress`` / ``get_features_from_texture_map`` /
  ``inverse_of_compressed_sigma`` (``examples/texture_utils.py:51-98,149-175``);
* the per-sample UV lookup of ``render_image_bake_texture_images_with_occgrid``
  (``examples/utils.py:1055-1063``) including trimesh 3.23.5
  ``triangles.points_to_barycentric`` method "cramer" in float64 (un-vendored,
  SURVEY.md A.6 -- parity unpinned for that one function).

Quirks reproduced: compress/inverse colour codecs test ``compress_type == "sigma"`` so
both script values ("sigmoid", "linear") take the linear +-12 branch (B-7); the azimuth
decode subtracts 128 from a uint8 tensor and wraps mod 256 (B-8); texture sigma decode
clips 1-a/255 at 1e-6 (B-9).
"""
import numpy as np
import torch
from torch import Tensor


def compress_sigma(sigma: Tensor) -> Tensor:
    """utils.py:54-58 / texture_utils.py:51-55."""
    alpha = 1 - torch.exp(-sigma * 0.005)
    return torch.clip(alpha * 255, 0, 255).to(torch.uint8)


def inverse_of_compressed_sigma_unclipped(alpha_u8: Tensor) -> Tensor:
    """utils.py:60-63 (no clip: a == 255 -> inf)."""
    a = alpha_u8.to(torch.float32) / 255.0
    return -torch.log(1 - a) / 0.005


def inverse_of_compressed_sigma(alpha_u8: Tensor) -> Tensor:
    """texture_utils.py:61-65 (clip at 1e-6)."""
    a = alpha_u8.to(torch.float32) / 255.0
    return -torch.log(torch.clip(1 - a, 1e-6)) / 0.005


def compress_polar_coordinates(vectors: Tensor):
    """ngp.py:239-243."""
    v = vectors / (torch.norm(vectors, dim=-1, keepdim=True) + 1e-6)
    azimuth = (torch.atan2(v[..., 1], v[..., 0]) * 128 / np.pi + 128).to(torch.uint8)
    elevation = (torch.acos(v[..., 2]) * 256 / np.pi).to(torch.uint8)
    return azimuth, elevation


def inverse_of_azimuth_and_elevation(azimuth_u8: Tensor, elevation_u8: Tensor) -> Tensor:
    """ngp.py:245-252 fed with uint8 tensors (texture_utils.py:161-167): the subtraction
    happens in uint8 and wraps."""
    az = (azimuth_u8 - 128) / 128 * np.pi
    el = elevation_u8 / 256 * np.pi
    return torch.stack([torch.cos(az) * torch.sin(el),
                        torch.sin(az) * torch.sin(el),
                        torch.cos(el)], dim=-1)


def compress_lambda(lambdas: Tensor, thres: float = 7.5) -> Tensor:
    """ngp.py:254-258."""
    log_l = torch.log(torch.clamp(lambdas, 1e-5, np.inf))
    c = torch.clamp((log_l + 2.5) / thres, 0.0, 1.0)
    return (255 * c).to(torch.uint8)


def inverse_of_compressed_lambda(c_u8: Tensor, thres: float = 7.5) -> Tensor:
    """ngp.py:260-262 (uint8 * python float -> float32)."""
    return torch.exp(c_u8 * thres / 255 - 2.5)


def compress_colors(colors: Tensor, thres: float = 12, compress_type: str = "sigma") -> Tensor:
    """ngp.py:264-273."""
    if compress_type == "sigma":
        c = torch.sigmoid(colors)
    else:
        c = (torch.clip(colors, -thres, thres) + thres) / 2 / thres
    return (c * 255).to(torch.uint8)


def inverse_of_compressed_colors(c_u8: Tensor, thres: float = 12, compress_type: str = "sigma") -> Tensor:
    """ngp.py:275-281."""
    c = c_u8.to(torch.float32) / 255.0
    if compress_type == "sigma":
        return torch.log(torch.clip(c / (1 - c), 1e-8, 1e37))
    return c * 2 * thres - thres


def compress_features(features: Tensor, n_lobes: int, compression_type: str, lambda_thres: float):
    """FeatureCompression.compress, texture_utils.py:67-98 -> dict of uint8 tensors."""
    n = features.shape[0]
    alpha = compress_sigma(features[:, -1])
    diffuse = compress_colors(features[..., :3], compress_type=compression_type)
    lobes = features[..., 3:-1].reshape(n, n_lobes, 7)
    az, el = compress_polar_coordinates(lobes[..., :3])
    lam = compress_lambda(torch.abs(lobes[..., 3]), lambda_thres)
    c = lobes[..., 4:]
    return {
        "alpha": alpha,
        "diffuse": diffuse,
        "lambdas": [torch.stack([lam[..., i], az[..., i], el[..., i]], dim=-1) for i in range(n_lobes)],
        "colors": [compress_colors(c[..., i, :], compress_type=compression_type) for i in range(n_lobes)],
    }


def features_from_texture_map(indices: Tensor, alpha: Tensor, diffuse: Tensor, sg_colors, lambdas,
                              compression_type: str, lambda_thres: float) -> Tensor:
    """get_features_from_texture_map, texture_utils.py:149-175.

    indices [S,2] int64 (row, col); alpha [T,T] u8; diffuse [T,T,3] u8; sg_colors[i],
    lambdas[i] [T,T,3] u8.  Returns [S, 3 + 7L + 1] fp32 = [diffuse | (axis3, lambda, colour3)*L | sigma]."""
    r, c = indices[:, 0], indices[:, 1]
    sigma = inverse_of_compressed_sigma(alpha[r, c])
    dif = inverse_of_compressed_colors(diffuse[r, c], compress_type=compression_type)
    n_lobes = len(sg_colors)
    feats = torch.zeros((indices.shape[0], 7 * n_lobes), dtype=torch.float32)
    for i in range(n_lobes):
        shared = lambdas[i][r, c]
        feats[:, 7 * i:7 * i + 3] = inverse_of_azimuth_and_elevation(shared[:, 1], shared[:, 2])
        feats[:, 7 * i + 3] = inverse_of_compressed_lambda(shared[:, 0], lambda_thres)
        feats[:, 7 * i + 4:7 * i + 7] = inverse_of_compressed_colors(sg_colors[i][r, c],
                                                                   compress_type=compression_type)
    return torch.cat([dif, feats, sigma.unsqueeze(1)], dim=-1)


def points_to_barycentric(triangles: np.ndarray, points: np.ndarray) -> np.ndarray:
    """trimesh.triangles.points_to_barycentric(method='cramer') in float64 (A.6).
    triangles [S,3,3], points [S,3] -> [S,3]."""
    tri = np.asarray(triangles, dtype=np.float64)
    p = np.asarray(points, dtype=np.float64)
    e0 = tri[:, 1] - tri[:, 0]
    e1 = tri[:, 2] - tri[:, 0]
    w = p - tri[:, 0]

    def dot(a, b):
        m = a * b
        return (m[:, 0] + m[:, 1]) + m[:, 2]

    d00, d01, d11 = dot(e0, e0), dot(e0, e1), dot(e1, e1)
    d02, d12 = dot(e0, w), dot(e1, w)
    inv = 1.0 / (d00 * d11 - d01 * d01)
    b2 = (d00 * d12 - d01 * d02) * inv
    b1 = (d11 * d02 - d01 * d12) * inv
    b0 = 1.0 - b1 - b2
    return np.stack([b0, b1, b2], axis=-1)


def texel_indices(tri_vertices: np.ndarray, points: np.ndarray, tri_uv: Tensor, texture_size: int) -> Tensor:
    """utils.py:1055-1063: barycentric (fp64) -> fp32 clamp [0,1] -> renormalise ->
    uv = sum_k uv_k * b_k (fp32, k = 0,1,2 in order) -> floor -> clip to [0, T-1] (int64).

    tri_vertices [S,3,3] float64, points [S,3] float32, tri_uv [S,3,2] fp32 (already scaled by T)."""
    b = torch.from_numpy(points_to_barycentric(tri_vertices, points).astype(np.float32))
    b = torch.clamp(b, 0, 1)
    b = b / ((b[:, 0] + b[:, 1]) + b[:, 2])[:, None]
    prod = tri_uv * b[..., None]
    uv = (prod[:, 0] + prod[:, 1]) + prod[:, 2]
    return torch.clip(torch.floor(uv).long(), 0, texture_size - 1)
