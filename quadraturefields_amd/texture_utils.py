"""Baked spherical-Gaussian textures: uint8 PNG set resident in HBM, decoded by the gfx950 texture kernels.

Mirrors ``FeatureCompression`` of ``examples/texture_utils.py:17-203``: same constructor, attributes
(``alpha``, ``diffuse``, ``sg_colors``, ``lambdas``, ``texture_size``) and methods.  The on-disk format is
the reference's: ``alpha.png`` [T,T], ``diffuse.png`` [T,T,3], ``color_{i}.png`` [T,T,3],
``lambda_axis_{i}.png`` [T,T,3] = (lambda, azimuth, elevation).  PNGs are read with Pillow (imageio is not
available here).
"""
import ctypes

import numpy as np
import torch

from . import _C
from .radiance_fields.ngp import (compress_colors, compress_lambda_torch, compress_polar_coordinates_torch)


def _read_png(path: str) -> np.ndarray:
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = 1000000000
    with Image.open(path) as im:
        return np.array(im)


def _write_png(path: str, arr: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(arr).save(path)


class FeatureCompression:
    def __init__(self, num_lobes, initialize=False, texture_size=None, path=None, compression_type="sigmoid",
                 lambda_thres=7.5, device="cuda:0"):
        self.num_lobes = num_lobes
        self.texture_size = texture_size
        self.compression_type = compression_type
        self.lambda_thres = lambda_thres
        self.device = _C.resolve_device(device)
        if not (1 <= num_lobes <= _C.QF_MAX_LOBES):
            raise ValueError(f"num_lobes must be in 1..{_C.QF_MAX_LOBES}")
        if initialize:
            t = texture_size
            self.alpha = torch.zeros((t, t), dtype=torch.uint8, device=self.device)
            self.diffuse = torch.zeros((t, t, 3), dtype=torch.uint8, device=self.device)
            self.sg_colors = {i: torch.zeros((t, t, 3), dtype=torch.uint8, device=self.device) for i in range(num_lobes)}
            self.lambdas = {i: torch.zeros((t, t, 3), dtype=torch.uint8, device=self.device) for i in range(num_lobes)}
        else:
            self.alpha = torch.from_numpy(_read_png(path + "alpha.png")).to(self.device)
            self.diffuse = torch.from_numpy(_read_png(path + "diffuse.png")).to(self.device)
            self.sg_colors = {i: torch.from_numpy(_read_png(path + "color_{}.png".format(i))).to(self.device)
                              for i in range(num_lobes)}
            self.lambdas = {i: torch.from_numpy(_read_png(path + "lambda_axis_{}.png".format(i))).to(self.device)
                            for i in range(num_lobes)}
            if self.texture_size is None:
                self.texture_size = int(self.alpha.shape[0])

    @classmethod
    def from_arrays(cls, alpha, diffuse, sg_colors, lambdas, compression_type="sigmoid", lambda_thres=7.5,
                    device="cuda:0"):
        """Build from in-memory uint8 arrays (tests / synthetic scenes)."""
        self = cls(len(sg_colors), initialize=True, texture_size=1, compression_type=compression_type,
                   lambda_thres=lambda_thres, device=device)
        to = lambda a: torch.as_tensor(a, dtype=torch.uint8).to(self.device).contiguous()
        self.alpha, self.diffuse = to(alpha), to(diffuse)
        self.sg_colors = {i: to(a) for i, a in enumerate(sg_colors)}
        self.lambdas = {i: to(a) for i, a in enumerate(lambdas)}
        self.texture_size = int(self.alpha.shape[0])
        return self

    # ------------------------------------------------------------------ encode side (texture_utils.py:51-124)
    def compress_sigma(self, sigma):
        alpha = 1 - torch.exp(-sigma * 0.005)
        return torch.clip(alpha * 255, 0, 255).to(torch.uint8)

    def inverse_of_compressed_sigma(self, alpha):
        alpha = alpha.to(torch.float32) / 255.0
        return -torch.log(torch.clip(1 - alpha, 1e-6)) / 0.005

    def compress(self, features):
        n = features.shape[0]
        alpha = self.compress_sigma(features[:, -1])
        diffuse = compress_colors(features[..., :3], compress_type=self.compression_type)
        lobes = torch.reshape(features[..., 3:-1], (n, self.num_lobes, 7))
        azimuth, elevation = compress_polar_coordinates_torch(lobes[..., :3])
        compressed_lambda = compress_lambda_torch(torch.abs(lobes[..., 3]), self.lambda_thres)
        c = lobes[..., 4:]
        data = {"alpha": alpha, "diffuse": diffuse, "lambdas": [], "colors": []}
        for i in range(self.num_lobes):
            data["lambdas"].append(torch.stack([compressed_lambda[..., i], azimuth[..., i], elevation[..., i]], axis=-1))
            data["colors"].append(compress_colors(c[..., i, :], compress_type=self.compression_type))
        return data

    def load_features_into_maps(self, features, indices):
        data = self.compress(features)
        self.alpha[indices[:, 0], indices[:, 1]] = data["alpha"]
        self.diffuse[indices[:, 0], indices[:, 1]] = data["diffuse"]
        for i in range(self.num_lobes):
            self.lambdas[i][indices[:, 0], indices[:, 1]] = data["lambdas"][i]
            self.sg_colors[i][indices[:, 0], indices[:, 1]] = data["colors"][i]

    assign_values_to_texture_map = load_features_into_maps

    def save_to_file(self, path):
        _write_png(path + "alpha.png", self.alpha.cpu().numpy())
        _write_png(path + "diffuse.png", self.diffuse.cpu().numpy())
        for i in range(self.num_lobes):
            _write_png(path + "color_{}.png".format(i), self.sg_colors[i].cpu().numpy())
            _write_png(path + "lambda_axis_{}.png".format(i), self.lambdas[i].cpu().numpy())

    def compress_features_and_save(self, features, path):
        """[N, N, 3+7L+1] float features -> the PNG set, without touching the maps (texture_utils.py:108-117)."""
        n = features.shape[0]
        data = self.compress(features.reshape((n * n, -1)))
        for i in range(self.num_lobes):
            _write_png(path + "color_{}.png".format(i), data["colors"][i].reshape((n, n, 3)).cpu().numpy())
            _write_png(path + "lambda_axis_{}.png".format(i), data["lambdas"][i].reshape((n, n, 3)).cpu().numpy())
        _write_png(path + "alpha.png", data["alpha"].cpu().numpy().reshape((n, n)))
        _write_png(path + "diffuse.png", data["diffuse"].reshape((n, n, 3)).cpu().numpy())

    @staticmethod
    def sigma_to_alpha(sigma):
        return 1 - np.exp(-sigma * 0.005)

    # ------------------------------------------------------------------ shading of fetched features (:126-147)
    def spherical_gaussian(self, x, direction):
        axis = x[..., :3]
        axis = axis / torch.linalg.norm(axis, dim=-1, keepdim=True)
        return x[..., 4:] * torch.exp(torch.abs(x[..., 3]) * (torch.sum(axis * direction, -1) - 1))[..., None]

    def spherical_gaussian_mixture(self, x, direction):
        rgb = torch.zeros((x.shape[0], 3), dtype=x.dtype, device=x.device)
        for x_ in torch.chunk(x, self.num_lobes, dim=-1):
            rgb = rgb + self.spherical_gaussian(x_, direction)
        return rgb

    def features_to_rgb(self, features, dir):
        """sigmoid(diffuse + SG mixture) of ``get_features_from_texture_map`` rows (extra columns ignored): the
        two-call form of ``shade``, one HIP launch (``qf_sg_features_to_rgb``)."""
        features = _C.f32c(features)
        dir = _C.f32c(dir.reshape(-1, 3))
        n = features.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=features.device)
        _C.check(_C.lib().qf_sg_features_to_rgb(_C.ptr(features), features.shape[1], _C.ptr(dir), n, self.num_lobes,
                                                _C.ptr(rgb), _C.stream()), "qf_sg_features_to_rgb")
        return rgb

    # ------------------------------------------------------------------ decode side (texture_utils.py:149-175)
    def texture_set(self) -> _C.TextureSet:
        t = _C.TextureSet()
        self.alpha, self.diffuse = self.alpha.contiguous(), self.diffuse.contiguous()
        t.alpha, t.diffuse = self.alpha.data_ptr(), self.diffuse.data_ptr()
        for i in range(self.num_lobes):
            self.sg_colors[i], self.lambdas[i] = self.sg_colors[i].contiguous(), self.lambdas[i].contiguous()
            t.colors[i] = self.sg_colors[i].data_ptr()
            t.lambda_axis[i] = self.lambdas[i].data_ptr()
        t.texture_size = int(self.alpha.shape[0])
        t.n_lobes = self.num_lobes
        t.sigmoid_codec = 1 if self.compression_type == "sigma" else 0   # the reference's string test (B-7)
        t.lambda_thres = float(self.lambda_thres)
        return t

    def get_features_from_texture_map(self, indices):
        """indices [S,2] int64 (row, col) -> [S, 3+7L+1] = [diffuse | (axis3, lambda, colour3)*L | sigma]."""
        indices = _C.i64c(indices)
        n = indices.shape[0]
        out = torch.empty((n, 3 + 7 * self.num_lobes + 1), dtype=torch.float32, device=indices.device)
        t = self.texture_set()
        _C.check(_C.lib().qf_texture_fetch(ctypes.byref(t), _C.ptr(indices), n, _C.ptr(out), _C.stream()),
                 "qf_texture_fetch")
        return out

    def records(self) -> torch.Tensor:
        """The device-resident form of the texture set: one 64-byte record per texel (all of a texel's quantised
        bytes in one sector), built from the planes on first use and rebuilt when a plane is modified in place or
        replaced.  [T*T, 64] uint8."""
        planes = [self.alpha, self.diffuse] + [self.sg_colors[i] for i in range(self.num_lobes)] + \
                 [self.lambdas[i] for i in range(self.num_lobes)]
        key = tuple((p.data_ptr(), p._version) for p in planes)
        cache = getattr(self, "_records", None)
        if cache is None or cache[0] != key:
            t = self.texture_set()
            size = int(self.alpha.shape[0])
            rec = torch.empty((size * size, _C.QF_TEXEL_RECORD_BYTES), dtype=torch.uint8, device=self.alpha.device)
            _C.check(_C.lib().qf_texture_pack(ctypes.byref(t), _C.ptr(rec), _C.stream()), "qf_texture_pack")
            planes = [self.alpha, self.diffuse] + [self.sg_colors[i] for i in range(self.num_lobes)] + \
                     [self.lambdas[i] for i in range(self.num_lobes)]      # texture_set() may have made them contiguous
            cache = self._records = (tuple((p.data_ptr(), p._version) for p in planes), rec)
        return cache[1]

    def shade(self, indices, dirs, packed: bool = True):
        """Fused fetch + dequantise + spherical-Gaussian shading: (rgb [S,3], sigma [S]).  ``packed`` reads the
        interleaved texel records (one sector per sample); ``packed=False`` reads the reference's planes.  Same bits."""
        indices = _C.i64c(indices)
        dirs = _C.f32c(dirs)
        n = indices.shape[0]
        rgb = torch.empty((n, 3), dtype=torch.float32, device=indices.device)
        sigma = torch.empty((n,), dtype=torch.float32, device=indices.device)
        if packed:
            _C.check(_C.lib().qf_texture_shade_packed(
                _C.ptr(self.records()), int(self.alpha.shape[0]), self.num_lobes,
                1 if self.compression_type == "sigma" else 0, float(self.lambda_thres), _C.ptr(indices), _C.ptr(dirs), n,
                _C.ptr(rgb), _C.ptr(sigma), _C.stream()), "qf_texture_shade_packed")
            return rgb, sigma
        t = self.texture_set()
        _C.check(_C.lib().qf_texture_shade(ctypes.byref(t), _C.ptr(indices), _C.ptr(dirs), n, _C.ptr(rgb),
                                           _C.ptr(sigma), _C.stream()), "qf_texture_shade")
        return rgb, sigma
