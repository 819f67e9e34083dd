"""Oracle: ray generation, mesh quadrature points and the two image renderers
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates on numpy / torch-CPU:

* ``SubjectLoader.fetch_data`` ray generation (``examples/datasets/nerf_synthetic.py:289-378``);
* ``MeshIntersection.sampling_raytrace_numpy`` / ``sampling_indexing`` / ``find_deltas``
  (``examples/mesh_utils.py:225-231,343-412``) on top of the brute-force intersector in
  ``intersect_ref.c``;
* ``generate_splits`` (``examples/train_finetune.py:419-439``);
* ``render_image_finetune_with_occgrid`` (``examples/utils.py:465-607``) and
  ``render_image_bake_texture_images_with_occgrid`` (``examples/utils.py:998-1095``).
"""
import ctypes
import os
import subprocess
from typing import Optional

import numpy as np
import torch

from . import fields, quantize, volrend

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def host_cores(cap: int = 16) -> int:
    """Usable host cores: affinity mask and cgroup quota respected, capped (a 1-GPU box shares its host)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def build() -> str:
    """Compile intersect_ref.c with gcc (oracle/Makefile) and return the .so path."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "_build", "libqf_oracle.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libqf_oracle.so")
        src = os.path.join(_HERE, "intersect_ref.c")
        if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.qf_oracle_multihit.restype = ctypes.c_int
        _LIB.qf_oracle_multihit.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
            ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _LIB.qf_oracle_bvh_build.restype = ctypes.c_void_p
        _LIB.qf_oracle_bvh_build.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        _LIB.qf_oracle_bvh_free.restype = None
        _LIB.qf_oracle_bvh_free.argtypes = [ctypes.c_void_p]
        _LIB.qf_oracle_bvh_multihit.restype = ctypes.c_int
        _LIB.qf_oracle_bvh_multihit.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def trimesh_ray_offset(vertices) -> float:
    """trimesh 3.23.5 ``ray_pyembree.RayMeshIntersector.intersects_id``: after every hit the ray restarts
    ``clip(_ray_offset_factor * _scale, _ray_offset_floor, inf)`` past it, ``_ray_offset_factor = 1e-4``,
    ``_ray_offset_floor = 1e-8``, ``_scale = 100 / mesh.scale`` (scale_to_box), ``mesh.scale`` = length of the
    bounding-box diagonal (restated from memory: PARITY UNPINNED)."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    if v.shape[0] == 0:
        return 1e-8
    scale = float(np.sqrt(((v.max(axis=0) - v.min(axis=0)) ** 2).sum()))
    if not scale > 0:
        return 1e-8
    return float(np.clip(1e-4 * (100.0 / scale), 1e-8, np.inf))


# ------------------------------------------------------------------ ray generation
def generate_rays(c2w: torch.Tensor, focal: float, width: int, height: int, opengl: bool = True):
    """Full-image rays, nerf_synthetic.py:310-373 (eval branch): pixel centres (+0.5),
    OpenGL -y/-z, normalised directions.  c2w [3,4] or [4,4] fp32.  Returns (origins, viewdirs) [H*W,3]."""
    c2w = c2w.to(torch.float32)
    K = torch.tensor([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], dtype=torch.float32)
    x, y = torch.meshgrid(torch.arange(width), torch.arange(height), indexing="xy")
    x, y = x.flatten(), y.flatten()
    sgn = -1.0 if opengl else 1.0
    cam = torch.nn.functional.pad(
        torch.stack([(x - K[0, 2] + 0.5) / K[0, 0], (y - K[1, 2] + 0.5) / K[1, 1] * sgn], dim=-1),
        (0, 1), value=sgn)
    directions = (cam[:, None, :] * c2w[None, :3, :3]).sum(dim=-1)
    origins = torch.broadcast_to(c2w[:3, -1], directions.shape)
    viewdirs = directions / torch.linalg.norm(directions, dim=-1, keepdims=True)
    return origins.reshape(-1, 3).contiguous(), viewdirs.reshape(-1, 3).contiguous()


# --------------------------------------------------------------------- intersector
class BruteForceIntersector:
    """Duck-types the trimesh intersector used at mesh_utils.py:350-354.  ``min_separation``: trimesh's re-origin
    distance -- ``"trimesh"`` (default: ``trimesh_ray_offset(vertices)``), a distance, or 0 / None for every hit."""

    def __init__(self, vertices: np.ndarray, faces: np.ndarray, min_separation="trimesh"):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.faces = np.ascontiguousarray(faces, dtype=np.int64)
        self.tri = np.ascontiguousarray(self.vertices.astype(np.float32)[self.faces].reshape(-1, 9))
        if isinstance(min_separation, str):
            assert min_separation == "trimesh"
            min_separation = trimesh_ray_offset(self.vertices)
        self.min_separation = max(float(min_separation or 0.0), 0.0)

    def _run(self, o, d, n, max_hits, n_threads, tri, t, cnt):
        return _lib().qf_oracle_multihit(
            self.tri.ctypes.data, self.tri.shape[0], o.ctypes.data, d.ctypes.data, n, int(max_hits),
            float(self.min_separation), int(n_threads or host_cores()), tri.ctypes.data, t.ctypes.data, cnt.ctypes.data)

    def hits(self, origins, vectors, max_hits, n_threads=None):
        """-> (tri [R,K] int32 (-1 pad), t [R,K] fp32 (+inf pad), count [R] int32)."""
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(vectors, dtype=np.float32)
        n = o.shape[0]
        tri = np.empty((n, max_hits), dtype=np.int32)
        t = np.empty((n, max_hits), dtype=np.float32)
        cnt = np.empty(n, dtype=np.int32)
        rc = self._run(o, d, n, max_hits, n_threads, tri, t, cnt)
        if rc != 0:
            raise RuntimeError("oracle multihit failed: %d" % rc)
        return tri, t, cnt

    def intersects_id(self, origins, vectors, multiple_hits=True, return_locations=True, max_hits=10):
        """(index_tri[S], index_ray[S], locations[S,3] float64), pass-major like trimesh:
        all first hits, then all second hits, ...  location = o + t*d in float64."""
        tri, t, cnt = self.hits(origins, vectors, max_hits)
        o64 = np.asarray(origins, dtype=np.float32).astype(np.float64)
        d64 = np.asarray(vectors, dtype=np.float32).astype(np.float64)
        out_t, out_r, out_p = [], [], []
        for k in range(max_hits):
            rays = np.nonzero(cnt > k)[0]
            if rays.size == 0:
                break
            tk = t[rays, k].astype(np.float64)
            out_t.append(tri[rays, k].astype(np.int64))
            out_r.append(rays.astype(np.int64))
            out_p.append(o64[rays] + tk[:, None] * d64[rays])
        if not out_t:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 3), np.float64)
        return np.concatenate(out_t), np.concatenate(out_r), np.concatenate(out_p)


class BVHIntersector(BruteForceIntersector):
    """The same answer as the brute force (tested), through the oracle's host BVH walk (intersect_ref.c): what the
    reference's CPU path does with Embree, and the intersection leg of bench.py's cpu_baseline."""

    def __init__(self, vertices, faces, min_separation="trimesh"):
        super().__init__(vertices, faces, min_separation)
        self._bvh = _lib().qf_oracle_bvh_build(self.tri.ctypes.data, self.tri.shape[0])
        if not self._bvh:
            raise RuntimeError("qf_oracle_bvh_build failed")

    def __del__(self):
        if getattr(self, "_bvh", None):
            try:
                _lib().qf_oracle_bvh_free(self._bvh)
            except Exception:           # interpreter shutdown
                pass
            self._bvh = None

    def _run(self, o, d, n, max_hits, n_threads, tri, t, cnt):
        return _lib().qf_oracle_bvh_multihit(
            self._bvh, o.ctypes.data, d.ctypes.data, n, int(max_hits), float(self.min_separation),
            int(n_threads or host_cores()), tri.ctypes.data, t.ctypes.data, cnt.ctypes.data)


def sampling_raytrace_numpy(intersector, vectors: np.ndarray, origins: np.ndarray, max_hits: int):
    """mesh_utils.py:343-387.  Returns (points, dirs, index_ray, depth, index_tri, 0, origins)
    or None when nothing is hit (:357-358).  Sorts are stable here (the reference's first
    argsort is numpy's default introsort; ties between equal ray ids are then resolved by the
    lexsort on depth, and equal depths keep the intersector's pass order)."""
    index_tri, index_ray, points = intersector.intersects_id(
        origins, vectors, multiple_hits=True, return_locations=True, max_hits=max_hits)
    if index_tri.shape[0] == 0:
        return None
    order = np.argsort(index_ray, kind="stable")
    index_tri, index_ray, points = index_tri[order], index_ray[order], points[order]
    vectors = vectors[index_ray]
    origins = origins[index_ray]
    norm = np.linalg.norm(vectors, axis=1) + 1e-7
    vectors = vectors / norm[:, None]
    depth = np.linalg.norm(points - origins, axis=1)
    order = np.lexsort((depth, index_ray))
    return (points[order], vectors[order], index_ray[order], depth[order], index_tri[order], 0, origins)


def to_loader_tensors(sample):
    """The casts of nerf_synthetic.py:256-257."""
    xyzs, dirs, index_ray, ts, index_tri, _, origins = sample
    return [torch.from_numpy(xyzs.astype(np.float32)), torch.from_numpy(dirs.astype(np.float32)),
            torch.from_numpy(index_ray.astype(np.int64)), torch.from_numpy(ts.astype(np.float32)),
            torch.from_numpy(index_tri.astype(np.int64)), torch.from_numpy(origins.astype(np.float32))]


def sampling_indexing(points, origins, vectors, index_ray, depth, index_tri, render_step_size=0.005):
    """mesh_utils.py:389-412: lexsort by (ray, depth), boundaries, constant deltas (B-4)."""
    order = torch.from_numpy(np.lexsort((depth.detach().numpy(), index_ray.numpy())))
    index_tri, index_ray = index_tri[order], index_ray[order]
    points, depth, origins, vectors = points[order], depth[order], origins[order], vectors[order]
    boundary = volrend.mark_pack_boundaries(index_ray)
    deltas = torch.full((depth.shape[0],), render_step_size, dtype=torch.float32)
    return points, deltas, boundary, vectors, index_ray, depth, index_tri, origins


def generate_splits(data, num_rays, chunk_size=160000):
    """train_finetune.py:419-439."""
    xyzs, dirs, index_ray, ts, index_tri, origins = data
    chunks = []
    for i in range(0, num_rays, chunk_size):
        mask = (index_ray < i + chunk_size) & (index_ray >= i)
        if mask.sum() == 0:
            continue
        chunks.append(tuple(t[mask].contiguous() for t in (xyzs, dirs, index_ray, ts, index_tri, origins)))
    return chunks


# ------------------------------------------------------------------------ renderers
def render_image_finetune(ngp: fields.NGPWeights, deform: Optional[fields.DeformWeights], data,
                          n_rays: int, scaling: float = 0.0, bg_color: str = "white",
                          render_bkgd=None, render_step_size: float = 0.005, sg: bool = False):
    """utils.py:538-607 for one split.  Returns (rgb[N,3], alpha[N,1], depth[N,1], n_samples,
    weights[S,1], points, index_ray, index_tri).  The regularisation loss and the random
    barycentric vertex sample (:543-546,583) do not touch the image and are omitted."""
    xyzs, dirs, index_ray, ts, index_tri, origins = data
    if deform is not None and scaling != 0.0:
        f = fields.deform_field(xyzs, deform)                       # [S,1]
        del_vector = torch.tanh(f.expand(-1, 3)) * scaling          # broadcast into 3 (B-15)
        del_delta = (del_vector * dirs).sum(-1, keepdim=True)
        xyzs = xyzs + del_delta * dirs
        ts = ts + del_delta.view(-1)
    points, deltas, boundary, dirs, index_ray, depth, index_tri, _ = sampling_indexing(
        xyzs, origins, dirs, index_ray, ts, index_tri, render_step_size)
    if sg:
        rgbs, sigmas = fields.sg_forward(points, dirs, ngp)
    else:
        rgbs, sigmas = fields.ngp_forward(points, dirs, ngp)
    rgb, alpha, _, dep, weights = volrend.derive_properties(
        rgbs, sigmas.squeeze(-1), depth, deltas, boundary, index_ray,
        bg_color=bg_color, render_bkgd=render_bkgd, N=n_rays)
    return rgb, alpha, dep, xyzs.shape[0], weights, points, index_ray, index_tri


def barycentric_vertex_samples(tri_vertices: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """utils.py:543-546: a random point of each sample's triangle, weights w [S,3] ~ U(0,1) (not normalised
    beforehand): sum(v * w) / (sum(w) + 1e-6).  tri_vertices [S,3,3]."""
    w = w[..., None]
    return torch.sum(tri_vertices * w, dim=1) / (torch.sum(w, dim=1) + 1e-6)


def finetune_regulariser(deform: fields.DeformWeights, xyzs: torch.Tensor, vertex_points: torch.Tensor,
                         scaling: float) -> torch.Tensor:
    """utils.py:549-566,583: mean(dv^2) + mean((dv_vertex - dv.detach())^2), dv = tanh(field)*scaling broadcast
    into 3 components (B-15).  Training only; differentiable w.r.t. the deformation field's weights."""
    dv_v = torch.tanh(fields.deform_field(vertex_points, deform)).expand(-1, 3) * scaling
    dv = torch.tanh(fields.deform_field(xyzs, deform)).expand(-1, 3) * scaling
    return ((dv ** 2).mean() + ((dv_v - dv.detach()) ** 2).mean()).reshape(1)


def render_image_bake_texture(data, n_rays: int, vertices64: np.ndarray, faces: np.ndarray,
                              uv_scaled: torch.Tensor, textures: dict, n_lobes: int,
                              compression_type: str, lambda_thres: float, bg_color: str = "white",
                              render_step_size: float = 0.005):
    """utils.py:1041-1095.  textures = {"alpha","diffuse","colors":[...],"lambdas":[...]} (uint8)."""
    xyzs, dirs, index_ray, ts, index_tri, origins = data
    points, deltas, boundary, dirs, index_ray, depth, index_tri, _ = sampling_indexing(
        xyzs, origins, dirs, index_ray, ts, index_tri, render_step_size)
    f = faces[index_tri.numpy()]
    size = textures["alpha"].shape[0]
    uv_pts = quantize.texel_indices(vertices64[f], points.numpy(), uv_scaled[torch.from_numpy(f)], size)
    tex = quantize.features_from_texture_map(
        uv_pts, textures["alpha"], textures["diffuse"], textures["colors"], textures["lambdas"],
        compression_type, lambda_thres)
    sigmas, feats = tex[:, -1], tex[:, :-1]
    rgbs = fields.features_to_rgb(feats, dirs, n_lobes)
    rgb, alpha, _, dep, weights = volrend.derive_properties(
        rgbs, sigmas, depth, deltas, boundary, index_ray, bg_color=bg_color, render_bkgd=None, N=n_rays)
    return rgb, alpha, dep, xyzs.shape[0], weights, points, uv_pts


def area_downsample(img: torch.Tensor, factor: int) -> torch.Tensor:
    """cv2.resize(..., INTER_AREA) by an integer factor == box average (B-13). img [H,W,C]."""
    if factor == 1:
        return img
    h, w, c = img.shape
    return img.reshape(h // factor, factor, w // factor, factor, c).mean(dim=(1, 3))


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """train_finetune.py:631-632."""
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(-10.0 * torch.log(mse) / np.log(10.0)) if mse > 0 else float("inf")
