"""Deterministic synthetic stand-ins for the assets the reference needs but the container lacks
(SURVEY.md section 8d): Blender-convention cameras, a nested-shell quadrature mesh, seeded field weights and
uint8 SG textures.  Pure numpy / torch-CPU data generation -- no arithmetic of the render path lives here.
"""
import math
from typing import Tuple

import numpy as np
import torch

from .mesh_io import TriMesh

LEGO_CAMERA_ANGLE_X = 0.6911112070083618   # transforms_*.json of the NeRF-synthetic "lego" scene
LEGO_RADIUS = 4.031128874


def lego_focal(width: int = 800) -> float:
    """focal = 0.5 w / tan(0.5 camera_angle_x) (nerf_synthetic.py:102) -> 1111.11 at w = 800."""
    return 0.5 * width / math.tan(0.5 * LEGO_CAMERA_ANGLE_X)


def orbit_cameras(n: int, radius: float = LEGO_RADIUS, seed: int = 42) -> torch.Tensor:
    """[n,3,4] camera-to-world matrices on the upper hemisphere looking at the origin (OpenGL: -z forward)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        az = rng.uniform(0, 2 * np.pi)
        el = rng.uniform(np.deg2rad(10), np.deg2rad(70))
        pos = radius * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        back = pos / np.linalg.norm(pos)
        right = np.cross(np.array([0.0, 0.0, 1.0]), back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        out[i, :, 0], out[i, :, 1], out[i, :, 2], out[i, :, 3] = right, up, back, pos
    return torch.from_numpy(out)


def camera_rays(c2w: torch.Tensor, focal: float, width: int, height: int, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """Full-image rays with the arithmetic of SubjectLoader.fetch_data (nerf_synthetic.py:310-373): pixel
    centres, OpenGL axes, normalised directions; [H*W,3] origins and viewdirs, row-major."""
    c2w = c2w.to(device=device, dtype=torch.float32)
    K = torch.tensor([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], dtype=torch.float32, device=device)
    x, y = torch.meshgrid(torch.arange(width, device=device), torch.arange(height, device=device), indexing="xy")
    x, y = x.flatten(), y.flatten()
    camera_dirs = torch.nn.functional.pad(
        torch.stack([(x - K[0, 2] + 0.5) / K[0, 0], (y - K[1, 2] + 0.5) / K[1, 1] * -1.0], dim=-1), (0, 1), value=-1.0)
    directions = (camera_dirs[:, None, :] * c2w[None, :3, :3]).sum(dim=-1)
    origins = torch.broadcast_to(c2w[:3, -1], directions.shape)
    viewdirs = directions / torch.linalg.norm(directions, dim=-1, keepdims=True)
    return origins.reshape(-1, 3).contiguous(), viewdirs.reshape(-1, 3).contiguous()


def icosphere(subdivisions: int) -> Tuple[np.ndarray, np.ndarray]:
    """Unit icosphere: 10*4^s + 2 vertices, 20*4^s faces."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdivisions):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        edges.sort(axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    return v, f


def shell_mesh(n_shells: int = 12, subdivisions: int = 6, r_min: float = 0.3, r_max: float = 1.2,
               noise: float = 0.03, seed: int = 42) -> TriMesh:
    """Nested noise-displaced icospheres (stand-in for the marching-cubes quadrature surfaces), with per-vertex
    UVs: shell i owns one cell of a 4-column atlas, (u,v) = (azimuth, elevation) inside the cell."""
    rng = np.random.default_rng(seed)
    sv, sf = icosphere(subdivisions)
    freqs = rng.normal(size=(6, 3)) * 3.0
    phases = rng.uniform(0, 2 * np.pi, size=6)
    amps = rng.uniform(0.3, 1.0, size=6)
    verts, faces, uvs = [], [], []
    cols = 4
    rows = (n_shells + cols - 1) // cols
    for i in range(n_shells):
        r = r_min + (r_max - r_min) * (i / max(n_shells - 1, 1))
        disp = sum(a * np.sin(sv @ fr * (1 + 0.15 * i) + ph) for a, fr, ph in zip(amps, freqs, phases)) / amps.sum()
        verts.append(sv * (r * (1.0 + noise * disp))[:, None])
        faces.append(sf + i * len(sv))
        az = (np.arctan2(sv[:, 1], sv[:, 0]) + np.pi) / (2 * np.pi)
        el = np.arccos(np.clip(sv[:, 2], -1, 1)) / np.pi
        cu, cv = i % cols, i // cols
        uvs.append(np.stack([(cu + 0.02 + 0.96 * az) / cols, (cv + 0.02 + 0.96 * el) / rows], axis=1))
    return TriMesh(np.concatenate(verts), np.concatenate(faces), np.concatenate(uvs))


def seeded_ngp_state(log2_hashmap_size: int, n_rows: int, seed: int = 42, sg_lobes: int = 0,
                     table_amp: float = 0.5, density_gain: float = 6.0):
    """Random-init weights in the reference's state-dict layout (there is no checkpoint to load).  The table
    amplitude and the gain on the density row are picked so densities span roughly [0, 200] at delta = 0.005
    (alpha from ~0 to ~0.6 per quadrature point) instead of the near-zero field tcnn's default init gives."""
    g = torch.Generator().manual_seed(seed)

    def xavier(out_d, in_d, gain=1.0):
        b = gain * math.sqrt(6.0 / (in_d + out_d))
        return (torch.rand(out_d, in_d, generator=g) * 2 - 1) * b

    w1 = xavier(64, 32, 2.0)
    w2 = xavier(16, 64, 1.5)
    w2[0] *= density_gain
    table = (torch.rand(n_rows * 2, generator=g) * 2 - 1) * table_amp
    state = {"mlp_base.params": torch.cat([w1.flatten(), w2.flatten(), table])}
    if sg_lobes == 0:
        state["mlp_head.params"] = torch.cat([xavier(64, 32, 1.5).flatten(), xavier(64, 64, 1.5).flatten(),
                                              xavier(16, 64, 2.0).flatten()])
    else:
        n_out = 3 + 7 * sg_lobes
        state["mlp_head.layers.0.weight"] = xavier(64, 15, 1.5)
        state["mlp_head.layers.0.bias"] = (torch.rand(64, generator=g) - 0.5) * 0.2
        state["mlp_head.layers.1.weight"] = xavier(64, 64, 1.5)
        state["mlp_head.layers.1.bias"] = (torch.rand(64, generator=g) - 0.5) * 0.2
        state["mlp_head.lout.weight"] = xavier(n_out, 64, 2.0)
        state["mlp_head.lout.bias"] = (torch.rand(n_out, generator=g) - 0.5) * 0.5
    return state


def seeded_deform_state(n_params_grid: int, seed: int = 7, table_amp: float = 0.5):
    g = torch.Generator().manual_seed(seed)

    def xavier(out_d, in_d, gain=1.0):
        b = gain * math.sqrt(6.0 / (in_d + out_d))
        return (torch.rand(out_d, in_d, generator=g) * 2 - 1) * b

    return {
        "xyz_encoder.params": (torch.rand(n_params_grid, generator=g) * 2 - 1) * table_amp,
        "decoder_field.layers.0.weight": xavier(32, 35, 1.5), "decoder_field.layers.0.bias": (torch.rand(32, generator=g) - 0.5) * 0.2,
        "decoder_field.layers.1.weight": xavier(32, 32, 1.5), "decoder_field.layers.1.bias": (torch.rand(32, generator=g) - 0.5) * 0.2,
        "decoder_field.lout.weight": xavier(1, 32, 2.0), "decoder_field.lout.bias": (torch.rand(1, generator=g) - 0.5) * 0.2,
    }


def random_textures(texture_size: int, n_lobes: int, seed: int = 42, fill: float = 0.85):
    """uint8 texture set, uniformly random per channel; alpha = 0 outside a random chart mask."""
    rng = np.random.default_rng(seed)
    t = texture_size
    coarse = rng.random((max(t // 16, 1), max(t // 16, 1))) < fill
    mask = np.kron(coarse, np.ones((16, 16), dtype=bool))[:t, :t]
    alpha = (rng.integers(1, 256, size=(t, t), dtype=np.uint8) * mask).astype(np.uint8)
    diffuse = rng.integers(0, 256, size=(t, t, 3), dtype=np.uint8)
    colors = [rng.integers(0, 256, size=(t, t, 3), dtype=np.uint8) for _ in range(n_lobes)]
    lambdas = [rng.integers(0, 256, size=(t, t, 3), dtype=np.uint8) for _ in range(n_lobes)]
    return {"alpha": alpha, "diffuse": diffuse, "colors": colors, "lambdas": lambdas}


def scaled_uv(mesh: TriMesh, texture_size: int) -> np.ndarray:
    """test_baking_texture_images.py:325-328: (uv - 1e-7) * SIZE as float32, clipped to [0, SIZE-1]."""
    uv = mesh.visual.uv - 1e-7
    uv = np.array(uv).astype(np.float32) * texture_size
    return np.clip(uv, 0, texture_size - 1)


def per_triangle_charts(mesh: TriMesh, texture_size: int, seed: int = 42, texels: float = 4.0):
    """SURVEY.md 8d's OTHER UV set for configs[4]: random per-triangle charts -- every triangle gets its own little
    right-angled chart (legs of ``texels`` texels) at a uniformly random place of the atlas, so neighbouring triangles
    share NO texel locality (the worst case for the texture fetch; the (azimuth, elevation) charts ``shell_mesh`` gives its
    shells are the contiguous-chart case, what an xatlas output looks like).  Per-vertex UVs cannot say that with shared
    vertices, so the mesh comes back with its vertices unshared (3 per face, ``faces = arange``): same triangles, same
    triangle ids.  Returns (mesh, uv already scaled to texels as ``scaled_uv`` does)."""
    rng = np.random.default_rng(seed)
    f = np.asarray(mesh.faces)
    v = np.asarray(mesh.vertices)[f.reshape(-1)]                       # [3F, 3]
    faces = np.arange(3 * f.shape[0], dtype=np.int64).reshape(-1, 3)
    t = float(texture_size)
    corner = rng.uniform(0.0, t - texels - 1.0, size=(f.shape[0], 2))
    uv = np.stack([corner, corner + np.array([texels, 0.0]), corner + np.array([0.0, texels])], axis=1).reshape(-1, 2)
    uv = np.clip(uv.astype(np.float32), 0, texture_size - 1)
    return TriMesh(v.copy(), faces, uv / t), uv
