"""Occupancy-grid sample estimator on the gfx950 marching kernel.

Mirrors ``nerfacc.estimators.occ_grid.OccGridEstimator`` (nerfacc 0.5.3, un-vendored) as the reference uses it
(``examples/utils.py:137-147``; ``examples/train_finetune.py:369,409``): same constructor, the same buffers and
state-dict keys (``resolution``, ``aabbs``, ``occs``, ``binaries``) so ``estimator.load_state_dict(ckpt["estimator"])``
works, and ``sampling(...) -> (ray_indices, t_starts, t_ends)``.  One grid level and ``cone_angle == 0`` (the
reference's NeRF-synthetic configuration); the training-time ``update_every_n_steps`` is not implemented.
"""
import ctypes
from typing import Callable, List, Optional, Union

import torch
from torch import Tensor, nn

from . import _C
from .field_rendering import pack_info, render_visibility_from_alpha, render_visibility_from_density


class OccGridEstimator(nn.Module):
    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[float], Tensor], resolution: Union[int, List[int], Tensor] = 128,
                 levels: int = 1, **kwargs) -> None:
        super().__init__()
        if levels != 1:
            raise NotImplementedError("multi-level grids belong to the unbounded scenes (out of scope)")
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).cpu()
        assert resolution.shape[0] == self.DIM and roi_aabb.shape[0] == self.DIM * 2
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", roi_aabb[None, :].clone())
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))

    def _host_geometry(self):
        key = (self.aabbs.data_ptr(), self.aabbs._version, self.resolution._version)
        if getattr(self, "_geo_cache", None) is None or self._geo_cache[0] != key:
            aabb = (ctypes.c_float * 6)(*self.aabbs[0].detach().cpu().tolist())
            res = (ctypes.c_int32 * 3)(*[int(v) for v in self.resolution.detach().cpu().tolist()])
            self._geo_cache = (key, aabb, res)
        return self._geo_cache[1], self._geo_cache[2]

    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0):
        """(ray_indices int64 [S], t_starts [S], t_ends [S]), samples grouped by ray, front to back."""
        if cone_angle != 0.0:
            raise NotImplementedError("cone_angle > 0 (unbounded scenes) is out of scope")
        rays_o, rays_d = _C.f32c(rays_o.reshape(-1, 3)), _C.f32c(rays_d.reshape(-1, 3))
        n = rays_o.shape[0]
        dev = rays_o.device
        if stratified:      # nerfacc jitters the near planes by one step
            jitter = torch.rand(n, device=dev) * render_step_size
            t_min = jitter + near_plane if t_min is None else torch.clamp(t_min, min=near_plane) + jitter
        t_min = None if t_min is None else _C.f32c(t_min.reshape(-1))
        t_max = None if t_max is None else _C.f32c(t_max.reshape(-1))
        aabb, res = self._host_geometry()
        binaries = self.binaries[0].contiguous()
        count = torch.empty((n,), dtype=torch.int32, device=dev)
        lib = _C.lib()
        common = (aabb, res, _C.ptr(binaries), _C.ptr(rays_o), _C.ptr(rays_d), _C.ptr(t_min), _C.ptr(t_max), n,
                  float(near_plane), float(far_plane), float(render_step_size))
        _C.check(lib.qf_grid_march_count(*common, _C.ptr(count), _C.stream()), "qf_grid_march_count")
        csum = torch.cumsum(count.to(torch.int64), dim=0)
        total = int(csum[-1].item()) if n else 0
        offsets = (csum - count).contiguous()
        t_starts = torch.empty((total,), dtype=torch.float32, device=dev)
        t_ends = torch.empty((total,), dtype=torch.float32, device=dev)
        ray_indices = torch.empty((total,), dtype=torch.int64, device=dev)
        if total:
            _C.check(lib.qf_grid_march_write(*common, _C.ptr(offsets), _C.ptr(t_starts), _C.ptr(t_ends),
                                             _C.ptr(ray_indices), _C.stream()), "qf_grid_march_write")
        # skip invisible space (nerfacc: only when a field callback is given)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            alpha_thre = min(alpha_thre, float(self.occs.mean().item()))
            info = torch.stack([offsets, count.to(torch.int64)], dim=-1).contiguous()
            if sigma_fn is not None:
                sigmas = sigma_fn(t_starts, t_ends, ray_indices) if total else torch.empty((0,), device=dev)
                assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
                masks = render_visibility_from_density(t_starts, t_ends, sigmas, packed_info=info,
                                                       early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
            else:
                alphas = alpha_fn(t_starts, t_ends, ray_indices) if total else torch.empty((0,), device=dev)
                assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
                masks = render_visibility_from_alpha(alphas, packed_info=info, early_stop_eps=early_stop_eps,
                                                     alpha_thre=alpha_thre)
            ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
        return ray_indices, t_starts, t_ends

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """nerfacc 0.5.3 ``OccGridEstimator.update_every_n_steps`` (train_finetune.py:482-487): in training mode,
        every ``n`` steps, re-estimate the occupancy of a subset of cells -- all of them during the first
        ``warmup_steps`` steps, afterwards a quarter drawn uniformly plus the currently occupied ones -- at one random
        point per cell: ``occs = max(occs * ema_decay, occ_eval_fn(x))``, ``binaries = occs > min(mean(occs),
        occ_thre)``.  Restated from nerfacc's published algorithm (parity unpinned: the source is not vendored)."""
        if not self.training or step % n != 0:
            return
        res = [int(v) for v in self.resolution.tolist()]
        cells = res[0] * res[1] * res[2]
        dev = self.aabbs.device
        if step < warmup_steps:
            indices = torch.arange(cells, device=dev)
        else:
            uniform = torch.randint(cells, (cells // 4,), device=dev)
            occupied = torch.nonzero(self.binaries[0].flatten())[:, 0]
            if occupied.shape[0] > cells // 4:
                occupied = occupied[torch.randint(occupied.shape[0], (cells // 4,), device=dev)]
            indices = torch.cat([uniform, occupied], dim=0)
        coords = torch.stack([indices // (res[1] * res[2]), (indices // res[2]) % res[1], indices % res[2]], dim=-1)
        x = (coords.to(torch.float32) + torch.rand((indices.shape[0], 3), device=dev)) / torch.tensor(res, device=dev)
        x = self.aabbs[0, :3] + x * (self.aabbs[0, 3:] - self.aabbs[0, :3])
        occ = occ_eval_fn(x).reshape(-1).to(torch.float32)
        self.occs[indices] = torch.maximum(self.occs[indices] * ema_decay, occ)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.binaries.copy_((self.occs > thre).reshape(self.binaries.shape))

    @torch.no_grad()
    def set_occupancy_from_density(self, density_fn: Callable, threshold: float = 0.01, chunk: int = 1 << 20) -> None:
        """Inference-side helper (no reference counterpart): fill ``occs`` / ``binaries`` by evaluating
        ``density_fn(points [N,3]) -> [N] or [N,1]`` at the cell centres; occupied where density > threshold."""
        res = [int(v) for v in self.resolution.tolist()]
        dev = self.aabbs.device
        lo, hi = self.aabbs[0, :3], self.aabbs[0, 3:]
        axes = [(torch.arange(r, device=dev, dtype=torch.float32) + 0.5) / r for r in res]
        grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3) * (hi - lo) + lo
        occ = torch.cat([density_fn(grid[i:i + chunk]).reshape(-1) for i in range(0, grid.shape[0], chunk)])
        self.occs.copy_(occ)
        self.binaries.copy_((occ > threshold).reshape([1] + res))
