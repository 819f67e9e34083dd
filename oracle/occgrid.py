"""Oracle: occupancy-grid ray marching (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``nerfacc.estimators.occ_grid.OccGridEstimator.sampling`` (nerfacc 0.5.3, un-vendored; called at
``examples/utils.py:137-147``) for one grid level and ``cone_angle = 0``.  PARITY UNPINNED: nerfacc's CUDA
``traverse_grids`` is not in the container.  Its published behaviour: march from the clipped aabb entry t0 with a
constant step that keeps its phase through empty cells; a sample [t, t+dt] is emitted iff its midpoint lies before
the exit of an OCCUPIED cell.  Evaluated here per step in float32 with individually rounded operations
(t_k = t0 + k*dt), which the HIP kernel reproduces exactly; then the optional visibility filter of ``sampling``.
"""
import numpy as np
import torch

from . import volrend

F = np.float32


def _safe_inv(d):
    tiny = F(1e-30)
    d = d.copy()
    small = np.abs(d) < tiny
    d[small] = np.where(np.signbit(d[small]), -tiny, tiny)
    return (F(1.0) / d).astype(F)


def march(aabb, binaries, rays_o, rays_d, near_plane, far_plane, step, t_min=None, t_max=None):
    """aabb [6], binaries bool [rx,ry,rz], rays [R,3] float32 -> (ray_indices int64, t_starts, t_ends float32)."""
    aabb = np.asarray(aabb, dtype=F)
    lo, hi = aabb[:3], aabb[3:]
    res = np.array(binaries.shape, dtype=np.int64)
    o, d = np.asarray(rays_o, dtype=F), np.asarray(rays_d, dtype=F)
    step = F(step)
    out_r, out_s, out_e = [], [], []
    for r in range(o.shape[0]):
        inv = _safe_inv(d[r])
        a = ((lo - o[r]) * inv).astype(F)
        b = ((hi - o[r]) * inv).astype(F)
        tn, tf = np.max(np.minimum(a, b)), np.min(np.maximum(a, b))
        near_r = F(near_plane) if t_min is None else max(F(near_plane), F(t_min[r]))
        far_r = F(far_plane) if t_max is None else min(F(far_plane), F(t_max[r]))
        t0, t1 = max(tn, near_r), min(tf, far_r)
        if not (tn <= tf and t0 < t1):
            continue
        n_max = int(np.ceil((float(t1) - float(t0)) / float(step))) + 2
        k = np.arange(n_max, dtype=np.int64)
        ts = (t0 + (k.astype(F) * step).astype(F)).astype(F)
        te = (t0 + ((k + 1).astype(F) * step).astype(F)).astype(F)
        tm = ((ts + te).astype(F) * F(0.5)).astype(F)
        alive = tm < t1
        # the kernel stops at the first midpoint that is not < t1 (tm is non-decreasing)
        if not alive.all():
            alive[np.argmin(alive):] = False
        p = (o[r][None, :] + (d[r][None, :] * tm[:, None]).astype(F)).astype(F)
        u = ((((p - lo).astype(F) / (hi - lo).astype(F)).astype(F)) * res.astype(F)).astype(F)
        f = np.floor(u)
        inside = ((f >= 0) & (f < res.astype(F))).all(axis=1)
        c = np.clip(f.astype(np.int64), 0, res - 1)
        occ = binaries[c[:, 0], c[:, 1], c[:, 2]]
        keep = alive & inside & occ
        out_r.append(np.full(int(keep.sum()), r, dtype=np.int64))
        out_s.append(ts[keep])
        out_e.append(te[keep])
    if not out_r:
        return np.zeros(0, np.int64), np.zeros(0, F), np.zeros(0, F)
    return np.concatenate(out_r), np.concatenate(out_s), np.concatenate(out_e)


def sampling(aabb, binaries, occs_mean, rays_o, rays_d, sigma_fn=None, near_plane=0.0, far_plane=1e10,
             render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0):
    """OccGridEstimator.sampling: march, then drop samples that are occluded (T < early_stop_eps) or transparent."""
    ridx, ts, te = (torch.from_numpy(x) for x in march(aabb, np.asarray(binaries), rays_o, rays_d, near_plane,
                                                       far_plane, render_step_size))
    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None:
        alpha_thre = min(alpha_thre, float(occs_mean))
        sigmas = sigma_fn(ts, te, ridx) if ts.shape[0] else torch.empty(0)
        masks = volrend.render_visibility_from_density(ts, te, sigmas, ray_indices=ridx, n_rays=len(rays_o),
                                                       early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
        ridx, ts, te = ridx[masks], ts[masks], te[masks]
    return ridx, ts, te
