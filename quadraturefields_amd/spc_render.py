"""kaolin.render.spc duck types on the gfx950 compositing kernels.

The reference imports ``kaolin.render.spc as spc_render`` and calls ``mark_pack_boundaries``
(``examples/mesh_utils.py:407``, ``examples/utils.py:709``), ``exponential_integration`` and
``sum_reduce`` (``examples/utils.py:869-879``).  Same names, argument meaning and return shapes.
"""
import torch

from . import _C


def mark_pack_boundaries(ridx: torch.Tensor) -> torch.Tensor:
    """bool[S]: True at i == 0 and wherever ridx[i] != ridx[i-1]."""
    ridx = _C.i64c(ridx)
    n = ridx.shape[0]
    out = torch.empty((n,), dtype=torch.bool, device=ridx.device)
    _C.check(_C.lib().qf_mark_pack_boundaries(_C.ptr(ridx), n, _C.ptr(out), _C.stream()), "qf_mark_pack_boundaries")
    return out


def _segment_starts(boundary: torch.Tensor) -> torch.Tensor:
    # output size depends on the data: one device->host sync, exactly as kaolin's API implies
    return torch.nonzero(boundary.reshape(-1)).reshape(-1).to(torch.int64).contiguous()


def exponential_integration(feats: torch.Tensor, tau: torch.Tensor, boundary: torch.Tensor, exclusive: bool = True):
    """(sum_seg w*feats [R',C], w [S,1]) with w = exp(-cumsum(tau)) * (1 - exp(-tau)).
    NB the second value is the per-sample weight (the reference names it ``transmittance``, utils.py:869)."""
    feats = _C.f32c(feats)
    if feats.dim() == 1:
        feats = feats[:, None]
    n, c = feats.shape
    tau = _C.f32c(tau.reshape(-1))
    starts = _segment_starts(boundary)
    n_seg = starts.shape[0]
    out = torch.empty((n_seg, c), dtype=torch.float32, device=feats.device)
    w = torch.empty((n,), dtype=torch.float32, device=feats.device)
    _C.check(_C.lib().qf_exponential_integration(_C.ptr(feats), c, _C.ptr(tau), _C.ptr(starts), n_seg, n,
                                                 1 if exclusive else 0, _C.ptr(out), _C.ptr(w), _C.stream()),
             "qf_exponential_integration")
    return out, w[:, None]


def sum_reduce(feats: torch.Tensor, boundary: torch.Tensor) -> torch.Tensor:
    """Per-segment sum [R',C] in segment order."""
    feats = _C.f32c(feats)
    if feats.dim() == 1:
        feats = feats[:, None]
    n, c = feats.shape
    starts = _segment_starts(boundary)
    n_seg = starts.shape[0]
    out = torch.empty((n_seg, c), dtype=torch.float32, device=feats.device)
    _C.check(_C.lib().qf_sum_reduce(_C.ptr(feats), c, _C.ptr(starts), n_seg, n, _C.ptr(out), _C.stream()),
             "qf_sum_reduce")
    return out
