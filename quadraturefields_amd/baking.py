"""Producer side of the baked-texture format and triangle pruning.

* ``bake_texture_images``: ``examples/bake_texture_images_shelly.py:270-294`` -- evaluate the SG field's features
  and the density field at every valid texel of ``V`` (texel -> 3-D point, all-zero rows are empty), quantise with
  the reference's codecs and scatter into the uint8 texture set.
* ``triangle_max_weights`` / ``prune_faces``: ``examples/prune_mesh_after_finetuning.py:323-373`` -- per-triangle
  maximum compositing weight over the training views, faces below 1e-3 dropped.
"""
import numpy as np
import torch

from . import _C
from .mesh_io import TriMesh


@torch.no_grad()
def bake_texture_images(radiance_field_sg, radiance_field, V, compressor, batch_size: int = 100000):
    """Fill ``compressor``'s texture maps in place; returns the boolean texel mask (V.sum(-1) != 0)."""
    V = np.asarray(V, dtype=np.float32)
    mask = ~(V.sum(-1) == 0)
    ind = np.argwhere(mask)
    dev = compressor.device
    for b in range(0, ind.shape[0], batch_size):
        rows = ind[b:b + batch_size]
        pts = torch.from_numpy(V[rows[:, 0], rows[:, 1]]).to(dev)
        features = radiance_field_sg.features(pts)
        density = radiance_field.query_density(pts)
        features[..., -1] = density.flatten()
        compressor.load_features_into_maps(features, torch.from_numpy(rows).to(dev))
    return mask


def triangle_max_weights(weights: torch.Tensor, index_tri: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[f] = max(out[f], max of ``weights`` over the samples on triangle f) -- scatter_max, in place."""
    w = _C.f32c(weights.reshape(-1))
    idx = _C.i64c(index_tri.reshape(-1))
    if not (out.is_contiguous() and out.dtype == torch.float32):
        raise ValueError("out must be a contiguous float32 tensor")
    _C.check(_C.lib().qf_scatter_max(_C.ptr(w), _C.ptr(idx), w.shape[0], out.shape[0], _C.ptr(out), _C.stream()),
             "qf_scatter_max")
    return out


def prune_faces(mesh: TriMesh, triangle_weights, threshold: float = 1e-3) -> TriMesh:
    """Mesh with the faces whose maximum weight is <= threshold removed (vertices kept, as ``update_faces``)."""
    tw = triangle_weights.detach().cpu().numpy() if isinstance(triangle_weights, torch.Tensor) else np.asarray(triangle_weights)
    keep = tw.reshape(-1) > threshold
    return TriMesh(mesh.vertices.copy(), mesh.faces[keep], mesh.visual.uv)
