"""Generates tests/golden/crop_ref.npz: BASELINE configs[0] / SURVEY.md 8c fixture (6) -- the 100x100 centre crop of
the 800x800 synthetic camera 0, rendered end to end on the CPU by the oracle (brute-force multi-hit intersection with
the reference's re-origin rule -> sampling_raytrace_numpy -> loader casts -> field -> derive_properties).

The reference's own field modules cannot be imported here (tinycudann / nerfacc / kaolin are absent), so
this fixture pins the ORACLE's end-to-end output (a regression vector the HIP path is also compared with), not the
reference's: the pieces of the reference that can be executed are pinned separately by gen_from_reference.py.

    python tests/golden/gen_crop.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import meshpath as om  # noqa: E402
from tests import helpers  # noqa: E402

W = H = 800
CROP = 100
LOG2_T = 14


def scene():
    """(mesh, product field module on the CPU -- only its state dict is used --, crop ray ids, origins, viewdirs)."""
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    mesh = synthetic.shell_mesh(n_shells=3, subdivisions=3, seed=42)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=LOG2_T)
    field.load_state_dict(synthetic.seeded_ngp_state(LOG2_T, field.mlp_base.grid.n_rows, seed=42), strict=False)
    c2w = synthetic.orbit_cameras(1, seed=42)[0]
    o, d = om.generate_rays(c2w, synthetic.lego_focal(W), W, H)
    y0 = x0 = 120                           # off-centre: the crop covers the object's rim and some background
    idx = (torch.arange(y0, y0 + CROP)[:, None] * W + torch.arange(x0, x0 + CROP)[None, :]).reshape(-1)
    return mesh, field, idx, o[idx].contiguous(), d[idx].contiguous()


def render_oracle(mesh, field, o, d, intersector=None):
    bf = intersector or om.BruteForceIntersector(mesh.vertices, mesh.faces)
    sample = om.sampling_raytrace_numpy(bf, d.numpy(), o.numpy(), 25)
    data = om.to_loader_tensors(sample)
    out = om.render_image_finetune(helpers.oracle_ngp_weights(field), None, data, o.shape[0])
    return out, data


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    mesh, field, idx, o, d = scene()
    (rgb, alpha, depth, n, weights, pts, iray, itri), data = render_oracle(mesh, field, o, d)
    np.savez_compressed(os.path.join(HERE, "crop_ref.npz"), rgb=rgb.numpy(), alpha=alpha.numpy(), depth=depth.numpy(),
                        n_samples=np.int64(n), index_ray=iray.numpy().astype(np.int32), index_tri=itri.numpy().astype(np.int32),
                        ray_ids=idx.numpy().astype(np.int32))
    print("crop_ref.npz:", n, "quadrature points,", float((alpha > 0).float().mean()), "of the crop hits the object")


if __name__ == "__main__":
    main()
