"""The finetune stage of the reference (examples/train_finetune.py) on a synthetic scene, written against
quadraturefields_amd with the reference's own call sequence:

    training step   (train_finetune.py:465-533)  estimator.update_every_n_steps -> render_image_finetune_with_occgrid
                                                  (deformation, re-sort, field, compositing, regulariser)
                                                  + render_image_with_occgrid (rgb_full) -> smooth-L1 -> Adam
    mesh update     (train_finetune.py:696-718)  MeshFinetune.update_faces / reset_d, BVH refit
    evaluation      (train_finetune.py:575-629)  whole frames, PSNR

There is no dataset in this environment: the "ground truth" images are rendered from a reference field with the same
pipeline, the trained field starts from a perturbed copy, and the mesh is the synthetic shell mesh.

    python examples/finetune_synthetic.py --steps 200 --image 200
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--image", type=int, default=200, help="training / evaluation image side in pixels")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--rays", type=int, default=1 << 14, help="rays per training step")
    ap.add_argument("--scaling", type=float, default=0.0434)
    ap.add_argument("--log2-hashmap-size", type=int, default=16)
    ap.add_argument("--update-mesh-every", type=int, default=100)
    args = ap.parse_args()

    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.utils import Rays
    from quadraturefields_amd.estimators import OccGridEstimator
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune, MeshIntersection, make_camera
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer, psnr

    torch.manual_seed(42)
    dev = torch.device("cuda:0")
    aabb = [-1.5] * 3 + [1.5] * 3
    step_size = 5e-3
    w = h = args.image
    T = args.log2_hashmap_size

    mesh = synthetic.shell_mesh(n_shells=6, subdivisions=5)
    mesh_intersect = MeshIntersection(mesh, simplify_mesh=False, scale=1.0, num_intersections=25, render_step_size=step_size)
    reference = NGPRadianceField(aabb=aabb, log2_hashmap_size=T)
    state = synthetic.seeded_ngp_state(T, reference.mlp_base.grid.n_rows)
    reference.load_state_dict(state, strict=False)
    reference = reference.to(dev)
    radiance_field = NGPRadianceField(aabb=aabb, log2_hashmap_size=T)
    g = torch.Generator().manual_seed(7)
    radiance_field.load_state_dict({k: v + 0.2 * v.abs().mean() * torch.randn(v.shape, generator=g) for k, v in state.items()},
                                   strict=False)
    radiance_field = radiance_field.to(dev)
    field_net = Field(scale=1.5, precision=16, log2_T=T, L=16, max_res=512, min_res=16, output_dim=1, hidden_size=32,
                      num_features=2, back_prop=False, nl="relu").to(dev)
    estimator = OccGridEstimator(roi_aabb=aabb, resolution=64, levels=1).to(dev)
    mesh_finetune = MeshFinetune(mesh.vertices, mesh.faces, args.scaling, device=dev)

    cams = synthetic.orbit_cameras(args.views, seed=3)
    focal = synthetic.lego_focal(800) * w / 800.0
    views = []
    with torch.no_grad():
        for c2w in cams:
            o, d = synthetic.camera_rays(c2w, focal, w, h, device=dev)
            pixels = FrameRenderer(mesh_intersect, reference).render(o, d, camera=make_camera(c2w, focal, w, h))[0]
            views.append((c2w, o, d, pixels))

    from quadraturefields_amd.optim import Adam          # torch.optim.Adam's update, one launch per tensor
    optimizer = Adam(list(radiance_field.parameters()) + list(field_net.parameters()), lr=2e-3, eps=1e-15)
    render_bkgd = torch.ones(3, device=dev)

    def evaluate(scaling):
        radiance_field.eval()
        with torch.no_grad():
            renderer = FrameRenderer(mesh_intersect, radiance_field, field_net=field_net, render_step_size=step_size)
            return float(np.mean([psnr(renderer.render(o, d, scaling=scaling, camera=make_camera(c2w, focal, w, h))[0], px)
                                  for c2w, o, d, px in views]))

    print(f"PSNR before training: {evaluate(args.scaling):.2f} dB")
    t0 = time.perf_counter()
    for step in range(args.steps + 1):
        radiance_field.train()
        estimator.train()
        c2w, o, d, pixels = views[step % len(views)]
        pick = torch.randint(0, w * h, (args.rays,), device=dev)
        rays = Rays(origins=o[pick].contiguous(), viewdirs=d[pick].contiguous())
        target = pixels[pick]
        estimator.update_every_n_steps(step=step, occ_eval_fn=lambda x: radiance_field.query_density(x) * step_size,
                                       occ_thre=1e-2)
        with torch.no_grad():
            data = mesh_intersect.sampling_raytrace_device(rays.viewdirs, rays.origins)     # the DataLoader's job
        if data is None:
            continue
        rgb, acc, depth, n_samples, weights, positions, index_ray, loss_reg, _ = utils.render_image_finetune_with_occgrid(
            radiance_field, field_net, estimator, rays, data, render_step_size=step_size, render_bkgd=render_bkgd,
            mesh_intersect=mesh_intersect, mesh_finetune=mesh_finetune, scaling=args.scaling)
        rgb_full, _, _, _, _ = utils.render_image_with_occgrid(radiance_field, estimator, rays, render_step_size=step_size,
                                                               render_bkgd=render_bkgd)
        loss = (F.smooth_l1_loss(rgb, target) + F.smooth_l1_loss(rgb_full, target)) / 2 + loss_reg.sum()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if step % 50 == 0:
            torch.cuda.synchronize()
            print(f"step {step:5d}  loss {float(loss.detach()):.5f}  samples {n_samples}  "
                  f"{(time.perf_counter() - t0) / (step + 1) * 1e3:.1f} ms/step")
        if step > 0 and step % args.update_mesh_every == 0:
            before = evaluate(args.scaling)
            mesh_finetune.update_faces()
            mesh_finetune.reset_d()
            mesh_intersect.mesh.vertices = mesh_finetune.vertices
            mesh_intersect.vertices = torch.from_numpy(mesh_finetune.vertices).to(dev)
            mesh_intersect.rayintersector.update_intersector(mesh_finetune.vertices)      # BVH refit
            print(f"step {step:5d}  mesh updated: PSNR with deformation {before:.2f} dB, on the moved mesh {evaluate(0):.2f} dB")
    print(f"PSNR after training: {evaluate(0):.2f} dB")


if __name__ == "__main__":
    main()
