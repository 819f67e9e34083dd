"""The "before" evaluation of the finetune stage (examples/train_finetune.py:575-629 called with the training
``scaling``): every quadrature point is moved along its ray by the deformation field (hash grid with T = 2^24, a 1 GB
fp32 table, + 35-32-32-1 MLP), re-sorted per ray, shaded and composited -- through ``FrameRenderer.render`` /
``render_image_finetune_with_occgrid``.  A scale exercise beside bench.py (whose line is the scaling = 0 test-set
evaluation): prints one JSON object.

    python tools/deformed_frame_bench.py --steps 10 --warmup 2 [--scaling 0.0434] [--deform-log2-t 24]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", type=float, default=0.0434)
    ap.add_argument("--deform-log2-t", type=int, default=24)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import make_camera
    from quadraturefields_amd.render import FrameRenderer
    device = torch.device("cuda:0")
    mesh, mi, field = bench.build_scene(device)
    net = Field(scale=1.5, precision=16, log2_T=args.deform_log2_t, L=16, max_res=512, min_res=16, output_dim=1,
                hidden_size=32, num_features=2, back_prop=False, nl="relu")
    net.load_state_dict(synthetic.seeded_deform_state(net.xyz_encoder.grid.n_params), strict=False)
    net = net.to(device)
    fr = FrameRenderer(mi, field, field_net=net)
    n_frames = args.steps + args.warmup
    cams = synthetic.orbit_cameras(n_frames, seed=42)
    focal = synthetic.lego_focal(bench.W)
    rays = [synthetic.camera_rays(cams[i], focal, bench.W, bench.H, device=device) for i in range(n_frames)]
    cameras = [make_camera(cams[i], focal, bench.W, bench.H) for i in range(n_frames)]
    for i in range(args.warmup):
        fr.render(rays[i][0], rays[i][1], scaling=args.scaling, camera=cameras[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts = 0
    for i in range(args.warmup, n_frames):
        pts += fr.render(rays[i][0], rays[i][1], scaling=args.scaling, camera=cameras[i])[3]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n_rays = bench.W * bench.H
    print(json.dumps({
        "workload": f"deformed ('before') evaluation frame 800x800, deformation table T=2^{args.deform_log2_t}, "
                    f"scaling {args.scaling}",
        "rays_per_s": n_rays * args.steps / el, "ms_per_frame": el / args.steps * 1e3,
        "points_per_frame": pts / args.steps,
        "deform_table_gb": net.xyz_encoder.grid.n_params * 4 / 1e9}))


if __name__ == "__main__":
    main()
