"""BASELINE config 3: "Shelly"-like 1920x1080 frame, T = 2^21, ~3 M triangles of thin
concentric shells (most object rays collect more than K = 25 candidates), bf16 tables + MLPs with fp32 accumulate.
A parity/scale exercise beside bench.py (whose line stays config 2): prints one JSON object with stage times.

    python tools/config3_bench.py --steps 5 --warmup 2 [--dtype fp32] [--intersector bvh]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--intersector", default="raster", choices=["raster", "bvh"])
    ap.add_argument("--shells", type=int, default=36)
    ap.add_argument("--subdiv", type=int, default=6)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    bench.W, bench.H, bench.LOG2_T = 1920, 1080, 21
    bench.N_SHELLS, bench.SUBDIV = args.shells, args.subdiv
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    device = torch.device("cuda:0")
    t0 = time.perf_counter()
    mesh, mi, field = bench.build_scene(device)
    field.compute_dtype = args.dtype
    bench.log(f"scene: {mesh.faces.shape[0]} triangles, {mi.rayintersector.num_nodes} BVH nodes, "
              f"{field.mlp_base.grid.n_rows} table rows, built in {time.perf_counter() - t0:.1f} s")
    n_frames = args.steps + args.warmup
    cams = synthetic.orbit_cameras(n_frames, seed=42)
    focal = synthetic.lego_focal(bench.W)
    rays = [synthetic.camera_rays(cams[i], focal, bench.W, bench.H, device=device) for i in range(n_frames)]
    cameras = [None if args.intersector == "bvh" else make_camera(cams[i], focal, bench.W, bench.H) for i in range(n_frames)]
    stages = bench.Stages(mi, field)
    for i in range(args.warmup):
        stages.frame(rays[i][0], rays[i][1], cameras[i], False)
        torch.cuda.synchronize()
        bench.log(f"warmup {i} done")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pts = 0
    for i in range(args.warmup, n_frames):
        pts += stages.frame(rays[i][0], rays[i][1], cameras[i], True)[3]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = stages.stage_ms()
    n_rays = bench.W * bench.H
    print(json.dumps({
        "workload": "config 3: 1920x1080, T=2^21, concentric thin shells, K=25", "dtype": args.dtype,
        "triangles": int(mesh.faces.shape[0]), "rays_per_frame": n_rays, "rays_per_s": n_rays * args.steps / el,
        "ms_per_frame": el / args.steps * 1e3, "points_per_frame": pts / args.steps,
        "mean_hits_per_ray": pts / args.steps / n_rays, "stage_ms": ms,
        "stage_ms_median": stages.stage_ms(np.median), "stage_ms_max": stages.stage_ms(np.max),
        "raster_wide": int(mi.rayintersector.raster_wide),
        "field_points_per_s_in_kernel": pts / args.steps / (ms["field"] * 1e-3),
        "intersector": args.intersector, "bvh_fallback_frames": getattr(stages, "fallbacks", 0)}))


if __name__ == "__main__":
    main()
