"""Times the general intersector (qf_bvh_intersect, the 8-wide tree walked by 8 lanes per ray) on the bench scene:
a full 800x800 camera frame (image-shaped, coherent), 2^17 random rays of random cameras (a finetune training batch,
examples/train_finetune.py:465-470) as they come and Morton-sorted by pixel, and the camera-coherent pass next to it.
Prints one JSON object.

    python tools/bvh_bench.py [--iters 10] [--min-sep trimesh|0]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import bench


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rays", type=int, default=1 << 17)
    ap.add_argument("--min-sep", default="trimesh")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from quadraturefields_amd import synthetic
    from quadraturefields_amd.mesh_utils import make_camera
    dev = torch.device("cuda:0")
    mesh, mi, field = bench.build_scene(dev)
    ri = mi.rayintersector
    if args.min_sep != "trimesh":
        ri.set_min_separation(float(args.min_sep))
    K, W, H = bench.MAX_HITS, bench.W, bench.H
    cams = synthetic.orbit_cameras(8, seed=1)
    focal = synthetic.lego_focal(W)
    rays = [synthetic.camera_rays(c, focal, W, H, device=dev) for c in cams]
    out = {"triangles": int(mesh.faces.shape[0]), "binary_nodes": ri.num_nodes, "wide_nodes": ri.num_wide_nodes,
           "max_stack": ri.max_stack, "min_separation": ri.min_separation}
    o, d = rays[0]
    out["frame_bvh_ms"] = timed(lambda: ri._hits_bvh(o, d, K, W), args.iters)
    out["frame_bvh_unshaped_ms"] = timed(lambda: ri._hits_bvh(o, d, K, 0), args.iters)
    cam = make_camera(cams[0], focal, W, H)
    out["frame_camera_coherent_ms"] = timed(lambda: ri._hits_raster_frame(o, d, K, cam), args.iters)
    pool_o, pool_d = torch.cat([r[0] for r in rays]), torch.cat([r[1] for r in rays])
    g = torch.Generator(device=dev).manual_seed(0)
    pick = torch.randint(0, pool_o.shape[0], (args.rays,), device=dev, generator=g)
    bo, bd = pool_o[pick].contiguous(), pool_d[pick].contiguous()
    out["batch_rays"] = args.rays
    out["batch_random_ms"] = timed(lambda: ri._hits_bvh(bo, bd, K, 0), args.iters)
    cnt = ri._hits_bvh(bo, bd, K, 0)[2]
    out["batch_mean_hits"] = float(cnt.float().mean())
    srt = torch.sort(pick).values          # by camera, then row-major pixel: coherent within a wave
    so, sd = pool_o[srt].contiguous(), pool_d[srt].contiguous()
    out["batch_sorted_ms"] = timed(lambda: ri._hits_bvh(so, sd, K, 0), args.iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
