"""Where the time of the reference-shaped eval loop goes (bench.py ``reference_route``): the ``test()`` closure of
train_finetune.py:575-629 over the reference-named entry points, piece by piece, with a device synchronisation after
every piece (so the pieces add up to MORE than the pipelined frame; the split tells what to fix, the bench leg tells
how fast it is).

    python tools/reference_route_profile.py [--frames 6] [--scaling 0.0] [--out profiles/r3/reference_route_profile.json]
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
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--scaling", type=float, default=0.0)
    ap.add_argument("--deform-log2-t", type=int, default=24)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    device = torch.device("cuda:0")
    from quadraturefields_amd import synthetic, utils
    from quadraturefields_amd.datasets.nerf_synthetic import SubjectLoader
    from quadraturefields_amd.field import Field
    from quadraturefields_amd.mesh_utils import MeshFinetune
    mesh, mi, field = bench.build_scene(device)
    W = H = 800
    n = args.frames + 2
    cams = np.stack([np.asarray(c, dtype=np.float32) for c in synthetic.orbit_cameras(n, seed=42)])
    ds = SubjectLoader.from_arrays(np.zeros((n, H, W, 4), dtype=np.uint8), cams, synthetic.lego_focal(W), split="test",
                                   mesh_intersect=mi, device=device)
    field_net = Field(scale=1.5, precision=16, log2_T=args.deform_log2_t, L=16, max_res=512, min_res=16, output_dim=1,
                      hidden_size=32, num_features=2, back_prop=False, nl="relu").to(device)
    mesh_finetune = MeshFinetune(mi.mesh.vertices, mi.mesh.faces, 0.0434, device=device)
    acc = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        acc.setdefault(name, []).append((t1 - t0) * 1e3)
        return t1

    def frame(i, record):
        torch.cuda.synchronize()
        t = t_start = time.perf_counter()
        item = ds[i]
        t = tick("loader_item", t) if record else t
        rays = item["rays"]
        splits = utils.generate_splits(item["data"], rays.origins.shape[0])
        t = tick("generate_splits", t) if record else t
        rgb = torch.ones((rays.origins.shape[0], 3), device=device)
        depth = torch.zeros((rays.origins.shape[0],), device=device)
        t = tick("alloc_outputs", t) if record else t
        for split in splits:
            color, _, d, _, _, _, _, _, _ = utils.render_image_finetune_with_occgrid(
                field, field_net, None, rays, split, near_plane=0.0, render_step_size=bench.STEP,
                render_bkgd=item["color_bkgd"], mesh_intersect=mi, mesh_finetune=mesh_finetune, scaling=args.scaling)
            t = tick("render_split", t) if record else t
            rgb[split[2]] = color[split[2]]
            depth[split[2]] = d.squeeze()[split[2]]
            t = tick("assemble", t) if record else t
        if record:
            acc.setdefault("frame_total_with_syncs", []).append((time.perf_counter() - t_start) * 1e3)

    for i in range(2):
        frame(i, False)
    for i in range(2, n):
        frame(i, True)
    per_frame = {k: float(np.sum(v) / args.frames) for k, v in acc.items()}
    # the same frames without the per-piece synchronisation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, n):
        frame(i, False)
    torch.cuda.synchronize()
    per_frame["frame_pipelined"] = (time.perf_counter() - t0) / args.frames * 1e3
    out = {"scaling": args.scaling, "frames": args.frames, "ms_per_frame": per_frame}
    print(json.dumps(out, indent=1))
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
