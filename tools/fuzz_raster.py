"""Fuzz of the camera-coherent intersector against the BVH traversal: random meshes (shell counts, subdivisions, scale),
random cameras (orbit radius 0.05..8, focal 0.2x..8x, off-centre principal points, non-square images, tilted).
Exits non-zero on the first mismatch.   python tools/fuzz_raster.py --cases 200
--dense: K = 2..6 with wide candidate lists of K+1..16 slots (qf_raster_intersect_wide + K-nearest selection + per-ray
BVH repair), compared as packed samples against the BVH path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--oracle", action="store_true",
                    help="also compare the BVH traversal with the CPU oracle's host-BVH walk on 400 random rays of every case "
                         "(the oracle is test infrastructure: this tool is a test, not product code)")
    ap.add_argument("--min-sep", default="trimesh", help="'trimesh' (default), or a distance, or 0")
    args = ap.parse_args()
    from quadraturefields_amd import _C, synthetic
    from quadraturefields_amd.mesh_utils import RayIntersector
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    meshes = {}
    compared = skipped = 0
    for case in range(args.cases):
        shells, sub = int(rng.integers(1, 7)), int(rng.integers(0, 6))
        key = (shells, sub)
        if key not in meshes:
            m = synthetic.shell_mesh(n_shells=shells, subdivisions=sub, seed=int(rng.integers(1 << 30)))
            sep = args.min_sep if args.min_sep == "trimesh" else float(args.min_sep)
            meshes[key] = RayIntersector(m, max_hits=25, min_separation=sep)
            if args.oracle:
                from oracle import meshpath as om
                meshes[key].oracle = om.BVHIntersector(m.vertices, m.faces, meshes[key].min_separation)
        ri = meshes[key]
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 400))
        c2w = synthetic.orbit_cameras(1, radius=float(np.exp(rng.uniform(np.log(0.05), np.log(8.0)))), seed=int(rng.integers(1 << 30)))[0].clone()
        # tilt: rotate the camera frame by a random small rotation, keep it orthonormal
        ang = rng.normal(size=3) * 0.3
        rx = torch.tensor([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]], dtype=torch.float32)
        ry = torch.tensor([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]], dtype=torch.float32)
        c2w[:3, :3] = c2w[:3, :3] @ rx @ ry
        focal = synthetic.lego_focal(w) * float(np.exp(rng.uniform(np.log(0.2), np.log(8.0))))
        cam = _C.Camera()
        for i in range(3):
            for j in range(4):
                cam.c2w[4 * i + j] = float(c2w[i, j])
        cam.fx, cam.fy = focal, focal * float(rng.uniform(0.7, 1.4))
        cam.cx, cam.cy = w / 2 + float(rng.uniform(-0.3, 0.3)) * w, h / 2 + float(rng.uniform(-0.3, 0.3)) * h
        cam.width, cam.height = w, h
        o = torch.empty((w * h, 3), device=dev)
        d = torch.empty((w * h, 3), device=dev)
        _C.check(_C.lib().qf_generate_rays(cam, 1, _C.ptr(o), _C.ptr(d), _C.stream()), "gen")
        if args.dense:
            k = int(rng.integers(2, 7))
            ri.raster_wide, ri._raster_backoff = int(rng.integers(k + 1, 17)), 0
            a = ri.sample_device(o, d, k, camera=cam)
            b = ri.sample_device(o, d, k, image_width=w)
            if (a is None) != (b is None) or (a is not None and not all(torch.equal(x, y) for x, y in zip(a, b))):
                print(f"MISMATCH case {case}: mesh {key} image {w}x{h} focal {focal:.1f} K {k} wide {ri.raster_wide}")
                sys.exit(1)
            compared += 1
            continue
        tri_r, t_r, cnt_r, ovf = ri._hits_raster(o, d, 25, cam)
        if int(ri._raster_words[1].item()):     # the pass's ray check (camera_rays_check) on rays that ARE this camera's grid
            print(f"GUARD FALSE POSITIVE case {case}: mesh {key} image {w}x{h} focal {focal:.1f} / {cam.fy:.1f} "
                  f"principal ({cam.cx:.1f}, {cam.cy:.1f})")
            sys.exit(1)
        if int(ovf.item()):
            skipped += 1
            continue
        tri_b, t_b, cnt_b = ri._hits_bvh(o, d, 25, w)
        if args.oracle:
            pick = torch.from_numpy(rng.choice(w * h, size=min(400, w * h), replace=False)).to(dev)
            tri_o, t_o, cnt_o = ri.oracle.hits(o[pick].cpu().numpy(), d[pick].cpu().numpy(), 25)
            if not (np.array_equal(cnt_b[pick].cpu().numpy(), cnt_o) and np.array_equal(tri_b[pick].cpu().numpy(), tri_o)
                    and np.array_equal(t_b[pick].cpu().numpy(), t_o)):
                print(f"ORACLE MISMATCH case {case}: mesh {key} image {w}x{h} focal {focal:.1f}")
                sys.exit(1)
        if not (torch.equal(cnt_r, cnt_b) and torch.equal(tri_r, tri_b) and torch.equal(t_r, t_b)):
            bad = torch.nonzero(cnt_r != cnt_b)[:5].flatten().tolist()
            print(f"MISMATCH case {case}: mesh {key} image {w}x{h} focal {focal:.1f} rays {bad}")
            sys.exit(1)
        compared += 1
    print(f"ok: {compared} cases identical, {skipped} skipped (overflow)")


if __name__ == "__main__":
    main()
