"""Fuzz of the packers (qf_pack_samples, qf_pack_tiles: lists read straight into the sort network's registers, float4 /
scalar rows, 16- / 32-key networks, the insertion path above K = 32) and of the frame routes built on them.

Per case: a random shell mesh, K from a list that hits every list route (1..64), a random pinhole camera, then
  1. the six sample tensors of the camera-coherent route == those of the BVH route (same packer, different list order);
  2. on 300 random rays: the six tensors == the CPU oracle's (host-BVH walk + sampling_raytrace_numpy);
  3. FrameRenderer.render with the camera (tile pack + tile compositor) == without it (ray-major pack, re-sort,
     derive_properties), and render_async (one bound call) == render: pixels, alpha, depth bit for bit.
Exits non-zero on the first mismatch.   python tools/fuzz_pack.py --cases 300
(The oracle is test infrastructure: this tool is a test, not product code.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

KS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 20, 24, 25, 28, 31, 32, 33, 40, 64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from oracle import meshpath as om
    from quadraturefields_amd import _C, synthetic
    from quadraturefields_amd.mesh_utils import MeshIntersection
    from quadraturefields_amd.radiance_fields.ngp import NGPRadianceField
    from quadraturefields_amd.render import FrameRenderer
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    field = NGPRadianceField(aabb=[-1.5] * 3 + [1.5] * 3, log2_hashmap_size=12)
    field.load_state_dict(synthetic.seeded_ngp_state(12, field.mlp_base.grid.n_rows), strict=False)
    field = field.to(dev)
    scenes = {}
    names = ["xyzs", "dirs", "index_ray", "ts", "index_tri", "origins"]
    for case in range(args.cases):
        shells, sub, k = int(rng.integers(1, 21)), int(rng.integers(0, 4)), int(rng.choice(KS))
        key = (shells, sub, k)
        if key not in scenes:
            m = synthetic.shell_mesh(n_shells=shells, subdivisions=sub, seed=int(rng.integers(1 << 30)))
            mi = MeshIntersection(m, simplify_mesh=False, scale=1.0, num_intersections=k, device=dev)
            scenes[key] = (mi, om.BVHIntersector(m.vertices, m.faces, mi.rayintersector.min_separation))
        mi, oracle = scenes[key]
        ri = mi.rayintersector
        w, h = int(rng.integers(4, 120)), int(rng.integers(4, 120))
        c2w = synthetic.orbit_cameras(1, radius=float(np.exp(rng.uniform(np.log(0.3), np.log(6.0)))), seed=int(rng.integers(1 << 30)))[0].clone()
        focal = synthetic.lego_focal(max(w, h)) * float(np.exp(rng.uniform(np.log(0.4), np.log(3.0))))
        cam = _C.Camera()
        for i in range(3):
            for j in range(4):
                cam.c2w[4 * i + j] = float(c2w[i, j])
        cam.fx = cam.fy = focal
        cam.cx, cam.cy = w / 2, h / 2
        cam.width, cam.height = w, h
        o = torch.empty((w * h, 3), device=dev)
        d = torch.empty((w * h, 3), device=dev)
        _C.check(_C.lib().qf_generate_rays(cam, 1, _C.ptr(o), _C.ptr(d), _C.stream()), "gen")
        what = f"case {case}: shells {shells} subdiv {sub} K {k} image {w}x{h} focal {focal:.1f}"
        # 1. camera-coherent route vs BVH route
        ri.raster_wide, ri._raster_backoff = 0, 0
        a = ri.sample_device(o, d, k, camera=cam)
        b = ri.sample_device(o, d, k)
        if (a is None) != (b is None) or (a is not None and not all(torch.equal(x, y) for x, y in zip(a, b))):
            print("MISMATCH (coherent vs BVH route)", what)
            sys.exit(1)
        # 2. the oracle on a subset of the rays
        pick = torch.from_numpy(rng.choice(w * h, size=min(300, w * h), replace=False)).to(dev)
        op, dp = o[pick].contiguous(), d[pick].contiguous()
        got = ri.sample_device(op, dp, k)
        want = om.sampling_raytrace_numpy(oracle, dp.cpu().numpy(), op.cpu().numpy(), k)
        if (got is None) != (want is None):
            print("MISMATCH (oracle: hit / no hit)", what)
            sys.exit(1)
        if got is not None:
            for name, g, wv in zip(names, got, om.to_loader_tensors(want)):
                if g.shape != wv.shape or not torch.equal(g.cpu(), wv):
                    print(f"MISMATCH (oracle: {name})", what)
                    sys.exit(1)
        # 3. frame routes
        fr = FrameRenderer(mi, field)
        ri.raster_wide, ri._raster_backoff = 0, 0
        f1 = fr.render(o, d, camera=cam)
        f2 = fr.render(o, d)
        ri.raster_wide, ri._raster_backoff = 0, 0
        f3 = fr.render_async(o, d, cam)
        n3 = ri.frame_samples()
        for i in range(3):
            if not (torch.equal(f1[i], f2[i]) and torch.equal(f1[i], f3[i])):
                print("MISMATCH (frame routes, output %d)" % i, what)
                sys.exit(1)
        if not (f1[3] == f2[3] == n3):
            print("MISMATCH (sample counts %s %s %s)" % (f1[3], f2[3], n3), what)
            sys.exit(1)
    print(f"ok: {args.cases} cases identical ({len(scenes)} scenes)")


if __name__ == "__main__":
    main()
